// xworld_amd/csrc/xwb_language.h -- the teacher's sentences of XWorld2D, host side (the C ABI's copy of xworld_amd/language.py).
//
// In the reference every task owns a context-free grammar (python/context_free_grammar.py; games/xworld3d/tasks/
// XWorld3DNav*.py and games/xworld/tasks/XWorldNav*.py: `_define_grammar`); its idle stage binds the start symbol and the
// goal names and calls CFG.generate(), a left-most expansion that draws random.choice for every non-terminal.  Here the
// sentence of an env is a pure function of the batch state (task, bound goal names, direction word, stage, event,
// episode) and of xwb-rng-v1 stream 3 ("language": key = (seed, global env id), counter = (block, episode, 3, 0); one
// below(n) per expanded non-terminal, also when it is bound), so nothing is stored per env and the device never sees
// strings.  The rule texts, the stream and the order of the draws are those of language.py; tests/test_gpu_language_c.py
// compares the two sentence for sentence, and language.py is pinned to the reference's CFG by tests/golden/sentences.json.
#pragma once
#include <cstdint>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace xwb {
namespace lang {

// Philox4x32-10 (Salmon et al. 2011), host copy of xwb_common.h's
inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// xwb-rng-v1 stream: words of successive blocks; below(n) = (u32 * n) >> 32 and always consumes one word
struct Stream {
    uint32_t k0, k1, episode, sid, blk = 0;
    uint32_t buf[4];
    int left = 0;
    Stream(uint32_t seed, uint32_t gid, uint32_t ep, uint32_t stream_id) : k0(seed), k1(gid), episode(ep), sid(stream_id) {}
    uint32_t below(uint32_t n) {
        if (left == 0) {
            buf[0] = blk; buf[1] = episode; buf[2] = sid; buf[3] = 0;
            philox4x32_10(buf, k0, k1);
            blk++;
            left = 4;
        }
        const uint32_t v = buf[4 - left];
        left--;
        return n > 1 ? (uint32_t)(((uint64_t)v * n) >> 32) : 0u;
    }
};

typedef std::map<std::string, std::string> Bindings;

inline std::vector<std::string> split_ws(const std::string &s) {
    std::vector<std::string> out;
    std::istringstream is(s);
    std::string w;
    while (is >> w) out.push_back(w);
    return out;
}

// Rules `X -> a b | 'c' d` (terminals in single quotes).  expand() = CFG.generate(): left-most derivation, one choice per
// non-terminal; a binding narrows a rule to one alternative.
struct Grammar {
    std::map<std::string, std::vector<std::vector<std::string>>> rules;
    explicit Grammar(const std::string &text) {
        std::istringstream is(text);
        std::string line;
        while (std::getline(is, line)) {
            const size_t arrow = line.find("->");
            if (arrow == std::string::npos) continue;
            const std::vector<std::string> lhs = split_ws(line.substr(0, arrow));
            if (lhs.empty()) continue;
            std::vector<std::vector<std::string>> alts;
            std::string rhs = line.substr(arrow + 2);
            size_t pos = 0;
            while (true) {
                const size_t bar = rhs.find('|', pos);
                alts.push_back(split_ws(rhs.substr(pos, bar == std::string::npos ? std::string::npos : bar - pos)));
                if (bar == std::string::npos) break;
                pos = bar + 1;
            }
            rules[lhs[0]] = alts;
        }
    }
    template <typename Choose>
    void gen(const std::string &sym, Choose &choose, const Bindings &b, std::vector<std::string> &out) const {
        if (sym[0] == '\'') { out.push_back(sym.substr(1, sym.size() - 2)); return; }
        const Bindings::const_iterator it = b.find(sym);
        std::vector<std::vector<std::string>> bound;
        const std::vector<std::vector<std::string>> *alts;
        if (it != b.end()) { bound.push_back(split_ws(it->second)); alts = &bound; }
        else alts = &rules.at(sym);
        const std::vector<std::string> alt = (*alts)[choose((uint32_t)alts->size())];
        for (size_t i = 0; i < alt.size(); ++i) gen(alt[i], choose, b, out);
    }
    template <typename Choose>
    std::string expand(Choose &choose, const Bindings &b) const {
        std::vector<std::string> words;
        gen("S", choose, b, words);
        std::string s;
        for (size_t i = 0; i < words.size(); ++i) { if (i) s += ' '; s += words[i]; }
        return s;
    }
};

struct First { uint32_t operator()(uint32_t) { return 0; } };
struct Draw { Stream &st; uint32_t operator()(uint32_t n) { return st.below(n); } };

// task id (include/xwb.h XWB_TASK_*) -> grammar; nullptr: the task never speaks
inline const Grammar *grammar_of(int task) {
    static const std::string common =
        "S -> start | timeup | correct | wrong\n"
        "correct -> 'Well' 'done' '!'\n"
        "wrong -> 'Wrong' '!'\n"
        "timeup -> 'Time' 'up' '.'\n"
        "Y -> 'Could' 'you' 'please' | 'Can' 'you' | 'Will' 'you'\n"
        "D -> 'destination' | 'target' | 'goal' | 'end'\n";
    static const std::string go5 = "A -> 'go' 'to' | 'navigate' 'to' | 'reach' | 'move' 'to' | 'collect'\n";
    static const std::string go4 = "A -> 'go' 'to' | 'navigate' 'to' | 'reach' | 'move' 'to'\n";
    static const std::string common2d =
        "S -> start | finish | timeup\n"
        "finish -> 'Well' 'done' '!'\n"
        "timeup -> 'Time' 'up' '.'\n"
        "A -> 'go' 'to' | 'navigate' 'to' | 'reach' | 'move' 'to'\n"
        "Y -> 'Could' 'you' 'please' | 'Can' 'you' | 'Will' 'you'\n"
        "D -> 'destination' | 'target' | 'goal'\n";
    static const Grammar g0(common + go5 +
        "start -> I0 | I1 | I2 | I3 | I4 | I5 | I6\n"
        "I0 -> G\n"
        "I1 -> A G 'please' '.'\n"
        "I2 -> 'Please' A G '.'\n"
        "I3 -> A G '.'\n"
        "I4 -> G 'is' 'your' D '.'\n"
        "I5 -> G 'is' 'the' D '.'\n"
        "I6 -> Y A G '?'\n");
    static const Grammar g1(common + go5 +
        "start -> I0 | I1 | I2 | I3 | I4\n"
        "I0 -> A NP G\n"
        "I1 -> A NP G 'please' '.'\n"
        "I2 -> 'Please' A NP G '.'\n"
        "I3 -> NP G 'is' 'your' D '.'\n"
        "I4 -> Y A NP G '?'\n"
        "NP -> 'the' 'object' N\n"
        "N -> 'near' | 'by' | 'besides'\n");
    static const Grammar g2(common + go4 +
        "start -> I0 | I1 | I2 | I3 | I4\n"
        "I0 -> A L B '.'\n"
        "I1 -> A L B 'please' '.'\n"
        "I2 -> 'Please' A L B '.'\n"
        "I3 -> L B 'is' 'your' D '.'\n"
        "I4 -> Y A L B '?'\n"
        "B -> 'between' G1 'and' G2\n"
        "L -> 'the' 'location' | 'the' 'grid' | 'the' 'place'\n");
    static const Grammar g3(common + go5 +
        "start -> I0 | I1 | I2 | I3 | I4\n"
        "I0 -> A NP G '.'\n"
        "I1 -> A NP G 'please' '.'\n"
        "I2 -> 'Please' A NP G '.'\n"
        "I3 -> NP G 'is' 'your' D '.'\n"
        "I4 -> Y A NP G '?'\n"
        "NP -> 'the' 'object' P | 'the' 'object' 'that' 'is' P\n"
        "P -> LEFT | RIGHT | BEHIND | FRONT\n"
        "LEFT -> 'left' 'of' | 'to' 'the' 'left' 'of'\n"
        "RIGHT -> 'right' 'of' | 'to' 'the' 'right' 'of'\n"
        "BEHIND -> 'behind'\n"
        "FRONT -> 'in' 'the' 'front' 'of' | 'front' 'of'\n");
    static const Grammar g4(common + go5 +
        "start -> I0 | I1 | I2 | I4 | I5 | I6\n"
        "I0 -> V G '.'\n"
        "I1 -> V G 'please' '.'\n"
        "I2 -> 'Please' V G '.'\n"
        "I4 -> E G 'is' 'your' D '.'\n"
        "I5 -> E G 'is' 'the' D '.'\n"
        "I6 -> Y VV G '?'\n"
        "V -> 'do' 'not' A | 'avoid'\n"
        "VV -> 'not' A | 'avoid'\n"
        "E -> 'anything' 'except' | 'anything' 'but'\n");
    static const Grammar g5(common2d +
        "start -> I1 | I2 | I3 | I4 | I5 | I6\n"
        "I1 -> A G 'please' '.'\n"
        "I2 -> 'Please' A G '.'\n"
        "I3 -> A G '.'\n"
        "I4 -> G 'is' 'your' D '.'\n"
        "I5 -> G 'is' 'the' D '.'\n"
        "I6 -> Y A G '?'\n");
    static const Grammar g7(common2d +
        "start -> I1 | I2 | I3 | I4 | I5 | I6 | I7\n"
        "I1 -> A G 'please' '.'\n"
        "I2 -> 'Please' A G '.'\n"
        "I3 -> A G '.'\n"
        "I4 -> G 'is' 'your' D '.'\n"
        "I5 -> G 'is' 'the' D '.'\n"
        "I6 -> Y A G '?'\n"
        "I7 -> G '.'\n"
        "G -> C O\n");
    switch (task) {
        case 0: return &g0;
        case 1: return &g1;
        case 2: return &g2;
        case 3: return &g3;
        case 4: return &g4;
        case 5: return &g5;
        case 7: return &g7;
        default: return nullptr;
    }
}

inline std::string quoted(const std::string &name) { return "'" + name + "'"; }

// language.sentence(): a 3-D task's sentence after the last call ("" where the reference's get_state() shows "-").
// stage / event as in xwb_env_state (1 = navigation; 1 correct, 2 wrong, 3 time-up); name_a / name_b = goal-name ids bound
// at the idle stage (0xffff: none); direction: xw_device.h DIR_* (1 front, 2 behind, 3 left, 4 right)
inline std::string sentence(int task, int stage, int event, const std::vector<std::string> &goal_names, uint32_t name_a, uint32_t name_b,
                            int direction, uint32_t seed, uint32_t gid, uint32_t episode) {
    const Grammar *g = grammar_of(task);
    if (!g) return "";
    static const char *const event_rule[4] = {nullptr, "correct", "wrong", "timeup"};
    if (event >= 1 && event <= 3) {
        First f;
        Bindings b;
        b["S"] = event_rule[event];
        return g->expand(f, b);
    }
    if (stage != 1 || name_a == 0xFFFFu || name_a >= goal_names.size()) return "";
    Stream st(seed, gid, episode, 3);
    Draw d{st};
    Bindings b;
    b["S"] = "start";
    if (task == 2) {
        if (name_b >= goal_names.size()) return "";
        b["G1"] = quoted(goal_names[name_a]); b["G2"] = quoted(goal_names[name_b]);
    } else {
        b["G"] = quoted(goal_names[name_a]);
    }
    if (task == 3) {
        static const char *const words[5] = {"", "FRONT", "BEHIND", "LEFT", "RIGHT"};
        if (direction < 1 || direction > 4) return "";
        b["P"] = words[direction];
    }
    return g->expand(d, b);
}

// language.sentence_2d(): the instruction of a 2-D-native task on the teach() call that picked its target
inline std::string sentence_2d(int task, const std::string &goal_name, const std::string &color, uint32_t seed, uint32_t gid,
                               uint32_t episode, uint32_t num_steps) {
    const Grammar *g = grammar_of(task);
    if (!g) return "";
    Stream st(seed, gid, episode, 3);
    st.blk = 4 * num_steps;
    Draw d{st};
    Bindings b;
    b["S"] = "start";
    if (task == 7) { b["O"] = quoted(goal_name); b["C"] = quoted(color); }
    else b["G"] = quoted(goal_name);
    return g->expand(d, b);
}

inline std::string sentence_2d_timeup(int task) {
    const Grammar *g = grammar_of(task);
    if (!g) return "";
    First f;
    Bindings b;
    b["S"] = "timeup";
    return g->expand(f, b);
}

}  // namespace lang
}  // namespace xwb

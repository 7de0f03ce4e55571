"""The teacher's sentences (the language side channel of XWorld2D), host side.

In the reference every task owns a context-free grammar (python/context_free_grammar.py, games/xworld3d/tasks/
XWorld3DNav*.py: `_define_grammar`); its idle stage binds the start symbol and the goal names and calls
`CFG.generate()`, a left-most expansion that draws `random.choice` for every non-terminal; the sentence is repeated
every step of the episode, replaced by the "correct" / "wrong" / "timeup" message on the step that ends it, and empty
("-" in get_state) afterwards.  Here the sentence of an env is a pure function of the batch state (task, bound goal
names, direction word, stage, event, episode) and of the xwb-rng-v1 stream 3 ("language": key = (seed, global env id),
counter = (block, episode, 3, 0); one `below(n)` per expanded non-terminal, also when it is bound), so nothing has to
be stored per env and the device never sees strings.  The 2-D-native group (rule D14b; games/xworld/tasks/XWorldNav*.py)
speaks only on the teach() call that picks a target (its navigation stage returns ""): XWorldNavTarget /
XWorldNavColorTarget instructions are expanded from stream 3 starting at block 4 * num_steps; the one_channel mode's
time-up step says "Time up ." (S -> timeup).  XWorldNavNear / XWorldNavBetween never find a target in this snapshot of
the reference and so never speak.

Pinned by tests/golden/sentences.json: the reference's CFG run on each task's own grammar, replayed decision by
decision (tests/test_language.py).
"""

MASK32 = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon et al. 2011), the generator behind every xwb-rng-v1 stream."""
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK32, p1 & MASK32, ((p0 >> 32) ^ c3 ^ k1) & MASK32, p0 & MASK32
        k0 = (k0 + 0x9E3779B9) & MASK32
        k1 = (k1 + 0xBB67AE85) & MASK32
    return c0, c1, c2, c3


class Stream:
    """xwb-rng-v1 stream: words of successive blocks; below(n) = (u32 * n) >> 32 and always consumes one word."""

    def __init__(self, seed, gid, episode, stream_id):
        self.key = (seed & MASK32, gid & MASK32)
        self.episode, self.sid, self.blk, self.buf = episode & MASK32, stream_id, 0, []

    def below(self, n):
        if not self.buf:
            self.buf = list(philox4x32_10((self.blk, self.episode, self.sid, 0), self.key))
            self.blk += 1
        v = self.buf.pop(0)
        return (v * n) >> 32 if n > 1 else 0


class Grammar:
    """Rules `X -> a b | 'c' d` (terminals in single quotes).  expand() = CFG.generate(): left-most derivation, one
    choice per non-terminal; bind() narrows a rule to one alternative for the next expand() only."""

    def __init__(self, rules):
        self.rules = {}
        for line in rules.splitlines():
            if not line.strip():
                continue
            lhs, rhs = line.split("->")
            self.rules[lhs.strip()] = [alt.split() for alt in rhs.split("|")]

    def expand(self, choose, bindings, symbol="S"):
        def gen(sym):
            if sym[0] == "'":
                return [sym[1:-1]]
            alts = [bindings[sym].split()] if sym in bindings else self.rules[sym]
            alt = alts[choose(len(alts))]
            return [w for s in alt for w in gen(s)]
        return " ".join(gen(symbol))


_COMMON = """
S -> start | timeup | correct | wrong
correct -> 'Well' 'done' '!'
wrong -> 'Wrong' '!'
timeup -> 'Time' 'up' '.'
Y -> 'Could' 'you' 'please' | 'Can' 'you' | 'Will' 'you'
D -> 'destination' | 'target' | 'goal' | 'end'
"""
_GO5 = "A -> 'go' 'to' | 'navigate' 'to' | 'reach' | 'move' 'to' | 'collect'\n"
_GO4 = "A -> 'go' 'to' | 'navigate' 'to' | 'reach' | 'move' 'to'\n"

# task id (include/xwb.h XWB_TASK_*) -> grammar; goal-name rules (G, G1, G2) are always bound
GRAMMARS = {
    0: Grammar(_COMMON + _GO5 + """
start -> I0 | I1 | I2 | I3 | I4 | I5 | I6
I0 -> G
I1 -> A G 'please' '.'
I2 -> 'Please' A G '.'
I3 -> A G '.'
I4 -> G 'is' 'your' D '.'
I5 -> G 'is' 'the' D '.'
I6 -> Y A G '?'
"""),
    1: Grammar(_COMMON + _GO5 + """
start -> I0 | I1 | I2 | I3 | I4
I0 -> A NP G
I1 -> A NP G 'please' '.'
I2 -> 'Please' A NP G '.'
I3 -> NP G 'is' 'your' D '.'
I4 -> Y A NP G '?'
NP -> 'the' 'object' N
N -> 'near' | 'by' | 'besides'
"""),
    2: Grammar(_COMMON + _GO4 + """
start -> I0 | I1 | I2 | I3 | I4
I0 -> A L B '.'
I1 -> A L B 'please' '.'
I2 -> 'Please' A L B '.'
I3 -> L B 'is' 'your' D '.'
I4 -> Y A L B '?'
B -> 'between' G1 'and' G2
L -> 'the' 'location' | 'the' 'grid' | 'the' 'place'
"""),
    3: Grammar(_COMMON + _GO5 + """
start -> I0 | I1 | I2 | I3 | I4
I0 -> A NP G '.'
I1 -> A NP G 'please' '.'
I2 -> 'Please' A NP G '.'
I3 -> NP G 'is' 'your' D '.'
I4 -> Y A NP G '?'
NP -> 'the' 'object' P | 'the' 'object' 'that' 'is' P
P -> LEFT | RIGHT | BEHIND | FRONT
LEFT -> 'left' 'of' | 'to' 'the' 'left' 'of'
RIGHT -> 'right' 'of' | 'to' 'the' 'right' 'of'
BEHIND -> 'behind'
FRONT -> 'in' 'the' 'front' 'of' | 'front' 'of'
"""),
    4: Grammar(_COMMON + _GO5 + """
start -> I0 | I1 | I2 | I4 | I5 | I6
I0 -> V G '.'
I1 -> V G 'please' '.'
I2 -> 'Please' V G '.'
I4 -> E G 'is' 'your' D '.'
I5 -> E G 'is' 'the' D '.'
I6 -> Y VV G '?'
V -> 'do' 'not' A | 'avoid'
VV -> 'not' A | 'avoid'
E -> 'anything' 'except' | 'anything' 'but'
"""),
}

_COMMON_2D = """
S -> start | finish | timeup
finish -> 'Well' 'done' '!'
timeup -> 'Time' 'up' '.'
A -> 'go' 'to' | 'navigate' 'to' | 'reach' | 'move' 'to'
Y -> 'Could' 'you' 'please' | 'Can' 'you' | 'Will' 'you'
D -> 'destination' | 'target' | 'goal'
"""
GRAMMARS[5] = Grammar(_COMMON_2D + """
start -> I1 | I2 | I3 | I4 | I5 | I6
I1 -> A G 'please' '.'
I2 -> 'Please' A G '.'
I3 -> A G '.'
I4 -> G 'is' 'your' D '.'
I5 -> G 'is' 'the' D '.'
I6 -> Y A G '?'
""")
GRAMMARS[7] = Grammar(_COMMON_2D + """
start -> I1 | I2 | I3 | I4 | I5 | I6 | I7
I1 -> A G 'please' '.'
I2 -> 'Please' A G '.'
I3 -> A G '.'
I4 -> G 'is' 'your' D '.'
I5 -> G 'is' 'the' D '.'
I6 -> Y A G '?'
I7 -> G '.'
G -> C O
""")

DIRECTION_WORDS = {1: "FRONT", 2: "BEHIND", 3: "LEFT", 4: "RIGHT"}      # xw_device.h DIR_*
EVENT_RULE = {1: "correct", 2: "wrong", 3: "timeup"}                    # xw_device.h EV_*


def instruction_bindings(task, name_a, name_b=None, direction=0):
    b = {"S": "start"}
    if task == 2:
        b["G1"], b["G2"] = "'%s'" % name_a, "'%s'" % name_b
    else:
        b["G"] = "'%s'" % name_a
    if task == 3:
        b["P"] = DIRECTION_WORDS[direction]
    return b


def sentence_2d(task, goal_name, color, seed, gid, episode, num_steps):
    """The instruction of a 2-D-native task on the teach() call that picked its target (XWorldNavTarget.py:22-33,
    XWorldNavColorTarget.py:8-20): G (or O and C) bound, the rest drawn from stream 3, blocks 4 * num_steps onwards."""
    st = Stream(seed, gid, episode, 3)
    st.blk = 4 * num_steps
    b = {"S": "start"}
    if task == 7:
        b["O"], b["C"] = "'%s'" % goal_name, "'%s'" % color
    else:
        b["G"] = "'%s'" % goal_name
    return GRAMMARS[task].expand(st.below, b)


def sentence_2d_timeup(task):
    """xworld_task.py:205-211: `self._bind("S -> timeup")`, then generate() -- no free choice left."""
    return GRAMMARS[task].expand(lambda n: 0, {"S": "timeup"})


def sentence(task, stage, event, goal_names, name_a, name_b, direction, seed, gid, episode):
    """The teacher's sentence of one env after the last call ("" where the reference's get_state() shows "-").

    stage / event as in xwb_env_state (1 = navigation, 2 = terminal; 1 correct, 2 wrong, 3 time-up); name_a / name_b =
    goal-name ids bound at the idle stage (0xffff: none)."""
    if task not in GRAMMARS:
        return ""
    if event in EVENT_RULE:
        return GRAMMARS[task].expand(lambda n: 0, {"S": EVENT_RULE[event]})
    if stage != 1 or name_a == 0xFFFF:
        return ""
    st = Stream(seed, gid, episode, 3)
    b = instruction_bindings(task, goal_names[name_a], goal_names[name_b] if name_b != 0xFFFF else None, direction)
    return GRAMMARS[task].expand(st.below, b)

// include/xwb_simulator.hpp -- header-only C++11 surface over the C ABI of libxwb.so.
//
// Mirrors, for holders of the reference's C++ types:
//   xwb::StatePacket / StateBuffer   simulator::StatePacket (data_packet.h:185-381): typed map
//                                    key -> {reals | pixels, ids, str}; encode()/decode() use the reference's
//                                    wire layout (data_packet.h:313-333, data_packet.cpp:143-174,
//                                    memory_util.h:307-333) byte for byte.
//   xwb::BatchedSimulator            the batch itself (RAII over xwb_sim*).
//   xwb::SimulatorInterface          simulator::SimulatorInterface (simulator_interface.h:40-89): the same verbs
//                                    and signatures, bound to ONE env slot of a batch.  take_actions() steps only
//                                    that slot (every other env gets XWB_ACTION_SKIP); with a 1-env batch this is
//                                    exactly the reference object.
// Errors: the reference aborts (CHECK / LOG(FATAL)); here every failing call throws xwb::Error.
#pragma once

#include "xwb.h"

#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <cstdio>
#include <string>
#include <vector>

namespace xwb {

struct Error : std::runtime_error {
    explicit Error(const std::string &m) : std::runtime_error(m) {}
};

inline void check(int rc) {
    if (rc != XWB_OK) throw Error(std::string("xwb: ") + xwb_last_error());
}

// ---------------------------------------------------------------- StatePacket ----
struct StateBuffer {
    bool has_reals = false, has_pixels = false, has_id = false, has_str = false;
    std::vector<float> reals;
    std::vector<uint8_t> pixels;
    std::vector<int32_t> id;
    std::string str;

    size_t get_value_size() const { return has_reals ? reals.size() : (has_pixels ? pixels.size() : 0); }
    size_t get_id_size() const { return has_id ? id.size() : 0; }
    bool operator==(const StateBuffer &o) const {
        return has_reals == o.has_reals && has_pixels == o.has_pixels && has_id == o.has_id && has_str == o.has_str &&
               reals == o.reals && pixels == o.pixels && id == o.id && str == o.str;
    }
};

class StatePacket {
  public:
    void add_key(const std::string &k) { data_[k]; }
    bool contain_key(const std::string &k) const { return data_.count(k) != 0; }
    size_t size() const { return data_.size(); }
    std::vector<std::string> get_keys() const {
        std::vector<std::string> ks;
        for (auto &kv : data_) ks.push_back(kv.first);
        return ks;
    }
    StateBuffer &get_buffer(const std::string &k) {
        auto it = data_.find(k);
        if (it == data_.end()) throw Error("StatePacket: no key " + k);
        return it->second;
    }
    const StateBuffer &get_buffer(const std::string &k) const {
        auto it = data_.find(k);
        if (it == data_.end()) throw Error("StatePacket: no key " + k);
        return it->second;
    }
    void add_buffer_id(const std::string &k, const std::vector<int32_t> &v) { auto &b = data_[k]; b.has_id = true; b.id = v; }
    void add_buffer_str(const std::string &k, const std::string &v) { auto &b = data_[k]; b.has_str = true; b.str = v; }
    void add_buffer_value(const std::string &k, const std::vector<float> &v) {
        auto &b = data_[k]; b.has_reals = true; b.has_pixels = false; b.reals = v;
    }
    void add_buffer_value(const std::string &k, const std::vector<uint8_t> &v) {
        auto &b = data_[k]; b.has_pixels = true; b.has_reals = false; b.pixels = v;
    }

    // DataPacket<T>::encode (data_packet.h:313-319)
    std::vector<uint8_t> encode() const {
        std::vector<uint8_t> out;
        put_u64(out, data_.size());
        for (auto &kv : data_) {
            put_str(out, kv.first);
            const StateBuffer &b = kv.second;
            out.push_back((uint8_t)((b.has_reals ? 1 : 0) | (b.has_pixels ? 2 : 0) | (b.has_id ? 4 : 0) | (b.has_str ? 8 : 0)));
            if (b.has_reals) { put_u64(out, b.reals.size()); put(out, b.reals.data(), 4 * b.reals.size()); }
            if (b.has_pixels) { put_u64(out, b.pixels.size()); put(out, b.pixels.data(), b.pixels.size()); }
            if (b.has_id) { put_u64(out, b.id.size()); put(out, b.id.data(), 4 * b.id.size()); }
            if (b.has_str) put_str(out, b.str);
        }
        return out;
    }

    // DataPacket<T>::decode (data_packet.h:321-333); throws on a truncated buffer (reference: CHECK_LE)
    void decode(const uint8_t *p, size_t len) {
        data_.clear();
        size_t at = 0;
        uint64_t n = get_u64(p, len, at);
        for (uint64_t i = 0; i < n; ++i) {
            std::string key = get_str(p, len, at);
            StateBuffer b;
            uint8_t flags = *take(p, len, at, 1);
            // element counts come from the peer: checked against the bytes that are left (without overflow) BEFORE anything
            // is sized by them
            if (flags & 1) { b.has_reals = true; const size_t m = get_count(p, len, at, 4); b.reals.resize(m); if (m) memcpy(b.reals.data(), take(p, len, at, 4 * m), 4 * m); }
            if (flags & 2) { b.has_pixels = true; const size_t m = get_count(p, len, at, 1); b.pixels.resize(m); if (m) memcpy(b.pixels.data(), take(p, len, at, m), m); }
            if (flags & 4) { b.has_id = true; const size_t m = get_count(p, len, at, 4); b.id.resize(m); if (m) memcpy(b.id.data(), take(p, len, at, 4 * m), 4 * m); }
            if (flags & 8) { b.has_str = true; b.str = get_str(p, len, at); }
            data_[key] = b;
        }
        if (at != len) throw Error("StatePacket::decode: trailing bytes");
    }
    void decode(const std::vector<uint8_t> &v) { decode(v.data(), v.size()); }

  private:
    static void put(std::vector<uint8_t> &o, const void *d, size_t n) {
        const uint8_t *b = static_cast<const uint8_t *>(d);
        o.insert(o.end(), b, b + n);
    }
    static void put_u64(std::vector<uint8_t> &o, uint64_t v) { put(o, &v, 8); }
    static void put_str(std::vector<uint8_t> &o, const std::string &s) { put_u64(o, s.size()); put(o, s.c_str(), s.size() + 1); }
    static const uint8_t *take(const uint8_t *p, size_t len, size_t &at, size_t n) {
        if (at > len || n > len - at) throw Error("StatePacket::decode: truncated buffer");
        const uint8_t *q = p + at;
        at += n;
        return q;
    }
    static uint64_t get_u64(const uint8_t *p, size_t len, size_t &at) { uint64_t v; memcpy(&v, take(p, len, at, 8), 8); return v; }
    // a count of `elem`-byte elements that must still fit into the buffer
    static size_t get_count(const uint8_t *p, size_t len, size_t &at, size_t elem) {
        const uint64_t m = get_u64(p, len, at);
        if (m > (uint64_t)(len - at) / elem) throw Error("StatePacket::decode: element count exceeds the buffer");
        return (size_t)m;
    }
    static std::string get_str(const uint8_t *p, size_t len, size_t &at) {
        const size_t n = get_count(p, len, at, 1);
        if (n == len - at) throw Error("StatePacket::decode: truncated buffer");     // the terminating NUL
        const uint8_t *q = take(p, len, at, n + 1);
        return std::string(reinterpret_cast<const char *>(q), n);
    }
    std::map<std::string, StateBuffer> data_;      // the reference uses an unordered_map: wire key order is unspecified
};

// ------------------------------------------------------------- BatchedSimulator ----
class BatchedSimulator {
  public:
    explicit BatchedSimulator(const xwb_config &cfg) { check(xwb_create(&cfg, &sim_)); check(xwb_num_envs(sim_, &n_)); }
    ~BatchedSimulator() { xwb_destroy(sim_); }
    BatchedSimulator(const BatchedSimulator &) = delete;
    BatchedSimulator &operator=(const BatchedSimulator &) = delete;

    xwb_sim *handle() const { return sim_; }
    int num_envs() const { return n_; }
    void reset(void *stream = nullptr) { check(xwb_reset(sim_, stream)); }
    void reset_done(void *stream = nullptr) { check(xwb_reset_done(sim_, stream)); }
    void step(const int32_t *actions_dev, int act_rep = 1, void *stream = nullptr) { check(xwb_step(sim_, actions_dev, act_rep, stream)); }
    void step_host(const std::vector<int32_t> &actions, int act_rep = 1, void *stream = nullptr) {
        if ((int)actions.size() != n_) throw Error("step_host: need one action per env");
        check(xwb_step_host(sim_, actions.data(), act_rep, stream));
    }
    // xworld: the strings behind the palette's name ids (goal names by id; per icon its name and colour, "na" = none): with
    // them get_state()'s "sentence" is the teacher's sentence, without them "-"
    void set_names(const std::vector<std::string> &goal_names, const std::vector<std::string> &icon_names, const std::vector<std::string> &icon_colors) {
        if (icon_names.size() != icon_colors.size()) throw Error("set_names: one colour per icon name");
        std::vector<const char *> g, n, c;
        for (const std::string &x : goal_names) g.push_back(x.c_str());
        for (const std::string &x : icon_names) n.push_back(x.c_str());
        for (const std::string &x : icon_colors) c.push_back(x.c_str());
        check(xwb_set_names(sim_, g.data(), (int32_t)g.size(), n.data(), c.data(), (int32_t)n.size()));
    }
    xwb_env_state env_state(int env, void *stream = nullptr) const { xwb_env_state s; check(xwb_get_env_state(sim_, env, stream, &s)); return s; }

  private:
    xwb_sim *sim_ = nullptr;
    int32_t n_ = 0;
};

// ----------------------------------------------------------- SimulatorInterface ----
class SimulatorInterface {
  public:
    SimulatorInterface(std::shared_ptr<BatchedSimulator> batch, int env) : batch_(std::move(batch)), env_(env) {
        if (env < 0 || env >= batch_->num_envs()) throw Error("SimulatorInterface: env out of range");
    }
    virtual ~SimulatorInterface() {}

    virtual void start() { running_ = true; }
    virtual void stop() { running_ = false; }

    virtual void reset_game() { check(xwb_reset_env(batch_->handle(), env_, nullptr)); }
    virtual int game_over() { return batch_->env_state(env_).game_over; }
    virtual std::string game_over_string() {
        char buf[64];
        check(xwb_decode_game_over_code(game_over(), buf, sizeof buf));
        return buf;
    }
    virtual int get_num_actions() { int32_t n; check(xwb_get_num_actions(batch_->handle(), &n)); return n; }
    virtual int get_lives() { return batch_->env_state(env_).lives; }
    virtual int64_t get_num_steps() { return batch_->env_state(env_).num_steps; }
    virtual void get_screen_out_dimensions(size_t &height, size_t &width, size_t &channels) {
        check(xwb_get_screen_out_dimensions(batch_->handle(), &height, &width, &channels));
    }
    virtual float take_actions(const StatePacket &actions, int act_rep, bool /*show_screen*/) {
        // SimpleGame / SimpleRace: CHECK_EQ(actions.size(), 1) and key "action" (simple_game_simulator.cpp:97-98)
        const StateBuffer &b = actions.get_buffer("action");
        if (!b.has_id || b.id.empty()) throw Error("take_actions: 'action' needs an id");
        std::vector<int32_t> a((size_t)batch_->num_envs(), XWB_ACTION_SKIP);
        a[(size_t)env_] = b.id[0];
        batch_->step_host(a, act_rep);
        int32_t bad = 0;
        check(xwb_check_errors(batch_->handle(), nullptr, &bad));
        if (bad) throw Error("take_actions: action id out of range");            // reference: CHECK_LT -> abort
        return batch_->env_state(env_).reward;
    }
    float take_action(const StatePacket &actions, bool show_screen) { return take_actions(actions, 1, show_screen); }
    virtual StatePacket get_state(const float reward) {
        size_t need = 0;
        check(xwb_get_state_packet(batch_->handle(), env_, reward, nullptr, nullptr, 0, &need));
        std::vector<uint8_t> buf(need);
        check(xwb_get_state_packet(batch_->handle(), env_, reward, nullptr, buf.data(), buf.size(), &need));
        StatePacket p;
        p.decode(buf);
        return p;
    }
    virtual void get_extra_info(std::string &info) {
        char buf[256];
        check(xwb_get_extra_info(batch_->handle(), env_, nullptr, buf, sizeof buf));
        info = buf;
    }
    // simulator_interface.cpp:149-153 -> Teacher::report_task_performance (teacher.cpp:175-200): the reference logs the table
    // (LOG(INFO)); here it goes to stderr, and task_performance_report() hands the same text back.  The numbers are the
    // batch's (every env slot), not this slot's alone.  No-op for games without a teacher.
    virtual std::string task_performance_report() {
        size_t need = 0;
        if (xwb_task_performance_report(batch_->handle(), nullptr, nullptr, 0, &need) != XWB_OK) return std::string();
        std::string text(need, '\0');
        check(xwb_task_performance_report(batch_->handle(), nullptr, &text[0], text.size(), &need));
        text.resize(need ? need - 1 : 0);
        return text;
    }
    virtual void teacher_report_task_performance() {
        const std::string text = task_performance_report();
        if (!text.empty()) fputs(text.c_str(), stderr);
    }
    virtual bool last_action_success() { return batch_->env_state(env_).last_action_success != 0; }
    virtual std::string last_action() {
        const int a = batch_->env_state(env_).last_action;
        return a < 0 ? std::string("") : std::to_string(a);
    }
    virtual void get_world_dimensions(double &X, double &Y, double &Z) {
        X = Y = Z = 0;
        xwb_get_world_dimensions(batch_->handle(), &X, &Y, &Z);        // only teaching environments answer
    }

  protected:
    std::shared_ptr<BatchedSimulator> batch_;
    int env_;
    bool running_ = false;
};

typedef std::shared_ptr<SimulatorInterface> SimInterfacePtr;

}  // namespace xwb

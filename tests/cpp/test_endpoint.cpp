// include/xwb_endpoint.hpp.  Usage: test_endpoint wire | game
//   wire : host only.  xwb::RemoteSimulator (the SimulatorServer side) against a scripted peer that checks the
//          raw bytes of every request (simulator_communication.h / memory_util.h layout) and answers by hand.
//   game : cuda:0.  The trainer side drives slot 1 of a 3-env SimpleGame batch and slot 0 of an XWorld2D batch
//          through xwb::SlotClient over localhost TCP: the reference's SimpleGame known-answer test
//          (tests/test_simple_game_simulator.cpp:21-47) and the example loop's verbs, all through the wire.
#include "../../include/xwb_endpoint.hpp"

#include <cmath>
#include <cstdio>
#include <thread>

#define EXPECT(c) do { if (!(c)) { printf("FAILED: %s (%s:%d)\n", #c, __FILE__, __LINE__); return 1; } } while (0)

static std::vector<uint8_t> bytes_of(const xwb::wire::Message &m) { return std::vector<uint8_t>(m.data(), m.data() + m.size()); }

static int wire_tests() {
    xwb::RemoteSimulator server("simple_game");
    const int port = server.port();
    int peer_rc = 0;
    std::thread peer([&] {
        auto fail = [&](int line) { if (!peer_rc) peer_rc = line; };
        xwb::wire::Socket s;
        if (!s.connect_local(port)) return fail(__LINE__);
        xwb::wire::Message m;
        m.append(std::string("simple_game"));
        // greeting: size_t 11, "simple_game", NUL
        const uint8_t want[] = {11, 0, 0, 0, 0, 0, 0, 0, 's', 'i', 'm', 'p', 'l', 'e', '_', 'g', 'a', 'm', 'e', 0};
        if (bytes_of(m) != std::vector<uint8_t>(want, want + sizeof want)) return fail(__LINE__);
        s.deliver(m);
        s.receive(m);
        std::string str;
        m.read(str);
        if (str != "accepted") return fail(__LINE__);
        // "reset"
        s.receive(m);
        const uint8_t want_reset[] = {5, 0, 0, 0, 0, 0, 0, 0, 'r', 'e', 's', 'e', 't', 0};
        if (bytes_of(m) != std::vector<uint8_t>(want_reset, want_reset + sizeof want_reset)) return fail(__LINE__);
        m.clear();
        m.append("reset"); m.append((int)2); m.append((int)0); m.append((int)1);
        m.append((size_t)1); m.append((size_t)8); m.append((size_t)1); m.append(0.0); m.append(0.0); m.append(0.0);
        if (m.size() != 14 + 3 * 4 + 3 * 8 + 3 * 8) return fail(__LINE__);
        s.deliver(m);
        // "take_actions": string, int act_rep, bool show, then the packet
        s.receive(m);
        m.read(str);
        int rep; bool show; xwb::StatePacket a;
        m.read(rep); m.read(show); m.read(a);
        if (str != "take_actions" || rep != 3 || show || a.get_buffer("action").id != std::vector<int32_t>{1}) return fail(__LINE__);
        m.clear();
        m.append("take_actions"); m.append(-0.5f); m.append((int64_t)1); m.append((int)0); m.append((int)1); m.append(true);
        m.append(std::string("1"));
        s.deliver(m);
        // "get_state"
        s.receive(m);
        m.read(str);
        float r;
        m.read(r);
        if (str != "get_state" || r != -0.5f) return fail(__LINE__);
        xwb::StatePacket st;
        st.add_buffer_value("reward", std::vector<float>{r});
        st.add_buffer_value("screen", std::vector<uint8_t>{0, 0, 1, 0});
        m.clear();
        m.append("get_state"); m.append(st);
        s.deliver(m);
        s.receive(m);
        m.read(str);
        if (str != "stop") return fail(__LINE__);
    });
    EXPECT(server.start());
    server.reset_game();
    size_t h, w, c;
    server.get_screen_out_dimensions(h, w, c);
    EXPECT(server.get_num_actions() == 2 && server.game_over() == 0 && server.get_lives() == 1 && h == 1 && w == 8 && c == 1);
    xwb::StatePacket a;
    a.add_buffer_id("action", {1});
    EXPECT(server.take_actions(a, 3, false) == -0.5f && server.get_num_steps() == 1 && server.last_action() == "1");
    xwb::StatePacket st = server.get_state(-0.5f);
    EXPECT(st.get_buffer("screen").pixels == (std::vector<uint8_t>{0, 0, 1, 0}) && st.get_buffer("reward").reals[0] == -0.5f);
    server.stop();
    peer.join();
    if (peer_rc) { printf("FAILED: scripted peer, line %d\n", peer_rc); return 1; }
    printf("wire ok\n");
    return 0;
}

static int game_tests() {
    {   // SimpleGame: tests/test_simple_game_simulator.cpp:21-47 through the wire, slot 1 of 3
        xwb_config cfg;
        xwb::check(xwb_default_config(XWB_SIMPLE_GAME, &cfg));
        cfg.array_size = 8; cfg.num_envs = 3;
        auto batch = std::make_shared<xwb::BatchedSimulator>(cfg);
        xwb::RemoteSimulator server("simple_game");
        xwb::SlotClient client(batch, 1, "simple_game", server.port());
        bool client_ok = false;
        std::thread th([&] { client_ok = client.start(); });
        EXPECT(server.start());
        server.reset_game();
        size_t h, w, c;
        server.get_screen_out_dimensions(h, w, c);
        EXPECT(server.get_num_actions() == 2 && server.game_over() == 0 && server.get_lives() == 1);
        EXPECT(h == 1 && w == 8 && c == 1);
        xwb::StatePacket a;
        a.add_buffer_id("action", {1});
        int pos = 4;
        for (int i = 0; i < 3; ++i) {
            xwb::StatePacket st = server.get_state(0);
            const std::vector<uint8_t> &px = st.get_buffer("screen").pixels;
            int sum = 0;
            for (auto v : px) sum += v;
            EXPECT(px.size() == 8 && sum == 1 && px[pos] == 1);
            const float r = server.take_actions(a, 1, false);
            pos += 1;
            EXPECT(std::fabs(r - (pos == 7 ? 2.0f : -0.1f)) < 1e-6f);
            EXPECT(server.get_num_steps() == i + 1 && server.last_action_success() && server.last_action() == "1");
        }
        EXPECT(server.game_over() != 0 && server.get_lives() == 0);
        std::string info = "x";
        server.get_extra_info(info);
        EXPECT(info == "");
        server.teacher_report_task_performance();
        server.reset_game();
        EXPECT(server.game_over() == 0 && server.get_num_steps() == 0);
        // the other slots never moved
        EXPECT(batch->env_state(0).num_steps == 0 && batch->env_state(2).num_steps == 0 && batch->env_state(0).sg_pos == 4);
        server.stop();
        th.join();
        EXPECT(client_ok);
    }
    printf("game ok\n");
    return 0;
}

// The reference's own gtest for the wire container, tests/test_binary_buffer.cpp:151-246 (TEST(BinaryBuffer, read_write)),
// run on xwb::wire::Message: vectors carry their element count, arrays do not, strings carry length and a NUL,
// buffers concatenate, insert() splices raw elements.
static int buffer_tests() {
    using xwb::wire::Message;
    Message b1, b2, b3, b5;
    std::vector<int> v({1, 2, 3, 4});
    float f[3] = {4, 5, 6};
    float ff[6];
    {
        b1.append(v);
        b1.append(std::vector<float>(0));
        b1.rewind();
        std::vector<int> v1, v2;
        b1.read(v1);
        b1.read(v2);
        EXPECT(v1.size() == v.size() && v1[0] == 1 && v1[1] == 2 && v1[2] == 3 && v1[3] == 4 && v2.size() == 0);
    }
    {
        b2.append(f, 3);
        b2.append(f, 3);
        b2.rewind();
        b2.read(ff, 6);
        EXPECT(b2.size() == 6 * sizeof(float));
        for (int i = 0; i < 6; ++i) EXPECT(ff[i] == f[i % 3]);
    }
    {
        std::string str("789");
        b3.append(str);
        b3.append(std::string(""));
        std::string tmp;
        b3.rewind();
        b3.read(tmp);
        EXPECT(tmp.length() == 3 && tmp == str);
        b3.read(tmp);
        EXPECT(tmp.length() == 0 && tmp == "");
        EXPECT(b3.size() == 2 * sizeof(size_t) + 3 + 1 + 1);
    }
    {
        b1.append(b2);
        b1.rewind();
        size_t sz;
        b1.read(sz);
        EXPECT(sz == v.size());
        int x;
        for (size_t i = 0; i < sz; ++i) { b1.read(x); EXPECT(x == v[i]); }
        b1.read(sz);
        EXPECT(sz == 0);
        float g;
        for (int i = 0; i < 3; ++i) { b1.read(g); EXPECT(g == ff[i]); }
        b1.read(g);
        EXPECT(!b1.eof());
    }
    {
        std::string s("a");
        b5.append(int(1));
        b5.append(s);
        std::vector<int> a({2, 3});
        b5.insert(sizeof(int), a.data(), a.size());
        b5.rewind();
        for (int i = 1; i <= 3; ++i) { int w; b5.read(w); EXPECT(w == i); }
        b5.read(s);
        EXPECT(s == "a" && b5.eof());
    }
    bool threw = false;
    try { int w; b5.read(w); } catch (const std::exception &) { threw = true; }   // BinaryBuffer::read CHECK_LE
    EXPECT(threw);
    printf("buffer ok\n");
    return 0;
}

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "wire";
    try {
        return mode == "game" ? game_tests() : (mode == "buffer" ? buffer_tests() : wire_tests());
    } catch (const std::exception &e) {
        printf("FAILED: exception %s\n", e.what());
        return 1;
    }
}

// C++ surface test (include/xwb_simulator.hpp).  Usage: test_cpp_interface packet | game
//   packet : host-only StatePacket checks, incl. the reference's serialization test
//            (tests/test_statepacket.cpp:77-104) -- runs without a GPU
//   game   : SimpleGame through xwb::SimulatorInterface on cuda:0: the reference's own known-answer test
//            (tests/test_simple_game_simulator.cpp:21-47) on slot 2 of a 4-env batch, the other slots untouched
#include "../../include/xwb_simulator.hpp"

#include <cmath>
#include <cstdio>

#define EXPECT(c) do { if (!(c)) { printf("FAILED: %s (%s:%d)\n", #c, __FILE__, __LINE__); return 1; } } while (0)

static int packet_tests() {
    xwb::StatePacket s1;
    s1.add_key("screen");
    s1.add_key("internal_state");
    s1.add_buffer_value("screen", std::vector<uint8_t>{1, 2, 3, 4});
    s1.add_buffer_id("screen", {10, 11});
    s1.add_buffer_value("internal_state", std::vector<float>{1.5f, 2.5f, 3.5f, 4.5f, 5.5f, 6.5f});
    s1.add_buffer_str("internal_state", "abc");
    std::vector<uint8_t> buf = s1.encode();
    xwb::StatePacket s2;
    s2.decode(buf);
    EXPECT(s2.contain_key("screen") && s2.contain_key("internal_state") && s2.size() == 2);
    EXPECT(s1.get_buffer("screen") == s2.get_buffer("screen"));
    EXPECT(s1.get_buffer("internal_state") == s2.get_buffer("internal_state"));
    // layout: u64 nkeys | u64 len, key, NUL | u8 flags | ...   (keys sorted here: "internal_state" first)
    uint64_t nkeys; memcpy(&nkeys, buf.data(), 8);
    EXPECT(nkeys == 2);
    uint64_t klen; memcpy(&klen, buf.data() + 8, 8);
    EXPECT(klen == 14 && memcmp(buf.data() + 16, "internal_state\0", 15) == 0 && buf[31] == (1 | 8));
    bool threw = false;
    try { xwb::StatePacket s3; s3.decode(buf.data(), buf.size() - 1); } catch (const xwb::Error &) { threw = true; }
    EXPECT(threw);
    // hostile counts: an element count or a string length far beyond the bytes that follow must be refused before anything
    // is sized by it (and 4 * m / n + 1 must not wrap)
    const uint64_t bad[] = {~0ull, ~0ull / 4 + 1, 1ull << 62, (uint64_t)buf.size()};
    for (uint64_t v : bad) {
        std::vector<uint8_t> evil = buf;
        memcpy(evil.data() + 32, &v, 8);                   // the reals count of "internal_state"
        threw = false;
        try { xwb::StatePacket s3; s3.decode(evil); } catch (const xwb::Error &) { threw = true; } catch (...) { threw = false; }
        EXPECT(threw);
        evil = buf;
        memcpy(evil.data() + 8, &v, 8);                    // the first key's length
        threw = false;
        try { xwb::StatePacket s3; s3.decode(evil); } catch (const xwb::Error &) { threw = true; } catch (...) { threw = false; }
        EXPECT(threw);
    }
    printf("packet ok (%zu bytes)\n", buf.size());
    return 0;
}

static int game_tests() {
    xwb_config cfg;
    xwb::check(xwb_default_config(XWB_SIMPLE_GAME, &cfg));
    cfg.array_size = 8;
    cfg.num_envs = 4;
    auto batch = std::make_shared<xwb::BatchedSimulator>(cfg);
    xwb::SimulatorInterface game(batch, 2), other(batch, 0);
    game.reset_game();
    size_t h, w, c;
    game.get_screen_out_dimensions(h, w, c);
    EXPECT(h == 1 && w == 8 && c == 1 && game.get_num_actions() == 2);
    int pos = 4;
    for (int i = 0; i < 3; ++i) {
        xwb::StatePacket st = game.get_state(0);
        const auto &scr = st.get_buffer("screen").pixels;
        EXPECT(scr.size() == 8);
        for (int j = 0; j < 8; ++j) EXPECT(int(scr[j]) == (j == pos ? 1 : 0));
        xwb::StatePacket a;
        a.add_buffer_id("action", {1});
        float reward = game.take_action(a, false);
        pos++;
        EXPECT(std::fabs(reward - (pos != 7 ? -0.1f : 2.0f)) < 1e-6);
    }
    EXPECT(game.game_over() == XWB_SUCCESS && game.game_over_string() == "success" && game.get_lives() == 0);
    EXPECT(game.get_num_steps() == 3 && game.last_action() == "1");
    EXPECT(other.get_num_steps() == 0 && other.game_over() == XWB_ALIVE);     // slot 0 never stepped
    bool threw = false;
    try { xwb::StatePacket a; a.add_buffer_id("action", {5}); game.take_action(a, false); } catch (const xwb::Error &) { threw = true; }
    EXPECT(threw);
    game.reset_game();
    EXPECT(game.game_over() == XWB_ALIVE && game.get_num_steps() == 0);
    printf("game ok\n");
    return 0;
}

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "packet";
    try {
        return mode == "game" ? game_tests() : packet_tests();
    } catch (const std::exception &e) {
        printf("exception: %s\n", e.what());
        return 2;
    }
}

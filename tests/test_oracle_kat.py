"""Pins the CPU oracle to the reference's own known-answer tests (runs on CPU).

  tests/test_simple_game_simulator.cpp:21-47   SimpleGame state / action / reward
  tests/test_simulator_seed.cpp:22-50          thread-local RNG sequences for FLAGS_simulator_seed 1 and 2
  tests/test_statepacket.cpp:77-104            StatePacket serialisation round trip
plus the known-answer values recorded in SURVEY.md 8(a) for SimpleGame contexts and SimpleRace
(the reference has no SimpleRace test and its C++ cannot be built here; see oracle/oracle.h).
"""
import ctypes as C

import numpy as np


def test_simple_game_reference_gtest(oracle):
    # FLAGS_array_size = 8; action 1 three times -> -0.1, -0.1, 2.0 ; screen one-hot at pos
    g = oracle.SimpleGame(8)
    pos = 4
    for i in range((8 - 1) // 2):
        scr = g.screen()
        assert scr.sum() == 1 and scr[pos] == 1
        r = g.take_actions(1)
        pos += 1
        assert abs(r - (2.0 if pos == 7 else -0.1)) < 1e-6


def test_simple_game_survey_kats(oracle):
    g = oracle.SimpleGame(16, context=2)
    g.reset_game()
    r = [g.take_actions(0) for _ in range(8)]
    assert np.allclose(r[:7], -0.1) and r[7] == 4.0
    assert g.game_over() == 4 and g.num_steps() == 8
    assert g.state_screen().tolist() == [0, 1] + [0] * 14 + [1] + [0] * 15     # older frame first
    assert np.float32(g.take_actions(0)) == np.float32(-0.1) and g.pos() == 0  # stepping past the end
    g = oracle.SimpleGame(16)
    g.reset_game()
    r = [g.take_actions(1) for _ in range(7)]
    assert np.allclose(r[:6], -0.1) and r[6] == 2.0 and g.num_steps() == 7


def test_simple_game_edge_sizes(oracle):
    g = oracle.SimpleGame(1)          # rewards[N-1] = 2 then rewards[0] = 4 on the same cell
    g.reset_game()
    assert g.game_over() == 4 and g.take_actions(0) == 4.0 and np.float32(g.take_actions(1)) == np.float32(-0.1)
    g = oracle.SimpleGame(2)          # starts on the right end: over at once, reward 2.0 still unclaimed
    g.reset_game()
    assert g.pos() == 1 and g.game_over() == 4 and g.take_actions(0) == 2.0
    g = oracle.SimpleGame(6, max_steps=3)
    g.reset_game()
    for _ in range(3):
        g.take_actions(0 if g.pos() > 3 else 1)
    assert g.game_over() & 1          # MAX_STEP from the GameSimulator base


def test_rng_reference_gtest(oracle):
    L = oracle.lib()
    g = oracle.MinStd()
    seq = []
    for n in range(1, 6):             # threads 1..5 under FLAGS_simulator_seed = 1
        L.orc_minstd_seed_thread(C.byref(g), 1, n)
        seq.append(L.orc_minstd_rand_ind(C.byref(g), 1000000))
    assert seq == [266148, 605992, 817626, 635637, 393423]
    seq = []
    for n in range(6, 11):            # the global thread counter keeps counting: threads 6..10, seed 2
        L.orc_minstd_seed_thread(C.byref(g), 2, n)
        seq.append(L.orc_minstd_rand_ind(C.byref(g), 1000000))
    assert seq == [258945, 847424, 238883, 918571, 875562]


def test_rng_hash_and_ranges(oracle):
    L = oracle.lib()
    # int(std::hash<std::string>("2".."7"))
    exp = [1874210838, -1146431235, -957938776, 1447751201, -458527649, -942936030]
    got = [C.c_int32(L.orc_std_hash_string(str(k).encode(), 1) & 0xFFFFFFFF).value for k in range(2, 8)]
    assert got == exp
    g = oracle.MinStd()
    L.orc_minstd_seed(C.byref(g), 1)
    for _ in range(1000):
        v = L.orc_minstd_rand_range(C.byref(g), 1.0)
        assert 0.0 <= v < 1.0
        k = L.orc_minstd_rand_ind(C.byref(g), 7)
        assert 0 <= k < 7


def test_philox_known_answers(oracle):
    # Random123 kat_vectors: philox4x32-10
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_simple_race_survey_kats_reward_unpinned_by_reference(oracle):
    r = oracle.SimpleRace()
    r.reset_game()
    exp_obs = [(1, 0, 0, -0.800000012), (0.95105654, -0.309016943, -0.0309020989, -0.780978978),
               (1, 0, -0.0309020989, -0.760978997), (0.95105648, 0.309017122, 0, -0.741958022),
               (1, 0, 0, -0.721957982), (0.95105654, -0.309016943, -0.0309020989, -0.702937007),
               (0.809017003, -0.587785244, -0.089680478, -0.686756611),
               (0.95105654, -0.309016943, -0.120582581, -0.667735577)]
    exp_rew = [0.920154452, 0.969097912, 0.95105648, 1, 0.920154452, 0.71933651, 0.830473959, 0.629656076]
    for a, eo, er in zip([0, 1, 1, 0, 0, 0, 1, 0], exp_obs, exp_rew):
        assert np.array_equal(r.screen(), np.float32(eo))
        assert np.float32(r.take_actions(a)) == np.float32(er)
    r.reset_game()
    total, k = np.float32(0), 0
    while r.game_over() == 0:
        total = np.float32(total + np.float32(r.take_actions(k % 2)))
        k += 1
    assert k == 65 and r.game_over() == 2 and total == np.float32(29.732481)
    assert [hex(x) for x in r.screen().view(np.uint32)] == ["0x3f737871", "0xbe9e3778", "0xbf8287cd", "0x3eef7571"]
    r.reset_game()
    rew = [np.float32(r.take_actions(0)) for _ in range(10)]
    exp = [0.920154452, 0.71933651, 0.417202592, 0.0433292389, -0.365687877, -0.769809961, -1.12948036,
           -1.40949047, -1.58243191, -1.63137507]
    assert np.array_equal(np.float32(rew), np.float32(exp)) and r.game_over() == 2 and r.num_steps() == 10


def test_state_packet_reference_roundtrip(oracle):
    """tests/test_statepacket.cpp:77-104: screen = {u8 [1,2,3,4], id [10,11]}, internal_state = {f32 x6, "abc"}."""
    L = oracle.lib()
    a = np.array([1, 2, 3, 4], np.uint8)
    ids = np.array([10, 11], np.int32)
    b = np.array([1.5, 2.5, 3.5, 4.5, 5.5, 6.5], np.float32)
    f = (oracle.PacketField * 2)()
    f[0] = oracle.PacketField(b"screen", 0, None, 0, 1, oracle.ptr(a, oracle.u8p), 4, 1, oracle.ptr(ids, oracle.i32p), 2, 0, None)
    f[1] = oracle.PacketField(b"internal_state", 1, oracle.ptr(b, oracle.f32p), 6, 0, None, 0, 0, None, 0, 1, b"abc")
    need = L.orc_packet_encode(f, 2, None, 0)
    buf = np.zeros(need, np.uint8)
    assert L.orc_packet_encode(f, 2, oracle.ptr(buf, oracle.u8p), need) == need
    # layout: u64 nkeys | u64 len "screen" NUL | flags 2|4 | u64 4, bytes | u64 2, ints | ...
    raw = buf.tobytes()
    assert raw[:8] == (2).to_bytes(8, "little") and raw[8:16] == (6).to_bytes(8, "little")
    assert raw[16:23] == b"screen\0" and raw[23] == 6
    out = (oracle.PacketField * 4)()
    assert L.orc_packet_decode(oracle.ptr(buf, oracle.u8p), need, out, 4) == 2
    assert out[0].key == b"screen" and out[0].has_pixels and not out[0].has_reals and out[0].has_id
    assert [out[0].pixels[i] for i in range(4)] == [1, 2, 3, 4] and [out[0].id[i] for i in range(2)] == [10, 11]
    assert out[1].key == b"internal_state" and out[1].has_str and out[1].str == b"abc"
    assert [out[1].reals[i] for i in range(6)] == [1.5, 2.5, 3.5, 4.5, 5.5, 6.5]
    # truncated buffers are rejected (BinaryBuffer::read CHECK_LE)
    assert L.orc_packet_decode(oracle.ptr(buf, oracle.u8p), need - 1, out, 4) == -1


def test_decode_game_over_code(oracle):
    assert oracle.decode_game_over_code(0) == "alive"
    assert oracle.decode_game_over_code(1) == "max_step"
    assert oracle.decode_game_over_code(2 | 4) == "dead|success"
    assert oracle.decode_game_over_code(15) == "max_step|dead|success|lost_life"


def test_oracle_ctypes_structs_match_the_header(oracle, tmp_path):
    """tests/_oracle.py mirrors oracle/oracle.h by hand: offsets and sizes against the C compiler's."""
    import ctypes as C
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = {"orc_xw_cfg": oracle.XwCfg, "orc_entity": oracle.Entity, "orc_icon_info": oracle.IconInfo}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "oracle.h"', 'int main(void) {']
    for cname, cls in structs.items():
        src.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            src.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    src += ['return 0;', '}']
    (tmp_path / "o.c").write_text("\n".join(src))
    subprocess.check_call(["gcc", "-I" + os.path.join(root, "oracle"), str(tmp_path / "o.c"), "-o", str(tmp_path / "o")])
    for line in subprocess.check_output([str(tmp_path / "o")], text=True).splitlines():
        cname, fname, off = line.split()
        cls = structs[cname]
        assert (C.sizeof(cls) if fname == "sizeof" else getattr(cls, fname).offset) == int(off), (cname, fname)

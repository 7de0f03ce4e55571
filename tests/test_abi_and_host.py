"""CPU tests: the C-ABI library loads and exports every symbol include/xwb.h declares, fails loudly
without a GPU (no CPU fallback, no oracle in the product), and the py_simulator-compatible host layer
keeps the reference's option handling and errors.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol():
    from xworld_amd import lib
    L = lib.load()
    header = open(os.path.join(ROOT, "include", "xwb.h")).read()
    declared = set(re.findall(r"\b(xwb_[a-z0-9_]+)\s*\(", header))
    declared -= {"xwb_sim", "xwb_config", "xwb_env_state"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), "libxwb.so does not export " + name
    assert declared == set(lib.EXPORTED_SYMBOLS), declared ^ set(lib.EXPORTED_SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (xwb_[a-z0-9_]+)", out))
    assert declared <= exported
    # ... the test / measurement hooks of include/xwb_testing.h under their own version node, outside the boundary ...
    hooks = set(re.findall(r"\b(xwb_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "xwb_testing.h")).read()))
    assert hooks == set(lib.TESTING_SYMBOLS) and not (hooks & declared)
    versions = subprocess.check_output(["objdump", "-T", lib.LIB_PATH]).decode()
    for ln in versions.splitlines():
        m = re.search(r"\b(XWB_1|XWB_TESTING)\s+(xwb_[a-z0-9_]+)$", ln.strip())
        if m:
            assert (m.group(1) == "XWB_TESTING") == (m.group(2) in hooks), ln
    assert "XWB_TESTING" in versions and "XWB_1" in versions
    # ... and nothing else: no kernel handles, launch helpers or other internals (-fvisibility=hidden + csrc/libxwb.map)
    everything = {ln.split()[-1].split("@")[0] for ln in out.splitlines() if ln.strip()} - {"XWB_1", "XWB_TESTING"}
    assert everything == declared | hooks, sorted(everything ^ (declared | hooks))


def test_ctypes_struct_matches_header_layout():
    """xwb_default_config writes through the C struct: field values land where the ctypes mirror reads them."""
    from xworld_amd import lib
    L = lib.load()
    for game in (0, 1, 2):
        c = lib.XwbConfig()
        assert L.xwb_default_config(game, C.byref(c)) == 0
        assert c.abi_version == lib.XWB_ABI_VERSION and c.game == game and c.num_envs == 1
        assert c.context == 1 and c.max_steps == 0 and c.array_size == 6
        assert c.track_width == 20.0 and c.track_length == 100.0 and c.track_radius == 30.0 and c.reward_scale == 1.0
        assert (c.map_kind, c.max_dim, c.dim, c.num_goals, c.num_blocks, c.max_steps_factor) == (0, 8, 8, 4, 16, 10)
    assert L.xwb_default_config(7, C.byref(c)) != 0
    assert b"unknown game" in L.xwb_last_error()


def test_decode_game_over_code_matches_reference_strings():
    from xworld_amd import lib
    assert lib.decode_game_over_code(0) == "alive"
    assert lib.decode_game_over_code(1) == "max_step"
    assert lib.decode_game_over_code(2) == "dead"
    assert lib.decode_game_over_code(4 | 1) == "max_step|success"
    assert lib.decode_game_over_code(15) == "max_step|dead|success|lost_life"
    with pytest.raises(lib.XwbError):
        lib.decode_game_over_code(16)


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_create_fails_loudly_without_gpu():
    from xworld_amd import lib
    from xworld_amd.batched import BatchedSimulator
    with pytest.raises(lib.XwbError, match="no CPU path|HIP"):
        BatchedSimulator("simple_game", {"array_size": 6})


def test_option_errors_match_py_simulator():
    """Missing required options / unknown games raise RuntimeError before anything touches the device
    (python/py_simulator.cpp:40-57,184-186)."""
    from xworld_amd.py_simulator import Simulator
    with pytest.raises(RuntimeError, match="Unrecognized game type"):
        Simulator.create("pong", {})
    with pytest.raises(RuntimeError, match="Key 'array_size' is required"):
        Simulator.create("simple_game", {})
    with pytest.raises(RuntimeError, match="Key 'track_width' is required"):
        Simulator.create("simple_race", {"track_type": "straight"})
    with pytest.raises(RuntimeError, match="Key 'xwd_conf_path' is required"):
        Simulator.create("xworld", {})
    with pytest.raises(RuntimeError):
        Simulator()


def test_palettes_and_conf():
    from xworld_amd import assets
    nav = assets.Palette(assets.MAP_CLASSES["XWorldNav"]["subtrees"])
    walls = assets.Palette(assets.MAP_CLASSES["XWorldWalls"]["subtrees"])
    assert len(nav) == 347 and len(walls) == 246          # 180+48+66+51+2 ; 180+48+16+2
    assert nav.icons64.shape == (347, 64, 64, 3)
    assert nav.names["block"] == ["brick"] and nav.names["agent"] == ["robot"]
    assert "shape" not in {m["subtree"] for m in nav.meta}
    conf = assets.read_conf(os.path.join(ROOT, "xworld_amd", "confs", "nav_target.json"))
    assert conf["map"] == "XWorldNav"


def test_product_never_touches_the_oracle():
    """The product package must not import, link or execute anything under oracle/ or tests/."""
    pkg = os.path.join(ROOT, "xworld_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "liboracle" not in src and "_oracle" not in src and "oracle/" not in src.replace("the test oracle", ""), f
    out = subprocess.check_output(["ldd", os.path.join(pkg, "libxwb.so")]).decode()
    assert "oracle" not in out


def test_ctypes_offsets_match_the_c_header(tmp_path):
    """Every field of xwb_config / xwb_env_state sits where the C compiler puts it (offsetof from include/xwb.h)."""
    import subprocess
    from xworld_amd import lib
    structs = {"xwb_config": lib.XwbConfig, "xwb_env_state": lib.XwbEnvState}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "xwb.h"', 'int main(void) {']
    for cname, cls in structs.items():
        src.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            src.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    src += ['return 0;', '}']
    c_file = tmp_path / "offsets.c"
    c_file.write_text("\n".join(src))
    exe = tmp_path / "offsets"
    subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), str(c_file), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True)
    seen = 0
    for line in out.splitlines():
        cname, fname, off = line.split()
        cls = structs[cname]
        if fname == "sizeof":
            assert C.sizeof(cls) == int(off), (cname, C.sizeof(cls), off)
        else:
            assert getattr(cls, fname).offset == int(off), (cname, fname)
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in structs.values())


def test_reference_rng_known_answers_through_libxwb(oracle):
    """The reference's tests/test_simulator_seed.cpp:22-50 through the product's own restatement (include/xwb_minstd.h,
    exported as xwb_minstd_*): FLAGS_simulator_seed = 1 -> the first get_rand_ind(1000000) of threads 1..5; the test binary's
    second case runs in the same process, so seed 2 meets threads 6..10.  Then the product's engines against the oracle's
    (an independent restatement pinned by the same test) over long mixed draw sequences."""
    from xworld_amd import lib
    L = lib.load()
    seq1 = [266148, 605992, 817626, 635637, 393423]
    seq2 = [258945, 847424, 238883, 918571, 875562]
    for seed, first, want in ((1, 1, seq1), (2, 6, seq2)):
        got = []
        for nth in range(first, first + 5):
            st = C.c_uint32(L.xwb_minstd_seed_thread(seed, nth))
            got.append(L.xwb_minstd_rand_ind(C.byref(st), 1000000))
        assert got == want, (seed, got)
    OL = oracle.lib()
    for seed, nth in ((1, 1), (2, 9), (12345, 77), (-3, 4)):
        st = C.c_uint32(L.xwb_minstd_seed_thread(seed, nth))
        g = oracle.MinStd()
        OL.orc_minstd_seed_thread(C.byref(g), seed, nth)
        assert st.value == g.x
        for k in range(3000):
            if k % 3 == 0:
                n = 1 + (k * 7919) % 1000
                assert L.xwb_minstd_rand_ind(C.byref(st), n) == OL.orc_minstd_rand_ind(C.byref(g), n)
            else:
                u = float(1 + k % 5)
                a, b = L.xwb_minstd_rand_range(C.byref(st), u), OL.orc_minstd_rand_range(C.byref(g), u)
                assert a == b and 0 <= a <= u
            assert st.value == g.x


def test_rccl_is_resolved_at_run_time_and_argument_errors():
    """xwb_comm_* / xwb_gather_*: libxwb.so has no link-time RCCL dependency (the exchange resolves the librccl.so.1 of the
    process when it is first used); bad arguments are refused before anything reaches RCCL or the device."""
    from xworld_amd import lib
    L = lib.load()
    out = subprocess.check_output(["readelf", "-d", lib.LIB_PATH]).decode()
    assert "rccl" not in out.lower()
    v = C.c_int32()
    assert L.xwb_comm_version(C.byref(v)) == 0 and v.value >= 20000, L.xwb_last_error()
    assert L.xwb_comm_version(None) != 0
    assert L.xwb_comm_unique_id(None) != 0
    h = C.c_void_p()
    ident = (C.c_uint8 * 128)()
    assert L.xwb_comm_init_rank(ident, 2, 2, 0, C.byref(h)) != 0 and b"rank" in L.xwb_last_error()
    assert L.xwb_comm_adopt(None, 0, C.byref(h)) != 0
    assert L.xwb_gather_results(None, None, None, None, None, 1, 0, None) != 0
    assert L.xwb_gather_screens_end(None, None) != 0
    assert L.xwb_comm_destroy(None) == 0

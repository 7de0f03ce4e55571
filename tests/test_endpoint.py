"""include/xwb_endpoint.hpp: the reference's SimulatorServer / SimulatorClient localhost RPC for one env slot."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_endpoint.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "test_endpoint")


def _build():
    from xworld_amd import lib
    lib.load()                                            # makes sure libxwb.so exists
    deps = [SRC] + [os.path.join(ROOT, "include", f) for f in ("xwb_endpoint.hpp", "xwb_simulator.hpp", "xwb.h")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", SRC, "-o", EXE, "-L" + os.path.join(ROOT, "xworld_amd"),
                           "-lxwb", "-lpthread", "-Wl,-rpath,$ORIGIN/../../xworld_amd"])


def test_rpc_wire_layout_against_scripted_peer():
    """Message framing and field order of every verb, byte-checked by a scripted peer (host only)."""
    _build()
    out = subprocess.run([EXE, "wire"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "wire ok" in out.stdout, out.stdout + out.stderr


def test_wire_container_against_reference_gtest():
    """tests/test_binary_buffer.cpp TEST(BinaryBuffer, read_write) restated on xwb::wire::Message (host only)."""
    _build()
    out = subprocess.run([EXE, "buffer"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "buffer ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_slot_served_over_tcp():
    """A trainer-side SimulatorServer drives one slot of a batch through xwb::SlotClient: the reference's SimpleGame
    known-answer test over the wire."""
    _build()
    out = subprocess.run([EXE, "game"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "game ok" in out.stdout, out.stdout + out.stderr

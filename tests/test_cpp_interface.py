"""include/xwb_simulator.hpp: the C++ SimulatorInterface / StatePacket surface over the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_cpp_interface.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "test_cpp_interface")


def _build():
    from xworld_amd import lib
    lib.load()                                            # makes sure libxwb.so exists
    deps = [SRC, os.path.join(ROOT, "include", "xwb_simulator.hpp"), os.path.join(ROOT, "include", "xwb.h")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", SRC, "-o", EXE, "-L" + os.path.join(ROOT, "xworld_amd"),
                           "-lxwb", "-Wl,-rpath,$ORIGIN/../../xworld_amd"])


def test_state_packet_wire_roundtrip_cpp():
    """tests/test_statepacket.cpp:77-104 through xwb::StatePacket (host only)."""
    _build()
    out = subprocess.run([EXE, "packet"], capture_output=True, text=True)
    assert out.returncode == 0 and "packet ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_simulator_interface_view_cpp():
    """tests/test_simple_game_simulator.cpp:21-47 through xwb::SimulatorInterface bound to one slot of a batch."""
    _build()
    out = subprocess.run([EXE, "game"], capture_output=True, text=True)
    assert out.returncode == 0 and "game ok" in out.stdout, out.stdout + out.stderr

"""`from py_simulator import Simulator` -- the import line of the reference's Python callers
(python/examples/test_simple_game.py:3, test_simple_race.py:3, test_xworld.py:3; module defined at
python/py_simulator.cpp:310-329).  The implementation is xworld_amd.py_simulator (HIP path, no CPU fallback)."""
from xworld_amd.py_simulator import Simulator, help  # noqa: F401

__all__ = ["Simulator", "help"]

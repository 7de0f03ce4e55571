"""xworld_amd/language.py against the reference's CFG + task grammars (tests/golden/sentences.json).  CPU only."""
import json
import os

import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TASKS = {"XWorld3DNavTarget": 0, "XWorld3DNavTargetNear": 1, "XWorld3DNavTargetBetween": 2, "XWorld3DNavTargetDirection": 3,
         "XWorld3DNavTargetAvoid": 4, "XWorldNavTarget": 5, "XWorldNavColorTarget": 7}


@pytest.mark.parametrize("name", sorted(TASKS))
def test_sentences_match_reference_cfg(name):
    from xworld_amd import language
    with open(os.path.join(GOLD, "sentences.json")) as f:
        gold = json.load(f)
    kinds = set()
    for rec in gold["tasks"][name]:
        decisions = list(rec["decisions"])

        def choose(n, _d=decisions):
            v = _d.pop(0)
            assert 0 <= v < n
            return v
        got = language.GRAMMARS[TASKS[name]].expand(choose, dict(rec["bind"]))
        assert got == rec["sentence"], rec
        assert not decisions                                   # the same number of random.choice calls
        kinds.add(rec["bind"]["S"])
    assert "start" in kinds and len(kinds) >= 2


def test_philox_stream_matches_the_oracle(oracle):
    from xworld_amd import language
    for seed, gid, ep in ((0xC0FFEE, 0, 0), (1, 77, 3), (0xFFFFFFFF, 123456, 9)):
        import ctypes as C
        a = language.Stream(seed, gid, ep, 3)
        b = oracle.Stream()
        oracle.lib().orc_stream_init(C.byref(b), seed, gid, ep, 3)
        for n in (7, 5, 3, 4, 1, 2, 115, 6, 2):
            assert a.below(n) == oracle.lib().orc_stream_below(C.byref(b), n)

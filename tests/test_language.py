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


def test_library_sentences_equal_language_py():
    """xworld_amd/csrc/xwb_language.h (what xwb_sentence and the state packets use) against language.py, which the tests above pin
    to the reference's CFG: the same grammars, the same stream, the same draws -- sentence for sentence, no GPU needed."""
    import ctypes as C
    import random
    from xworld_amd import language, lib
    L = lib.load()
    names = ["apple", "avocado", "banana", "blueberry", "cabbage", "cauliflower", "cherry", "coconut", "cucumber", "fig"]
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])

    def c3(task, stage, event, a, b, direction, seed, gid, ep):
        need = C.c_size_t()
        lib.check(L.xwb_language_sentence(task, stage, event, arr, len(names), a, b, direction, seed, gid, ep, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        lib.check(L.xwb_language_sentence(task, stage, event, arr, len(names), a, b, direction, seed, gid, ep, buf, need.value, C.byref(need)))
        return buf.value.decode()

    def c2(task, timeup, name, color, seed, gid, ep, steps):
        need = C.c_size_t()
        lib.check(L.xwb_language_sentence_2d(task, timeup, name.encode(), color.encode(), seed, gid, ep, steps, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        lib.check(L.xwb_language_sentence_2d(task, timeup, name.encode(), color.encode(), seed, gid, ep, steps, buf, need.value, C.byref(need)))
        return buf.value.decode()

    rng = random.Random(7)
    seen = set()
    for i in range(3000):
        task = rng.randrange(5)
        seed, gid, ep = rng.getrandbits(32), rng.getrandbits(20), rng.randrange(500)
        a, b = rng.randrange(len(names)), rng.randrange(len(names))
        direction = rng.randrange(1, 5)
        want = language.sentence(task, 1, 0, names, a, b if task == 2 else 0xFFFF, direction if task == 3 else 0, seed, gid, ep)
        got = c3(task, 1, 0, a, b if task == 2 else 0xFFFF, direction if task == 3 else 0, seed, gid, ep)
        assert got == want != "", (task, seed, gid, ep)
        seen.add(got.split()[0])
    assert len(seen) >= 8                                  # the grammars' variety shows
    for task in range(5):
        for event in (1, 2, 3):
            assert c3(task, 2, event, 0xFFFF, 0xFFFF, 0, 1, 2, 3) == language.sentence(task, 2, event, names, 0xFFFF, 0xFFFF, 0, 1, 2, 3)
        assert c3(task, 2, 0, 0, 0xFFFF, 0, 1, 2, 3) == "" == c3(task, 1, 0, 0xFFFF, 0xFFFF, 0, 1, 2, 3)      # silent: terminal stage / nothing bound
    assert c3(6, 1, 0, 0, 0xFFFF, 0, 1, 2, 3) == ""        # a task without a grammar
    for i in range(1500):
        task = rng.choice((5, 7))
        seed, gid, ep, steps = rng.getrandbits(32), rng.getrandbits(20), rng.randrange(500), rng.randrange(60)
        name, color = rng.choice(names), rng.choice(["red", "green", "blue", "yellow"])
        assert c2(task, 0, name, color, seed, gid, ep, steps) == language.sentence_2d(task, name, color, seed, gid, ep, steps) != ""
    assert c2(5, 1, "", "", 0, 0, 0, 0) == language.sentence_2d_timeup(5) == "Time up ."

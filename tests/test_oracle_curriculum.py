"""FLAGS_curriculum != 0 (XWorldNav.py:36-53; xworld_env.py:103-110; xworld3d_task.py:129-146): the oracle's level logic
against tests/golden/curriculum.json -- one reference XWorldNav env and its task objects driven through 920 resets."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fixture():
    with open(os.path.join(GOLDEN, "curriculum.json")) as f:
        return json.load(f)


def test_reference_records_follow_the_events():
    """What the restated rule relies on: a task records exactly one result per correct_goal (1), wrong_goal (0) and
    time_up (0) event, and nothing else."""
    fx = _fixture()
    seen = set()
    for ep in fx["episodes"]:
        for event, rec in ep["events"]:
            assert (event, rec) in {("correct_goal", 1), ("wrong_goal", 0), ("time_up", 0)}, (event, rec)
            seen.add(event)
        assert [r for _, r in ep["events"]] == ep["records"]
    assert seen == {"correct_goal", "wrong_goal", "time_up"}


def test_level_logic_matches_reference(oracle):
    fx = _fixture()
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    w = oracle.XWorld(pal, render=False, map_kind=0, max_dim=8, dim=8, curriculum=fx["threshold"], tasks=fx["tasks"])
    levels = set()
    stalled = False
    for i, ep in enumerate(fx["episodes"]):
        lv, dim, goals, blocks = w.curriculum_configure()
        assert (lv, dim, goals, blocks) == (ep["level"], ep["dim"], ep["num_goals"], ep["num_blocks"]), i
        assert w.curriculum_state() == (ep["level"], ep["counter"]), i
        if ep["counter"] == 0 and i and fx["episodes"][i - 1]["level"] == lv and lv < 5:
            stalled = True                                   # a check that found the success rate too low
        levels.add(lv)
        for r in ep["records"]:
            w.record_result(ep["task"], r)
    assert levels == {0, 1, 2, 3, 4, 5} and stalled


def test_oracle_episodes_follow_the_level(oracle):
    """The oracle's own resets under a curriculum: dims, goals and blocks of every episode are the level's; the 3-D tasks'
    time-up uses the actual dims (xworld3d_task.py:472-482); XWorldWalls ignores the flag."""
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    goals_seq, blocks_seq = [2, 2, 2, 4, 4, 4], [0, 3, 6, 9, 12, 16]
    w = oracle.XWorld(pal, render=False, map_kind=0, max_dim=8, dim=8, curriculum=0.02, seed=77,
                      tasks=["XWorld3DNavTarget", "XWorld3DNavTargetAvoid"])
    rng = np.random.default_rng(5)
    seen = set()
    for ep in range(620):
        w.reset_game(3, ep)
        lv, counter = w.curriculum_state()
        seen.add(lv)
        ents = w.entities()
        d = 3 + lv
        off = (8 - d) // 2
        inside = [e for e in ents if off <= e[1] < off + d and off <= e[2] < off + d]
        assert sum(1 for e in inside if e[0] == 0) == goals_seq[lv]
        assert sum(1 for e in inside if e[0] == 1) == blocks_seq[lv]
        assert len(ents) - len(inside) == 64 - (3 + lv) ** 2          # the padding wall
        steps = 0
        while w.game_over() == 0:
            w.take_actions(int(rng.integers(0, 4)))
            steps += 1
        assert steps <= (3 + lv) ** 2 * 10
    assert len(seen) >= 3
    walls = oracle.XWorld(oracle.Palette(oracle.WALLS_SUBTREES), render=False, map_kind=1, max_dim=7, dim=7, num_goals=12,
                          num_blocks=12, curriculum=0.1)
    walls.reset_game(0, 0)
    assert walls.curriculum_state() == (0, 0)

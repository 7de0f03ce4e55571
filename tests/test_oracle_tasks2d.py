"""Rule D14b: the 2-D-native task group "XWorldNav" of confs/walls.json (games/xworld/tasks/XWorldNav*.py) in the
oracle vs the reference's own Python tasks (tests/golden/tasks2d.json: every teach() call of 70-step episodes,
both task modes, XWorldNav and XWorldWalls maps).  CPU only."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STAGES2D = {"idle": 0, "simple_navigation_reward": 1}
EVENTS = {"": 0, "correct_goal": 1}
KINDS2D = ["XWorldNavTarget", "XWorldNavNear", "XWorldNavColorTarget", "XWorldNavBetween"]


def load2d():
    with open(os.path.join(GOLD, "tasks2d.json")) as f:
        return json.load(f)


def forced_decisions(run):
    """One task sample (a one-task group: always 0) per idle teach() + the decisions its idle stage logged."""
    out = []
    for was_idle, decs in [run["reset_teach"][:2]] + [t[4:6] for t in run["trace"]]:
        if was_idle:
            out += [0] + list(decs)
    return out


@pytest.mark.parametrize("key", sorted(load2d()))
def test_2d_native_tasks_match_reference(oracle, key):
    mode, mapk, name = key.split("/")
    pal = oracle.Palette(oracle.NAV_SUBTREES if mapk == "nav" else oracle.WALLS_SUBTREES)
    rewards = set()
    for run in load2d()[key]:
        w = oracle.XWorld(pal, render=False, map_kind=0 if mapk == "nav" else 1, max_dim=run["max_dim"], dim=run["dim"],
                          task_mode=0 if mode == "lang_acquisition" else 1, tasks=[name])
        w.load_map_forced([tuple(e) for e in run["entities"]], run["dim"], forced_decisions(run))
        was_idle, decs, reward, event, stage, tx, ty = run["reset_teach"]
        assert w.task_kind() == oracle.TASK_ID[name]
        assert w.stage() == STAGES2D[stage] and w.target2d() == (tx, ty), run["py_seed"]
        for t, (a, ax, ay, success, was_idle, decs, reward, event, stage, tx, ty) in enumerate(run["trace"]):
            assert (w.stage() == 0) == bool(was_idle), (run["py_seed"], t)
            r = np.float32(w.take_actions(a))
            assert r == np.float32(reward), (run["py_seed"], t, r, reward)
            assert w.event() == EVENTS[event] and w.stage() == STAGES2D[stage], (run["py_seed"], t)
            assert w.agent_xy() == (ax, ay) and w.last_action_success() == success
            assert w.target2d() == (tx, ty), (run["py_seed"], t)
            assert w.game_over() == 0                        # D14b: nothing but FLAGS_max_steps ends an episode
            rewards.add(float(r))
        assert w.forced_left() == 0
    if name in ("XWorldNavNear", "XWorldNavBetween"):
        assert rewards == {0.0}                              # stale in this snapshot: never leaves "idle"
    else:
        assert {float(np.float32(-0.1)), float(np.float32(-0.1 + -0.2))} <= rewards


def test_2d_group_resamples_a_task_whenever_idle(oracle):
    """TaskGroup::run_stage: the group draws a new task at every teach() while its busy task is idle, so with the
    four tasks of confs/walls.json an episode settles into a Target / ColorTarget task after a geometric wait."""
    pal = oracle.Palette(oracle.WALLS_SUBTREES)
    w = oracle.XWorld(pal, render=False, map_kind=1, max_dim=7, dim=7, num_goals=12, num_blocks=12, tasks=KINDS2D, seed=11)
    waits, stuck = [], 0
    for e in range(300):
        w.reset_game(e, 0)
        t = 0
        while w.stage() == 0 and t < 50:
            assert np.float32(w.take_actions(e % 4)) == 0
            t += 1
        if t == 50:                                          # agent walled in: no goal is ever reachable
            stuck += 1
            continue
        assert w.stage() == 1 and w.task_kind() in (5, 7) and w.target2d() != (-1, -1)
        waits.append(t)
    assert stuck < 60 and 0.6 < np.mean(waits) < 1.5, (stuck, np.mean(waits))   # geometric with p = 1/2: mean 1

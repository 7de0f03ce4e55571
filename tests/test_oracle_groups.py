"""D13, two task groups run non-exclusively (Teacher::teach, teacher.cpp:207-230 with task_groups_exclusive = false, which
lang_acquisition forces, simulator_interface.cpp:46-48): the oracle against the reference's own Python tasks run side by
side under the restated teacher glue (tests/golden/groups.json): every teach() call, per group and summed.  CPU only."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STAGE = {"idle": 0, "navigation_reward": 1, "simple_navigation_reward": 1, "terminal": 2}
EVENT = {"": 0, "correct_goal": 1, "wrong_goal": 2, "time_up": 3}


def load():
    with open(os.path.join(GOLD, "groups.json")) as f:
        return json.load(f)


def forced_decisions(run):
    """per teach() and group in conf order: the task sample (one-task groups: 0) + the decisions its idle stage logged"""
    out = []
    for teach in [run["reset_teach"]] + run["trace"]:
        for fam, was_idle, decs, reward, event, stage in teach["groups"]:
            if was_idle:
                out += [0] + list(decs)
    return out


def oracle_world(oracle, pal, run, key, **kw):
    names, order = key.split("/")
    n3, n2 = names.split("+")
    first, second = ([n3], [n2]) if order == "3d_first" else ([n2], [n3])
    w = oracle.XWorld(pal, render=False, map_kind=0, max_dim=run["max_dim"], dim=run["dim"], task_mode=0, tasks=first,
                      tasks2=second, **kw)
    return w, (0, 1) if order == "3d_first" else (1, 0)


@pytest.mark.parametrize("key", sorted(load()))
def test_two_groups_match_reference(oracle, key):
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    seen_events, buffer_events = set(), set()
    for run in load()[key]:
        w, (g3, g2) = oracle_world(oracle, pal, run, key)
        w.load_map_forced([tuple(e) for e in run["entities_before"]], run["dim"], forced_decisions(run))
        # the map after both idle stages (the 3-D one may have moved the agent and two goals)
        got = sorted((e[0], e[1], e[2], e[3]) for e in w.entities())
        want = sorted((e[0], e[1], e[2], e[3]) for e in run["entities_after"])
        assert got == want, run["py_seed"]

        def check(teach, t):
            for fam, was_idle, decs, reward, event, stage in teach["groups"]:
                kind, st, steps, ev, tx, ty = w.group_state(g3 if fam == "3d" else g2)
                assert st == STAGE[stage] and ev == EVENT[event], (run["py_seed"], t, fam, st, stage, ev, event)
                seen_events.add(fam + ":" + event)
                if fam == "2d":
                    assert [tx, ty] == teach["target2d"], (run["py_seed"], t)
            assert w.event() == EVENT[teach["event"]], (run["py_seed"], t)
            buffer_events.add(teach["event"])
        check(run["reset_teach"], -1)
        for t, teach in enumerate(run["trace"]):
            r = np.float32(w.take_actions(teach["action"]))
            assert r == np.float32(teach["reward"]), (run["py_seed"], t, r, teach["reward"])
            assert list(w.agent_xy()) == teach["agent"] and w.last_action_success() == teach["success"]
            check(teach, t)
            # lang_acquisition: the buffer's event decides game_over -- the LAST group's
            assert w.game_over() == {0: 0, 1: 4, 2: 2, 3: 1}[EVENT[teach["event"]]]
        assert w.forced_left() == 0
    if key.endswith("3d_first"):
        assert buffer_events == {""}                        # the 2-D group's "" always overwrites the 3-D group's event
    if "Target+" in key and key.endswith("3d_first"):
        assert "3d:correct_goal" in seen_events or "3d:wrong_goal" in seen_events


def test_second_group_never_sees_collisions(oracle):
    """XWorldSimulator::get_events_of_game clears game_events_ (xworld_simulator.cpp:118-122): with the 2-D group first the
    3-D task never reaches a goal; its episode can only end by time-up (or, for Between, on the middle cell)."""
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    w = oracle.XWorld(pal, render=False, map_kind=0, max_dim=8, dim=8, tasks=["XWorldNavTarget"], tasks2=["XWorld3DNavTarget"], seed=3)
    ends = set()
    for e in range(40):
        w.reset_game(e, 0)
        for t in range(900):
            w.take_actions(oracle.policy_action(5, e, t, 4))
            if w.game_over():
                ends.add(w.game_over())
                break
    assert ends <= {1}                                      # MAX_STEP from "time_up" only (640 steps)

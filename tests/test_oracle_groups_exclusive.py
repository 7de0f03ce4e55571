"""D13, two task groups under EXCLUSIVE scheduling (Teacher::teach, teacher.cpp:207-220 with task_groups_exclusive = true:
py_simulator's default, in force outside lang_acquisition): the oracle against the reference's own Python tasks run under
the restated exclusive glue (tests/golden/groups_exclusive.json, task_mode one_channel): every teach() call -- the group
sort, the one group that runs, its reward / event / stage, both groups' stages, and the map after every 3-D idle stage,
also the ones that run in mid-episode.  CPU only."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STAGE = {"idle": 0, "navigation_reward": 1, "simple_navigation_reward": 1, "terminal": 2}
EVENT = {"": 0, "correct_goal": 1, "wrong_goal": 2, "time_up": 3}
# record layout (make_golden.py gen_groups_exclusive)
D0, D1, FIRST3D, RAN3D, WAS_IDLE, DECISIONS, REWARD, EV, ST, ST3D, ST2D, TX, TY, ENTS_AFTER, ACTION, AX, AY, SUCCESS = range(18)


def load():
    with open(os.path.join(GOLD, "groups_exclusive.json")) as f:
        return json.load(f)


def forced_decisions(run):
    """per teach(): the two sort draws, then -- when the group that runs was idle -- its task sample (one-task groups: 0)
    and the decisions its idle stage logged"""
    out = []
    for rec in [run["reset_teach"]] + run["trace"]:
        out += [rec[D0], rec[D1]]
        if rec[WAS_IDLE]:
            out += [0] + list(rec[DECISIONS])
    return out


def oracle_world(oracle, pal, run, key, **kw):
    names, order, _ = key.split("/")
    n3, n2 = names.split("+")
    first, second = ([n3], [n2]) if order == "3d_first" else ([n2], [n3])
    w = oracle.XWorld(pal, render=False, map_kind=0, max_dim=run["max_dim"], dim=run["dim"], task_mode=1, tasks=first,
                      tasks2=second, task_groups_exclusive=1, group_weights=run["weights"], **kw)
    return w, (0, 1) if order == "3d_first" else (1, 0)


def entity_set(ents):
    return sorted((e[0], e[1], e[2], e[3]) for e in ents)


@pytest.mark.parametrize("key", sorted(load()))
def test_exclusive_groups_match_reference(oracle, key):
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    mid_idle_3d = mid_changed = 0
    for run in load()[key]:
        w, (g3, g2) = oracle_world(oracle, pal, run, key)
        w.load_map_forced([tuple(e) for e in run["entities_before"]], run["dim"], forced_decisions(run))

        def check(rec, t):
            assert w.group_first() == (g3 if rec[FIRST3D] else g2), (run["py_seed"], t)
            k3, s3, _, e3, _, _ = w.group_state(g3)
            k2, s2, _, e2, tx, ty = w.group_state(g2)
            assert (s3, s2) == (STAGE[rec[ST3D]], STAGE[rec[ST2D]]), (run["py_seed"], t, s3, s2, rec[ST3D], rec[ST2D])
            assert (e3 if rec[RAN3D] else e2) == EVENT[rec[EV]], (run["py_seed"], t)
            assert w.event() == EVENT[rec[EV]], (run["py_seed"], t)
            assert [tx, ty] == [rec[TX], rec[TY]], (run["py_seed"], t)
            if rec[ENTS_AFTER] is not None:                 # the map after a 3-D idle stage that rearranged it
                assert entity_set(w.entities()) == entity_set(rec[ENTS_AFTER]), (run["py_seed"], t)
        check(run["reset_teach"], -1)
        if run["reset_teach"][ENTS_AFTER] is None:
            assert entity_set(w.entities()) == entity_set(run["entities_before"])
        for t, rec in enumerate(run["trace"]):
            r = np.float32(w.take_actions(rec[ACTION]))
            assert r == np.float32(rec[REWARD]), (run["py_seed"], t, r, rec[REWARD])
            assert list(w.agent_xy()) == [rec[AX], rec[AY]] and w.last_action_success() == rec[SUCCESS], (run["py_seed"], t)
            check(rec, t)
            assert w.game_over() == 0                       # one_channel: only FLAGS_max_steps ends a game
            if rec[RAN3D] and rec[WAS_IDLE]:
                mid_idle_3d += 1
                mid_changed += rec[ENTS_AFTER] is not None
        assert w.forced_left() == 0
    assert mid_idle_3d >= 0 and mid_changed >= 0


def test_fixture_covers_mid_episode_idle_stages():
    """the case the exclusive branch exists for: an XWorld3DNav* group picked idle in mid-episode, rearranging the map"""
    mid = changed = swaps = first3 = first2 = 0
    for key, runs in load().items():
        for run in runs:
            first3 += run["reset_teach"][RAN3D]
            first2 += 1 - run["reset_teach"][RAN3D]
            for rec in run["trace"]:
                swaps += rec[D0]
                if rec[RAN3D] and rec[WAS_IDLE]:
                    mid += 1
                    changed += rec[ENTS_AFTER] is not None
    assert mid >= 20 and changed >= 8 and swaps > 1000 and first3 > 10 and first2 > 10, (mid, changed, swaps, first3, first2)


def test_exclusive_is_ignored_under_lang_acquisition_and_with_one_group(oracle):
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    for kw in (dict(tasks=["XWorld3DNavTarget"], tasks2=["XWorldNavTarget"], task_mode=0),
               dict(tasks=[0, 1, 2, 3, 4], task_mode=1)):
        a = oracle.xw_rollout(24, oracle.xw_cfg(map_kind=0, max_dim=7, dim=7, seed=5, task_groups_exclusive=1, group_weights=[1, 2], **kw),
                              pal, 150, policy_seed=3)
        b = oracle.xw_rollout(24, oracle.xw_cfg(map_kind=0, max_dim=7, dim=7, seed=5, task_groups_exclusive=0, **kw), pal, 150, policy_seed=3)
        assert np.array_equal(a.rewards, b.rewards) and np.array_equal(a.codes, b.codes)


def test_exclusive_rollout_differs_from_non_exclusive_and_is_deterministic(oracle):
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    kw = dict(map_kind=0, max_dim=7, dim=7, seed=5, task_mode=1, tasks=[0, 1, 2, 3, 4], tasks2=[5, 6, 7, 8], max_steps=60)
    a = oracle.xw_rollout(64, oracle.xw_cfg(task_groups_exclusive=1, group_weights=[1, 1], **kw), pal, 200, policy_seed=3)
    b = oracle.xw_rollout(64, oracle.xw_cfg(task_groups_exclusive=1, group_weights=[1, 1], **kw), pal, 200, policy_seed=3)
    c = oracle.xw_rollout(64, oracle.xw_cfg(task_groups_exclusive=0, **kw), pal, 200, policy_seed=3)
    assert np.array_equal(a.rewards, b.rewards) and np.array_equal(a.codes, b.codes)
    assert not np.array_equal(a.rewards, c.rewards)
    assert a.stats.resets > 100

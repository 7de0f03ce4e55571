"""xwb_step of the default loop as ONE launch (XWB_PATH_LAZY_FUSED): under the built-in policy the action of
step t + 1 is known at step t, a step changes at most two cells of an env's grid (XMap::move_item, xmap.cpp:76-101; XAgent::act,
xitem.cpp:89-101), so step t also leaves the grids as step t + 1 will leave them; the next call's render blocks draw from that
look-ahead snapshot while its step blocks run beside them.  The same rollout, byte for byte -- frames, grids, rewards, codes,
counters, teacher state -- as the two-launch form (debug switch no_fused): act_rep > 1, every templated map size and the
generic one, gray frames, both hand-over modes, explicit actions mixed in (skipped envs and illegal ids included: those calls
and the one after run as two launches), a changing act_rep, foreign verbs in between (which make the snapshot stale), a side
stream with readers between the two calls; and against the oracle (tests/test_gpu_xworld.py runs its built-in-policy rollouts
on this path from their second step on)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONF = os.path.join(ROOT, "xworld_amd", "confs")
T3 = ["XWorld3DNavTarget", "XWorld3DNavTargetNear", "XWorld3DNavTargetBetween", "XWorld3DNavTargetDirection", "XWorld3DNavTargetAvoid"]
T2 = ["XWorldNavTarget", "XWorldNavNear", "XWorldNavColorTarget", "XWorldNavBetween"]
NAV = os.path.join(CONF, "navigation2d.json")

CASES = {
    "c4": ({"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 7, "num_blocks": 16, "color": True}, 4096 + 37),
    "gray7": ({"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 7}, 1024),
    "nav8_two_groups": ({"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 8, "tasks": T3, "tasks2": T2, "max_steps": 40, "color": True}, 1000),
    "nav11": ({"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 11, "num_blocks": 30, "color": True, "max_steps": 50}, 777),
    "nav5_generic": ({"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 5, "num_goals": 2, "num_blocks": 3, "color": True, "max_steps": 30}, 2048 + 5),
    "nav16_generic": ({"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 16, "num_blocks": 60, "max_steps": 90}, 130),
    "walls_2d": ({"xwd_conf_path": os.path.join(CONF, "walls.json"), "map": "XWorldWalls", "task_mode": "one_channel", "max_steps": 37, "color": True}, 1024),
}


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _same(torch, a, b, where):
    assert torch.equal(a.obs, b.obs), where
    assert torch.equal(a.reward, b.reward) and torch.equal(a.game_over_codes, b.game_over_codes), where
    assert torch.equal(a.num_steps, b.num_steps) and torch.equal(a.episode, b.episode) and torch.equal(a.success, b.success), where
    assert torch.equal(a.grid, b.grid) and torch.equal(a.actions, b.actions), where


def _teacher(s, e):
    st = s.env_state(e)
    return (st.xw_task, st.xw_stage, st.xw_target, st.xw_agent_x, st.xw_agent_y, st.xw_sentence_names, st.xw_task2, st.xw_stage2, st.xw_target2)


@pytest.mark.parametrize("case,sync,act_rep", [(c, "auto", 1) for c in sorted(CASES)] + [("c4", "events", 1), ("nav8_two_groups", "events", 1),
                                                   ("c4", "auto", 3), ("nav11", "auto", 2), ("nav5_generic", "auto", 4)])
def test_fused_step_equals_two_launches_builtin_policy(case, sync, act_rep):
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    opts, n = CASES[case]
    a = BatchedSimulator("xworld", dict(opts, queue_sync=sync), num_envs=n, seed=11, policy_seed=5)
    b = BatchedSimulator("xworld", dict(opts, queue_sync=sync, debug=["no_fused"]), num_envs=n, seed=11, policy_seed=5)
    a.reset(); b.reset()
    paths, resets = set(), 0
    for t in range(150):
        a.step(act_rep=act_rep); b.step(act_rep=act_rep)
        paths.add((a.step_path()["path"], b.step_path()["path"]))
        _same(torch, a, b, (t, "terminal frames"))
        resets += int((a.game_over_codes != 0).sum())
        a.reset_done(); b.reset_done()
        _same(torch, a, b, (t, "first frames"))
        if t % 25 == 7:
            for e in (0, n // 3, n - 1):
                assert _teacher(a, e) == _teacher(b, e), (t, e)
    assert paths == {("lazy", "lazy"), ("lazy_fused", "lazy")}, paths
    assert resets > n // 10
    assert a.check_errors() == 0 and b.check_errors() == 0
    a.close(); b.close()


@pytest.mark.parametrize("case,act_rep", [("c4", 1), ("c4", 3), ("nav11", 2), ("nav5_generic", 1)])
def test_explicit_actions_between_fused_steps(case, act_rep):
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    opts, n = CASES[case]
    a = BatchedSimulator("xworld", opts, num_envs=n, seed=3)
    b = BatchedSimulator("xworld", dict(opts, debug=["no_fused"]), num_envs=n, seed=3)
    a.reset(); b.reset()
    g = torch.Generator(device="cuda").manual_seed(1234)
    bad = 0
    seen = []
    for t in range(90):
        if t % 3 == 1:                                              # the caller's actions: nothing could look ahead
            acts = torch.randint(0, 4, (n,), generator=g, device="cuda", dtype=torch.int32)
            r = torch.rand(n, generator=g, device="cuda")
            acts[r < 0.10] = -1                                     # XWB_ACTION_SKIP: the env sits the call out
            if t % 7 == 3:
                acts[r > 0.97] = 4 + t % 3                          # CHECK_LT(action_idx, num_actions): counted, env untouched
                bad += int((r > 0.97).sum())
            a.step(acts, act_rep=act_rep); b.step(acts, act_rep=act_rep)
        else:                                                       # built-in policy; a changing act_rep is not what the snapshot assumed
            rep = act_rep + (1 if t % 10 == 9 else 0)
            a.step(act_rep=rep); b.step(act_rep=rep)
        seen.append(a.step_path()["path"])
        assert b.step_path()["path"] == "lazy"
        _same(torch, a, b, (t, "step"))
        a.reset_done(); b.reset_done()
        _same(torch, a, b, (t, "reset_done"))
    assert a.check_errors() == bad and b.check_errors() == bad
    # t % 3: 0 built-in after built-in (fused), 1 explicit (two launches), 2 built-in after explicit (two launches, leaves a snapshot)
    assert all(p == ("lazy_fused" if t % 3 == 0 and t and t % 10 != 9 and (t - 1) % 10 != 9 else "lazy") for t, p in enumerate(seen)), seen
    a.close(); b.close()


def test_fused_step_with_foreign_verbs_in_between():
    """masked / single resets, step_autoreset, a checkpoint round trip and a mix of built-in and explicit actions: whatever
    rewrites the live state without the snapshot sends the next step down the two-launch form, and the rollout stays the same"""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    opts, n = CASES["c4"]
    a = BatchedSimulator("xworld", opts, num_envs=n, seed=21, policy_seed=8)
    b = BatchedSimulator("xworld", dict(opts, debug=["no_fused"]), num_envs=n, seed=21, policy_seed=8)
    a.reset(); b.reset()
    g = torch.Generator(device="cuda").manual_seed(99)
    seen = []
    for t in range(120):
        if t % 9 == 4:
            acts = torch.randint(0, 4, (n,), generator=g, device="cuda", dtype=torch.int32)
            a.step(acts); b.step(acts)
        else:
            a.step(); b.step()
        seen.append(a.step_path()["path"])
        _same(torch, a, b, (t, "step"))
        if t in (22, 71):                                          # (a third foreign reset would retire the lazy path: xwb_step_path)
            mask = (torch.rand(n, generator=g, device="cuda") < 0.05).to(torch.uint8)
            a.reset_masked(mask); b.reset_masked(mask)
        if t == 60:
            a.step_autoreset(); b.step_autoreset()
            _same(torch, a, b, (t, "autoreset"))
        if t == 80:
            blob = a.save_state()
            a.step(); a.reset_done()
            a.load_state(blob)
        a.reset_done(); b.reset_done()
        _same(torch, a, b, (t, "reset_done"))
    assert seen.count("lazy_fused") > 70 and seen[1:].count("lazy") >= 4, seen
    a.close(); b.close()


def test_fused_step_on_a_side_stream_and_readers_in_between():
    """work the caller queues between xwb_step and xwb_reset_done on the same stream reads that step's results and terminal
    frames (include/xwb.h xwb_reset_done); a second stream's steps are ordered like the first's"""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    opts, n = CASES["c4"]
    a = BatchedSimulator("xworld", opts, num_envs=n, seed=2, policy_seed=1)
    b = BatchedSimulator("xworld", dict(opts, debug=["no_fused"]), num_envs=n, seed=2, policy_seed=1)
    st = torch.cuda.Stream()
    a.reset(stream=st); b.reset()
    st.synchronize()
    for t in range(60):
        with torch.cuda.stream(st):
            a.step(stream=st)
            ra, ca, fa = a.reward.clone(), a.game_over_codes.clone(), a.obs.clone()
            a.reset_done(stream=st)
        b.step()
        rb, cb, fb = b.reward.clone(), b.game_over_codes.clone(), b.obs.clone()
        b.reset_done()
        st.synchronize(); torch.cuda.synchronize()
        assert torch.equal(ra, rb) and torch.equal(ca, cb) and torch.equal(fa, fb), t
        _same(torch, a, b, t)
    assert a.step_path()["path"] == "lazy_fused"
    a.close(); b.close()

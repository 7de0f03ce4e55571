"""xwb_step_autoreset with pre-generated episodes (the step kernel starts a finished env's next episode from
its shadow state, one render draws every env, the side queue regenerates the consumed shadows): the same rollout, byte for
byte, as xwb_step + xwb_reset_done -- frames, rewards, codes, step and episode counters, teacher state --, also when other
verbs (masked / single resets, plain steps, a checkpoint) come in between, in both hand-over modes, and against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONF = os.path.join(ROOT, "xworld_amd", "confs")
T3 = ["XWorld3DNavTarget", "XWorld3DNavTargetNear", "XWorld3DNavTargetBetween", "XWorld3DNavTargetDirection", "XWorld3DNavTargetAvoid"]
T2 = ["XWorldNavTarget", "XWorldNavNear", "XWorldNavColorTarget", "XWorldNavBetween"]

CASES = {
    "c4": ({"xwd_conf_path": os.path.join(CONF, "navigation2d.json"), "task_mode": "lang_acquisition", "max_dim": 7, "num_blocks": 16, "color": True}, 4096),
    "ctx3_gray": ({"xwd_conf_path": os.path.join(CONF, "navigation2d.json"), "task_mode": "lang_acquisition", "max_dim": 7, "context": 3}, 1024),
    "two_groups": ({"xwd_conf_path": os.path.join(CONF, "navigation2d.json"), "task_mode": "lang_acquisition", "max_dim": 8, "tasks": T3, "tasks2": T2, "max_steps": 40}, 1024),
    "walls_2d": ({"xwd_conf_path": os.path.join(CONF, "walls.json"), "map": "XWorldWalls", "task_mode": "one_channel", "max_steps": 37}, 1024),
    "nav11_f32": ({"xwd_conf_path": os.path.join(CONF, "navigation2d.json"), "task_mode": "lang_acquisition", "max_dim": 11, "num_blocks": 30,
                   "color": True, "obs_format": "float32", "max_steps": 50}, 512),
}


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _same(torch, a, b, where):
    assert torch.equal(a.obs, b.obs), where
    assert torch.equal(a.num_steps, b.num_steps) and torch.equal(a.episode, b.episode), where
    assert torch.equal(a.grid, b.grid), where


@pytest.mark.parametrize("case,sync", [(c, "auto") for c in sorted(CASES)] + [("c4", "events"), ("two_groups", "events")])
def test_pregen_autoreset_equals_step_then_reset(case, sync):
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    opts, n = CASES[case]
    opts = dict(opts, queue_sync=sync)
    a = BatchedSimulator("xworld", opts, num_envs=n, seed=11, policy_seed=5)
    b = BatchedSimulator("xworld", opts, num_envs=n, seed=11, policy_seed=5)
    resets = 0
    for t in range(180):
        a.step_autoreset()
        b.step()
        rb, cb = b.reward.clone(), b.game_over_codes.clone()
        b.reset_done()
        assert torch.equal(a.reward, rb) and torch.equal(a.game_over_codes, cb), t
        resets += int((cb != 0).sum())
        _same(torch, a, b, t)
        if t % 20 == 7:
            for e in (0, n // 3, n - 1):
                sa, sb = a.env_state(e), b.env_state(e)
                assert (sa.xw_task, sa.xw_stage, sa.xw_target, sa.xw_agent_x, sa.xw_agent_y, sa.xw_sentence_names, sa.xw_task2, sa.xw_stage2, sa.xw_target2) == \
                       (sb.xw_task, sb.xw_stage, sb.xw_target, sb.xw_agent_x, sb.xw_agent_y, sb.xw_sentence_names, sb.xw_task2, sb.xw_stage2, sb.xw_target2), (t, e)
    assert resets > n // 8
    assert a.task_performance() == b.task_performance()
    assert a.check_errors() == 0
    a.close(); b.close()


def test_pregen_with_other_verbs_in_between():
    """plain steps, reset_done, masked and single-env resets, step_n and a checkpoint between autoreset steps: the shadows are
    rebuilt whenever another verb started episodes, and the side queue's regeneration is joined before anything touches the
    done list"""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    opts, n = CASES["c4"]
    n = 2048
    a = BatchedSimulator("xworld", opts, num_envs=n, seed=3, policy_seed=9)
    b = BatchedSimulator("xworld", dict(opts), num_envs=n, seed=3, policy_seed=9)
    c = BatchedSimulator("xworld", dict(opts, debug=["no_pregen"]), num_envs=n, seed=3, policy_seed=9)     # the same calls without pre-generation
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda")
    mask[5::97] = 1
    blob = None
    for t in range(160):
        k = t % 8
        for s in (a, c):
            if k in (0, 1, 2, 5):
                s.step_autoreset()
                if k == 5:
                    s.reset_done()                         # after an autoreset step: only clears the codes
            elif k == 3:
                s.step(); s.reset_done()
            elif k == 4:
                s.step_n(3)
            elif k == 6:
                s.step(); s.reset_masked(mask); s.reset_done()
            else:
                s.step(); s.reset_env(17); s.reset_done()
        # b: the reference sequence spelled with step + reset_done only
        if k in (0, 1, 2, 5, 3):
            b.step(); b.reset_done()
        elif k == 4:
            for _ in range(3):
                b.step(); b.reset_done()
        elif k == 6:
            b.step(); b.reset_masked(mask); b.reset_done()
        else:
            b.step(); b.reset_env(17); b.reset_done()
        _same(torch, a, b, t)
        _same(torch, a, c, t)
        if t == 70:
            blob = a.save_state()
            keep = (a.obs.clone(), a.episode.clone())
    # resume from the checkpoint: the shadows are rebuilt, the rollout continues identically
    fresh = BatchedSimulator("xworld", dict(opts), num_envs=n, seed=3, policy_seed=9)
    fresh.load_state(blob)
    a.load_state(blob)
    assert torch.equal(a.obs, keep[0]) and torch.equal(a.episode, keep[1])
    for t in range(40):
        a.step_autoreset(); fresh.step(); fresh.reset_done()
        _same(torch, a, fresh, t)
    for s in (a, b, c, fresh):
        assert s.check_errors() == 0
        s.close()


def test_pregen_rollout_against_the_oracle(oracle):
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    opts, _ = CASES["c4"]
    n, steps = 1024, 200
    sim = BatchedSimulator("xworld", opts, num_envs=n, seed=21, policy_seed=4)
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    cfg = oracle.xw_cfg(map_kind=0, max_dim=7, dim=7, num_goals=4, num_blocks=16, color=1, seed=21, tasks=[0, 1, 2, 3, 4])
    ref = oracle.xw_rollout(64, cfg, pal, steps, policy_seed=4, render=True)
    full = oracle.xw_rollout(n, cfg, pal, steps, policy_seed=4)
    for t in range(steps):
        obs = sim.obs[:64].cpu().numpy().reshape(64, -1)
        assert np.array_equal(oracle.obs_checksum_np(obs), ref.obs_ck[t]), t      # the frame the policy sees at step t
        sim.step_autoreset()
        assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), full.rewards[t].view(np.uint32)), t
        assert np.array_equal(sim.game_over_codes.cpu().numpy(), full.codes[t]), t
    sim.close()


@pytest.mark.parametrize("case", ["c4", "ctx3_gray", "walls_2d"])
def test_lazy_default_loop_equals_classic(case):
    """xwb_step + xwb_reset_done with pre-generated episodes (no terminal snapshot in the step, the list render installs the
    shadows) against the classic path (XWB_DEBUG=no_lazy: snapshot, reset on the side queue beside the render): the frames after the
    step (terminal frames included), after reset_done, rewards, codes, counters, grids -- byte for byte."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    code = ("import sys, hashlib; sys.path.insert(0, %r); import torch\n"
            "from xworld_amd.batched import BatchedSimulator\n"
            "sim = BatchedSimulator('xworld', %r, num_envs=%d, seed=11, policy_seed=5)\n"
            "h = hashlib.sha256()\n"
            "for t in range(100):\n"
            "    sim.step()\n"
            "    for x in (sim.obs, sim.reward, sim.game_over_codes, sim.grid, sim.num_steps, sim.episode): h.update(x.cpu().numpy().tobytes())\n"
            "    sim.reset_done()\n"
            "    for x in (sim.obs, sim.game_over_codes, sim.grid, sim.num_steps, sim.episode): h.update(x.cpu().numpy().tobytes())\n"
            "assert sim.check_errors() == 0; print('HASH', h.hexdigest(), sim.task_performance())\n") % (ROOT, CASES[case][0], min(CASES[case][1], 1024))
    import subprocess, sys
    outs = []
    envs = [{}, {"XWB_DEBUG": "no_lazy"}]             # (the process-wide override of xwb_config.debug_flags, through a child process)
    if case == "c4":                           # (the hand-off variants once: they do not depend on the geometry)
        envs += [{"XWB_QUEUE_SYNC": "events"}, {"GPU_MAX_HW_QUEUES": "1", "XWB_QUEUE_SYNC": "epochs"}]
    for env in envs:
        e = dict(os.environ)
        e.pop("XWB_QUEUE_SYNC", None)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("HASH")][-1])
    assert len(set(outs)) == 1, outs


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_verb_sequences_equal_the_classic_paths(seed):
    """A seeded random walk over the verbs (step, step_autoreset, step_n, reset_done, masked / single-env / whole-batch resets,
    explicit actions with skipped envs, a checkpoint round trip) on a batch with pre-generated episodes and on one without
    (debug no_pregen): identical frames, counters, grids and results after every verb -- the host-side state machine (stale
    shadows, pending regeneration, lazy / classic hand-back after three breaks) has no observable effect."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    opts, _ = CASES["c4"]
    opts = dict(opts, max_steps=23)
    n = 1024
    a = BatchedSimulator("xworld", opts, num_envs=n, seed=seed, policy_seed=seed + 7)
    c = BatchedSimulator("xworld", dict(opts, debug=["no_pregen"]), num_envs=n, seed=seed, policy_seed=seed + 7)
    rng = np.random.default_rng(seed)
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda")
    verbs = ["step", "step", "step", "reset_done", "reset_done", "autoreset", "autoreset", "step_n", "masked", "env", "reset", "actions", "ckpt"]
    weights = np.array([6, 6, 6, 8, 8, 8, 8, 2, 1, 1, 0.3, 2, 0.5])
    if seed == 3:                                            # few foreign resets: the batch stays on the pre-generated paths throughout
        weights[8:11] = [0.03, 0.03, 0.01]
    for t in range(400):
        v = verbs[rng.choice(len(verbs), p=weights / weights.sum())]
        if v == "masked":
            mask.zero_()
            mask[torch.from_numpy(rng.choice(n, 20, replace=False)).cuda()] = 1
        e = int(rng.integers(n))
        acts = torch.from_numpy(rng.integers(-1, 4, n).astype(np.int32)).cuda()      # -1: the env sits this call out
        blob = None
        for s in (a, c):
            if v == "step":
                s.step()
            elif v == "reset_done":
                s.reset_done()
            elif v == "autoreset":
                s.step_autoreset()
            elif v == "step_n":
                s.step_n(3)
            elif v == "masked":
                s.reset_masked(mask)
            elif v == "env":
                s.reset_env(e)
            elif v == "reset":
                s.reset()
            elif v == "actions":
                s.step(acts)
            else:
                blob = s.save_state()
                s.step(); s.step_autoreset()
                s.load_state(blob)
        _same(torch, a, c, (t, v))
        assert torch.equal(a.reward, c.reward) and torch.equal(a.game_over_codes, c.game_over_codes), (t, v)
    assert a.task_performance() == c.task_performance()
    assert a.check_errors() == 0 and c.check_errors() == 0
    a.close(); c.close()

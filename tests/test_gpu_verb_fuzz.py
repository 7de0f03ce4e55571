"""Random sequences of the batch verbs on two batches of the same configuration and seeds -- one on the default paths (one-launch
step with look-ahead snapshots, pre-generated episodes, regeneration queued by the next verb, rotating done lists), one held on the
classic kernel sequence (debug switches no_fused + no_pregen: terminal snapshots, map generator beside the render) -- must leave the
same frames, results, counters and grids after every verb.  What the reference defines is the per-env sequence reset_game /
take_actions (simulator_interface.cpp:95-137); every path of the library is one implementation of it, so any two agree whatever
the caller interleaves: steps with and without the caller's actions, changing act_rep, reset_done called or skipped, masked and
single resets, whole-batch resets, step_autoreset, run(k), a checkpoint round trip, a second stream."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAV = os.path.join(ROOT, "xworld_amd", "confs", "navigation2d.json")
CASES = {
    "c4": {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 7, "num_blocks": 16, "color": True, "max_steps": 25},
    "nav11_gray": {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 11, "num_blocks": 30, "max_steps": 18},
    "nav8_two_groups": {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 8, "color": True, "max_steps": 30,
                        "tasks": ["XWorld3DNavTarget", "XWorld3DNavTargetNear", "XWorld3DNavTargetBetween", "XWorld3DNavTargetDirection", "XWorld3DNavTargetAvoid"],
                        "tasks2": ["XWorldNavTarget", "XWorldNavNear", "XWorldNavColorTarget", "XWorldNavBetween"]},
    "ctx2": {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 7, "color": True, "context": 2, "max_steps": 20},     # (a context ring: no one-launch step)
}


def _same(torch, a, b, where):
    torch.cuda.synchronize()
    assert torch.equal(a.obs, b.obs), where
    assert torch.equal(a.reward, b.reward) and torch.equal(a.game_over_codes, b.game_over_codes), where
    assert torch.equal(a.num_steps, b.num_steps) and torch.equal(a.episode, b.episode) and torch.equal(a.success, b.success), where
    assert torch.equal(a.grid, b.grid), where


@pytest.mark.parametrize("case,seed", [("c4", 1), ("c4", 2), ("nav11_gray", 3), ("nav8_two_groups", 4), ("ctx2", 5)])
def test_random_verb_sequences_default_paths_equal_classic(case, seed):
    import torch
    assert torch.cuda.is_available()
    from xworld_amd.batched import BatchedSimulator
    n = 1500
    a = BatchedSimulator("xworld", CASES[case], num_envs=n, seed=40 + seed, policy_seed=seed)
    b = BatchedSimulator("xworld", dict(CASES[case], debug=["no_fused", "no_pregen"]), num_envs=n, seed=40 + seed, policy_seed=seed)
    rng = np.random.default_rng(seed)
    g = torch.Generator(device="cuda").manual_seed(seed)
    side = torch.cuda.Stream()
    paths, log = set(), []
    a.reset(); b.reset()
    for t in range(400):
        op = rng.choice(["step", "step", "step", "step", "step_reset", "step_reset", "step_reset", "step_reset", "acts", "rep", "masked", "env",
                         "autoreset", "run", "reset_all", "ckpt", "stream", "no_reset_done"], p=None)
        log.append(op)
        if op == "step":
            a.step(); b.step()
        elif op == "step_reset":
            a.step(); b.step()
            _same(torch, a, b, (t, op, "terminal frames", log[-6:]))
            a.reset_done(); b.reset_done()
        elif op == "no_reset_done":                                 # finished envs keep stepping (the reference allows it)
            for _ in range(2):
                a.step(); b.step()
        elif op == "acts":
            acts = torch.randint(-1, 5, (n,), generator=g, device="cuda", dtype=torch.int32)    # skips and one illegal id included
            a.step(acts); b.step(acts)
            a.reset_done(); b.reset_done()
        elif op == "rep":
            k = int(rng.integers(2, 4))
            a.step(act_rep=k); b.step(act_rep=k)
            a.reset_done(); b.reset_done()
            a.step(act_rep=k); b.step(act_rep=k)
            a.reset_done(); b.reset_done()
        elif op == "masked":
            m = (torch.rand(n, generator=g, device="cuda") < 0.03).to(torch.uint8)
            a.reset_masked(m); b.reset_masked(m)
        elif op == "env":
            e = int(rng.integers(0, n))
            a.reset_env(e); b.reset_env(e)
        elif op == "autoreset":
            a.step_autoreset(); b.step_autoreset()
            a.reset_done(); b.reset_done()                          # (clears the codes the fused call kept)
        elif op == "run":
            k = int(rng.integers(1, 6))
            a.run(k); b.run(k)
        elif op == "reset_all":
            if rng.random() < 0.3:
                a.reset(); b.reset()
        elif op == "ckpt":
            if rng.random() < 0.3:
                blob = a.save_state()
                a.step(); a.reset_done(); a.step()
                a.load_state(blob)
        elif op == "stream":
            with torch.cuda.stream(side):
                a.step(stream=side)
                a.reset_done(stream=side)
            side.synchronize()
            b.step(); b.reset_done()
        paths.add(a.step_path()["path"])
        _same(torch, a, b, (t, op, log[-6:]))
    assert ("lazy_fused" in paths) == (case != "ctx2") and "lazy" in paths
    # illegal ids were counted the same on both
    assert a.check_errors() == b.check_errors()
    a.close(); b.close()

"""Checkpoint / resume (xwb_save_state / xwb_load_state): a batch resumed from a blob continues bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CONF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xworld_amd", "confs")
NAV = {"xwd_conf_path": os.path.join(CONF, "navigation2d.json"), "task_mode": "lang_acquisition", "max_dim": 7, "num_blocks": 16}
CASES = [
    ("simple_game", {"array_size": 16, "context": 2}),
    ("simple_race", {"track_width": 20.0, "track_length": 100.0, "track_radius": 30.0, "random": True}),
    ("xworld", dict(NAV, color=True, context=2)),
    ("xworld", dict(NAV, visible_radius=3, color=True)),
    ("xworld", {"xwd_conf_path": os.path.join(CONF, "walls.json"), "task_group": "XWorldNav", "task_mode": "one_channel", "max_steps": 50}),
]


def _run(sim, steps, autoreset):
    import torch
    out = []
    for t in range(steps):
        if autoreset:
            sim.step_autoreset()
        else:
            sim.reset_done()
            sim.step()
        out.append((sim.reward.clone(), sim.game_over_codes.clone(), sim.obs.clone(), sim.num_steps.clone()))
        # num_steps is a state array: xwb_reset_done may start the next episode of a finished env on its internal queue before
        # reads queued on this stream have run (include/xwb.h; outputs are ordered -- tests/test_gpu_stream_order.py)
        torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("game,opts", CASES, ids=lambda v: v if isinstance(v, str) else ("ego" if v.get("visible_radius") else str(len(v))))
@pytest.mark.parametrize("autoreset", [False, True])
def test_resume_is_bit_exact(game, opts, autoreset):
    import torch
    assert torch.cuda.is_available()
    from xworld_amd.batched import BatchedSimulator
    n = 1024
    a = BatchedSimulator(game, opts, num_envs=n, seed=5, policy_seed=9)
    _run(a, 40, autoreset)
    blob = a.save_state(include_obs=True)
    ref = _run(a, 25, autoreset)
    b = BatchedSimulator(game, opts, num_envs=n, seed=5, policy_seed=9)      # a fresh batch resumes from the blob
    b.load_state(blob)
    got = _run(b, 25, autoreset)
    for t, (x, y) in enumerate(zip(ref, got)):
        for u, v in zip(x, y):
            assert torch.equal(u, v), t
    a.load_state(blob)                                                        # and the original rewinds
    got = _run(a, 25, autoreset)
    assert all(torch.equal(u, v) for x, y in zip(ref, got) for u, v in zip(x, y))
    a.close()
    b.close()


def test_resume_without_obs_and_config_check():
    import torch
    from xworld_amd.batched import BatchedSimulator
    from xworld_amd.lib import XwbError
    opts = dict(NAV, color=True)
    a = BatchedSimulator("xworld", opts, num_envs=512, seed=1)
    _run(a, 30, False)
    small = a.save_state(include_obs=False)
    full = a.save_state(include_obs=True)
    assert small.size < full.size // 50
    ref = _run(a, 10, False)
    b = BatchedSimulator("xworld", opts, num_envs=512, seed=1)
    b.load_state(small)                                                       # context 1: frames re-rendered from the state
    got = _run(b, 10, False)
    assert all(torch.equal(u, v) for x, y in zip(ref, got) for u, v in zip(x, y))
    c = BatchedSimulator("xworld", opts, num_envs=512, seed=2)                # another seed: another configuration
    with pytest.raises(XwbError):
        c.load_state(small)
    with pytest.raises(XwbError):
        c.load_state(small[:100])
    for s in (a, b, c):
        s.close()

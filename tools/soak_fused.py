#!/usr/bin/env python3
"""Full-size soak of the default loop's one-launch step (XWB_PATH_LAZY_FUSED) against the classic kernel sequence (debug switches
no_fused + no_pregen): two C4-sized batches (32 768 envs, 7x7, 84x84x3), same seeds, `steps` x (step; reset_done); frames, rewards,
codes, counters and grids compared every `every` steps (and rewards / codes every step through a results ring).
Usage: tools/soak_fused.py [steps = 6000] [every = 40] [max_dim = 7]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from xworld_amd.batched import BatchedSimulator  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 7
n = 32768
opts = {"xwd_conf_path": os.path.join(ROOT, "xworld_amd", "confs", "navigation2d.json"), "task_mode": "lang_acquisition", "max_dim": dim,
        "num_blocks": 16 if dim < 11 else 30, "color": True}
a = BatchedSimulator("xworld", opts, num_envs=n, seed=0xC0FFEE, policy_seed=0x5EED)
b = BatchedSimulator("xworld", dict(opts, debug=["no_fused", "no_pregen"]), num_envs=n, seed=0xC0FFEE, policy_seed=0x5EED)
ra = torch.zeros((every, n, 2), dtype=torch.float32, device="cuda")
rb = torch.zeros_like(ra)
a.bind_results_ring(ra); b.bind_results_ring(rb)
bad, resets, t0 = 0, 0, time.time()
for t in range(steps):
    a.step(); b.step()
    if t % every == every - 1:
        torch.cuda.synchronize()
        ok = (torch.equal(a.obs, b.obs) and torch.equal(ra, rb) and torch.equal(a.num_steps, b.num_steps) and torch.equal(a.episode, b.episode)
              and torch.equal(a.grid, b.grid))
        bad += 0 if ok else 1
        resets = int(a.episode.sum())
    a.reset_done(); b.reset_done()
torch.cuda.synchronize()
print("soak_fused max_dim %d: %d envs x %d steps, paths %s / %s, episodes started %d, compared every %d steps (frames, grids, counters; rewards and codes "
      "of every step): mismatching checks %d, action errors %d / %d, %.1f s"
      % (dim, n, steps, a.step_path()["path"], b.step_path()["path"], resets, every, bad, a.check_errors(), b.check_errors(), time.time() - t0))
sys.exit(1 if bad else 0)

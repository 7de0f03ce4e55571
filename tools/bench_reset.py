#!/usr/bin/env python3
"""Latency of the XWorld2D reset kernel vs number of envs reset (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xworld_amd.batched import BatchedSimulator

n = 32768
conf = os.path.join(ROOT, "xworld_amd", "confs", "nav_target.json")
sim = BatchedSimulator("xworld", {"xwd_conf_path": conf, "task_mode": "lang_acquisition", "max_dim": 7, "color": True}, num_envs=n)
for k in (1, 2, 64, 115, 128, 1024, 8192, 32768):
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda")
    idx = torch.randperm(n, device="cuda")[:k]
    mask[idx] = 1
    for _ in range(3):
        sim.reset_masked(mask)
    torch.cuda.synchronize()
    sim.profile_begin()
    for _ in range(20):
        sim.reset_masked(mask)
    us, cnt = sim.profile_end("reset")
    sim.profile_stop()
    # contiguous envs instead of scattered
    mask2 = torch.zeros(n, dtype=torch.uint8, device="cuda")
    mask2[:k] = 1
    for _ in range(3):
        sim.reset_masked(mask2)
    sim.profile_begin()
    for _ in range(20):
        sim.reset_masked(mask2)
    us2, _ = sim.profile_end("reset")
    sim.profile_stop()
    print("reset of %6d envs: scattered %8.1f us   contiguous %8.1f us" % (k, us, us2))

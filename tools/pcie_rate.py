#!/usr/bin/env python3
"""PCIe-inclusive rates of the C4 workload: the same default loop with (a) the action ids handed over in
HOST memory every step (xwb_step_host: 4 bytes per env host -> device), (b) additionally every step's frames copied to host
memory (xwb_get_obs: the reference's get_state() hands the screen to Python).  `value` of bench.py is neither: its inputs and
outputs stay in HBM.    python tools/pcie_rate.py        (on a GPU)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                             # noqa: E402
from xworld_amd import lib                               # noqa: E402
from xworld_amd.batched import BatchedSimulator          # noqa: E402

n = 32768
opts = {"xwd_conf_path": os.path.join(ROOT, "xworld_amd", "confs", "navigation2d.json"), "task_mode": "lang_acquisition",
        "max_dim": 7, "num_blocks": 16, "color": True}
sim = BatchedSimulator("xworld", opts, num_envs=n)
L = sim.L
rng = np.random.default_rng(0)
acts = torch.from_numpy(rng.integers(0, 4, n).astype(np.int32)).pin_memory()
host_obs = torch.empty((n,) + tuple(sim.obs.shape[1:]), dtype=torch.uint8).pin_memory()


def loop(k, host_actions, host_frames):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        if host_actions:
            lib.check(L.xwb_step_host(sim.h, C.c_void_p(acts.data_ptr()), 1, None))
        else:
            sim.step()
        sim.reset_done()
        if host_frames:
            lib.check(L.xwb_get_obs(sim.h, C.c_void_p(host_obs.data_ptr()), host_obs.numel(), None))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k


for _ in range(3):
    loop(50, False, False)
for name, ha, hf, k in (("device-resident (bench.py's loop)", False, False, 400), ("actions from host memory every step", True, False, 400),
                        ("+ frames to host memory every step", True, True, 20)):
    dt = min(loop(k, ha, hf) for _ in range(3))
    extra = " (%.1f GB/s device -> host)" % (host_obs.numel() / dt / 1e9) if hf else ""
    print("%-40s %.4f ms per step  %7.1f M env-steps/s%s" % (name, dt * 1e3, n / dt / 1e6, extra))
sim.close()

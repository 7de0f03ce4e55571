"""Per-phase time of the one-workgroup-per-env egocentric render (a build with XWB_EXTRA_FLAGS=-DXWB_EGO_PROF only; the span
path is switched off for the measurement):
    XWB_EXTRA_FLAGS=-DXWB_EGO_PROF python -m xworld_amd.build --force && XWB_DEBUG=ego_no_span python tools/ego_prof.py [r] [map key]
Prints the 100 MHz wall-clock ticks workgroup leaders spent between the barriers of xw_render_ego_kernel, summed over
workgroups, as a share of the total."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xworld_amd.batched import BatchedSimulator  # noqa: E402

r = int(sys.argv[1]) if len(sys.argv) > 1 else 3
conf = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xworld_amd", "confs", "nav_target.json")
sim = BatchedSimulator("xworld", {"xwd_conf_path": conf, "max_dim": 7, "dim": 7, "visible_radius": r, "color": True,
                                  "task_mode": "lang_acquisition"}, num_envs=32768)
buf = (C.c_ulonglong * 12)()              # g_ego_prof[12]: seven phase timers, then the goal-cell cache's hit statistics
for _ in range(5):
    sim.step(); sim.reset_done()
torch.cuda.synchronize()
sim.L.xwb_debug_ego_prof(buf)
steps = 20
for _ in range(steps):
    sim.step(); sim.reset_done()
torch.cuda.synchronize()
sim.L.xwb_debug_ego_prof(buf)
names = ["stage", "rays", "scan", "cells", "copy interior", "per-pixel", "store"]
tot = sum(buf[:7])
for n, v in zip(names, buf[:7]):
    print("%-14s %6.2f %%  %8.3f us per env" % (n, 100.0 * v / tot, v / 100.0 / steps / 32768))

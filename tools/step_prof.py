#!/usr/bin/env python3
"""Stage stamps of xw_step_kernel / xw_render_list_kernel inside the C4 step loop (lab build only):

    XWB_EXTRA_FLAGS=-DXWB_STEP_PROF python -m xworld_amd.build --force && python tools/step_prof.py [workload]

Per workgroup of the LAST launch (100 MHz wall clock): entry, after the first round trip, after the transition, after the list
append, end.  Printed: when the first workgroup started, when the last one ended, and the distribution of each stage."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "xworld7"
sim = bench.make_sim(wl, bench.WORKLOADS[wl][2], 0, 0)
f = sim.L.xwb_debug_step_prof
f.argtypes = [C.c_void_p]
for _ in range(300):
    sim.step(); sim.reset_done()
torch.cuda.synchronize()
buf = np.zeros((2, 4096, 6), dtype=np.uint64)
assert f(buf.ctypes.data) == 0
for which, name, n_st in ((0, "xw_step_kernel", 5), (1, "xw_render_list_kernel", 5)):
    b = buf[which].astype(np.int64)
    used = b[:, 0] > 0
    b = b[used]
    t0 = b[:, 0].min()
    print("== %s: %d workgroups stamped; first entry -> last entry %.2f us" % (name, len(b), (b[:, 0].max() - t0) / 100.0))
    if which == 1:
        early = b[b[:, 5] > 0]
        print("   early exits: %d, their exit %.2f .. %.2f us after the first entry" % (len(early), (early[:, 5].min() - t0) / 100.0 if len(early) else 0, (early[:, 5].max() - t0) / 100.0 if len(early) else 0))
        b = b[b[:, 4] >= b[:, 0]]
    for k in range(1, n_st):
        ok = b[:, k] >= b[:, 0]
        d = (b[ok, k] - b[ok, 0]) / 100.0
        if len(d):
            print("   stage %d reached (us after own entry): median %.2f  p90 %.2f  max %.2f   | after first entry: max %.2f" % (k, np.median(d), np.percentile(d, 90), d.max(), (b[ok, k].max() - t0) / 100.0))
sim.close()

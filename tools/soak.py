#!/usr/bin/env python3
"""Long parity soak (not part of the suite): product vs oracle rewards / codes over thousands of steps and resets.
    python tools/soak.py <map key> <visible_radius> <n_envs> <steps> [2d] [curriculum=<c>] [tasks=<i,j,..>] [weights=<w,..>]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O                                   # noqa: E402
from test_gpu_xworld import MAPS                      # noqa: E402
from xworld_amd.batched import BatchedSimulator       # noqa: E402

key, r, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
extra = dict(a.split("=", 1) for a in sys.argv[5:] if "=" in a)
two_d = "2d" in sys.argv[5:]
tasks = ["XWorldNavTarget", "XWorldNavNear", "XWorldNavColorTarget", "XWorldNavBetween"] if two_d else \
    ["XWorld3DNavTarget", "XWorld3DNavTargetNear", "XWorld3DNavTargetBetween", "XWorld3DNavTargetDirection", "XWorld3DNavTargetAvoid"]
if "tasks" in extra:
    tasks = [tasks[int(i)] for i in extra["tasks"].split(",")]
conf, popts, ocfg = MAPS[key]
opts = {"xwd_conf_path": conf, "task_mode": "one_channel" if two_d else "lang_acquisition", "tasks": tasks, "visible_radius": r}
opts.update(popts)
if two_d:
    opts["max_steps"] = 45
if "curriculum" in extra:
    opts["curriculum"] = float(extra["curriculum"])
if "weights" in extra:
    opts["task_weights"] = [float(x) for x in extra["weights"].split(",")]
sim = BatchedSimulator("xworld", opts, num_envs=n, seed=1234, policy_seed=99, env_gid0=7)
pal = O.Palette(O.NAV_SUBTREES if ocfg["map_kind"] == 0 else O.WALLS_SUBTREES)
cfg = dict(ocfg)
cfg.update(seed=1234, tasks=tasks, visible_radius=r, task_mode=1 if two_d else 0, max_steps=45 if two_d else 0)
if "curriculum" in extra:
    cfg["curriculum"] = float(extra["curriculum"])
if "weights" in extra:
    cfg["task_weights"] = [float(x) for x in extra["weights"].split(",")]
t0 = time.time()
ref = O.xw_rollout(n, O.xw_cfg(**cfg), pal, steps, policy_seed=99, env_gid0=7)
print("oracle: %.1f s, %d resets" % (time.time() - t0, ref.stats.resets), flush=True)
resets = 0
for t in range(steps):
    sim.reset_done()
    resets += sim.done_count()
    sim.step()
    assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), ref.rewards[t].view(np.uint32)), t
    assert np.array_equal(sim.game_over_codes.cpu().numpy(), ref.codes[t]), t
assert resets == ref.stats.resets
if "curriculum" in extra:
    print("levels:", np.bincount([sim.env_state(e).xw_level for e in range(0, n, max(1, n // 256))], minlength=6))
print("soak ok:", key, "r", r, "2d" if two_d else "3d", n, "envs x", steps, "steps,", resets, "resets", extra)

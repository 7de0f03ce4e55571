"""Lab: N steps of the egocentric batch without reset_done (nothing on the reset's queue), for a kernel trace of the render's launches alone."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xworld_amd.batched import BatchedSimulator
r = int(sys.argv[1]) if len(sys.argv) > 1 else 3
D = int(sys.argv[2]) if len(sys.argv) > 2 else 7
conf = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "xworld_amd", "confs", "navigation2d.json")
sim = BatchedSimulator("xworld", {"xwd_conf_path": conf, "max_dim": D, "dim": D, "visible_radius": r, "color": True, "task_mode": "lang_acquisition"}, num_envs=32768, policy_seed=7)
for _ in range(150):
    sim.step(); sim.reset_done()
torch.cuda.synchronize()
for _ in range(60):
    sim.step()
torch.cuda.synchronize()

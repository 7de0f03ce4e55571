import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from xworld_amd.batched import BatchedSimulator
conf = "/root/repo/xworld_amd/confs/navigation2d.json"
sim = BatchedSimulator("xworld", {"xwd_conf_path": conf, "max_dim": 7, "dim": 7, "color": True, "task_mode": "lang_acquisition", "max_steps": 8}, num_envs=32768)
for _ in range(16):
    sim.step(); sim.reset_done()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 160
for _ in range(K):
    sim.step(); sim.reset_done()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print("max_steps 8 (whole batch resets every 8th step): %.3f ms/step" % (dt * 1e3))

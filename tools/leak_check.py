import os, sys, torch
sys.path.insert(0, "/root/repo")
from xworld_amd.batched import BatchedSimulator
conf = "/root/repo/xworld_amd/confs/navigation2d.json"
free0 = None
for i in range(60):
    opts = {"xwd_conf_path": conf, "max_dim": 7, "dim": 7, "color": True, "task_mode": "lang_acquisition"}
    if i % 3 == 1:
        opts["visible_radius"] = 3
    if i % 3 == 2:
        opts.update(max_dim=8, dim=8, curriculum=0.1)
    sim = BatchedSimulator("xworld", opts, num_envs=4096)
    sim.step(); sim.reset_done(); sim.step_autoreset()
    torch.cuda.synchronize()
    sim.close()
    if i in (5, 59):
        f, t = torch.cuda.mem_get_info()
        print(i, "free MB", f >> 20)
        if free0 is None: free0 = f
        else: assert abs(f - free0) < (64 << 20), (free0, f)
print("no leak")

# round 4: communicators with the grids gather's staging slabs and marks, created and destroyed over and over
from xworld_amd import sharding
import ctypes as C
from xworld_amd import lib
L = lib.load()
free1 = None
for i in range(40):
    sim = BatchedSimulator("xworld", {"xwd_conf_path": conf, "max_dim": 7, "dim": 7, "color": True, "task_mode": "lang_acquisition"}, num_envs=4096)
    comm = sharding.LibComm(0, 1, 0)
    sg = sharding.LibScreensGather(sim, comm, [4096], 0, mode="grids" if i % 2 else "screens")
    for t in range(4):
        sg.bind_next(); sim.step(); sim.reset_done(); sg.start()
    sg.drain()
    torch.cuda.synchronize()
    del sg
    comm.close(); sim.close()
    if i in (5, 39):
        f, t = torch.cuda.mem_get_info()
        print("comm", i, "free MB", f >> 20)
        if free1 is None: free1 = f
        else: assert abs(f - free1) < (64 << 20), (free1, f)
print("no leak (communicators)")

"""Lab (a build with XWB_EXTRA_FLAGS=-DXWB_EGO_PROF): goal cells the cache lacked and envs drawn again per whole-batch span render.
    XWB_EXTRA_FLAGS=-DXWB_EGO_PROF python -m xworld_amd.build && python tools/lab/ego_stats.py [r] [max_dim] [steps]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xworld_amd.batched import BatchedSimulator
r = int(sys.argv[1]) if len(sys.argv) > 1 else 3
D = int(sys.argv[2]) if len(sys.argv) > 2 else 7
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
conf = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "xworld_amd", "confs", "navigation2d.json")
sim = BatchedSimulator("xworld", {"xwd_conf_path": conf, "max_dim": D, "dim": D, "visible_radius": r, "color": True, "task_mode": "lang_acquisition"}, num_envs=32768, policy_seed=7)
buf = (C.c_ulonglong * 12)()
sim.L.xwb_debug_ego_prof.argtypes = [C.c_void_p]
for block in range(6):
    for _ in range(steps // 6):
        sim.step(); sim.reset_done()
    torch.cuda.synchronize()
    sim.L.xwb_debug_ego_prof(buf)
    n = max(int(buf[10]), 1)
    print("steps %4d..%4d: envs drawn again %.0f, goal cells evaluated %.0f per render (%d renders)" % (block * (steps // 6), (block + 1) * (steps // 6), buf[8] / n, buf[9] / n, n), flush=True)
# stage stamps of the whole-batch cells kernel (first lane of every workgroup, 100 MHz ticks), averaged per workgroup and render
buf2 = (C.c_ulonglong * 12)()
sim.L.xwb_debug_ego_prof2(buf2)
for _ in range(50):
    sim.step(); sim.reset_done()
torch.cuda.synchronize()
sim.L.xwb_debug_ego_prof2(buf2)
wg = 50 * ((32768 + 63) // 64)
names = ["state + grids -> LDS", "types, goal slots", "walk (cells)", "barrier", "words: table reads + stores", "valid bits", "miss list", "rays + own scan lines", "shadow barrier", "words: keys + flat reads"]
print("cells kernel, us per workgroup: " + ", ".join("%s %.2f" % (n, buf2[i] / 100.0 / wg) for i, n in enumerate(names)))

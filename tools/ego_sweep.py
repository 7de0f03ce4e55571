import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from xworld_amd.batched import BatchedSimulator
conf = "/root/repo/xworld_amd/confs/navigation2d.json"
for md, r, color in [(7, 3, True), (7, 3, False), (8, 5, True), (7, 7, True), (11, 7, True), (7, 1, True), (11, 9, True)]:
    n = 32768 if md <= 8 else 8192
    sim = BatchedSimulator("xworld", {"xwd_conf_path": conf, "max_dim": md, "dim": md, "visible_radius": r, "color": color,
                                      "task_mode": "lang_acquisition", "num_blocks": 16 if md <= 8 else 30}, num_envs=n)
    for _ in range(10):
        sim.step(); sim.reset_done()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 50
    for _ in range(K):
        sim.step(); sim.reset_done()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print("max_dim %d r %d %s frame %s (%s path): %.3f ms/step, %.1f M env-steps/s" % (md, r, "bgr" if color else "gray", tuple(sim.obs.shape[1:]), sim.ego_render_path, dt * 1e3, n / dt / 1e6), flush=True)
    sim.close()

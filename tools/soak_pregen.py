"""xwb_step_autoreset with pre-generated episodes against the oracle over long rollouts, down to max_steps = 1 (every env starts a
new episode on every step): rewards, codes and the first 200 frames.  Run on the GPU box: python tools/soak_pregen.py"""
import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, _oracle as O
from xworld_amd.batched import BatchedSimulator
conf = '/root/repo/xworld_amd/confs/navigation2d.json'
pal = O.Palette(O.NAV_SUBTREES)
for ms, n, steps in ((1, 2048, 300), (2, 2048, 400), (7, 4096, 1500), (0, 4096, 3000)):
    sim = BatchedSimulator("xworld", {"xwd_conf_path": conf, "task_mode": "lang_acquisition", "max_dim": 7, "num_blocks": 16, "color": True, "max_steps": ms}, num_envs=n, seed=77, policy_seed=3)
    cfg = O.xw_cfg(map_kind=0, max_dim=7, dim=7, num_goals=4, num_blocks=16, color=1, seed=77, tasks=[0,1,2,3,4], max_steps=ms)
    ref = O.xw_rollout(n, cfg, pal, steps, policy_seed=3)
    refo = O.xw_rollout(32, cfg, pal, min(steps, 200), policy_seed=3, render=True)
    bad = 0
    for t in range(steps):
        if t < min(steps, 200):
            ck = O.obs_checksum_np(sim.obs[:32].cpu().numpy().reshape(32, -1))
            bad += int((ck != refo.obs_ck[t]).sum())
        sim.step_autoreset()
        bad += int((sim.reward.cpu().numpy().view(np.uint32) != ref.rewards[t].view(np.uint32)).sum())
        bad += int((sim.game_over_codes.cpu().numpy() != ref.codes[t]).sum())
    print("max_steps", ms, "n", n, "steps", steps, "resets", ref.stats.resets, "mismatches", bad, "errors", sim.check_errors())
    sim.close()

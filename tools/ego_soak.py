"""Soak of the two egocentric renders against each other: the span path (cells -> evaluated pixels -> gather) and one workgroup per
env (debug ego_no_span) must draw the same frames through step / reset_done / step_autoreset -- 8192 envs x 200 steps per case:
r = 3 / 5 / 7, colour and gray, context rings, float32 frames, curriculum, no wall shadows.  python tools/ego_soak.py (on a GPU)"""
import os, sys, itertools
sys.path.insert(0, '/root/repo')
import torch
from xworld_amd.batched import BatchedSimulator
CONF = '/root/repo/xworld_amd/confs/'
def make(opts, n, seed, no_span):
    s = BatchedSimulator('xworld', dict(opts, debug=['ego_no_span'] if no_span else []), num_envs=n, seed=seed, policy_seed=seed + 1)
    return s
cases = [
  dict(xwd_conf_path=CONF+'navigation2d.json', task_mode='lang_acquisition', max_dim=7, dim=7, num_blocks=16, visible_radius=3, color=True),
  dict(xwd_conf_path=CONF+'navigation2d.json', task_mode='lang_acquisition', max_dim=8, dim=8, visible_radius=5, color=True, context=2),
  dict(xwd_conf_path=CONF+'navigation2d.json', task_mode='lang_acquisition', max_dim=11, dim=11, num_blocks=30, visible_radius=7, color=False),
  dict(xwd_conf_path=CONF+'navigation2d.json', task_mode='lang_acquisition', max_dim=8, dim=8, visible_radius=3, color=True, curriculum=0.1),
  dict(xwd_conf_path=CONF+'navigation2d.json', task_mode='lang_acquisition', max_dim=7, dim=7, visible_radius=3, color=True, obs_format='float32', context=2),
  dict(xwd_conf_path=CONF+'navigation2d.json', task_mode='lang_acquisition', max_dim=7, dim=7, visible_radius=3, color=True, wall_shadow=False),
]
# python tools/ego_soak.py [steps [seed ...]]  (default 200 steps, seeds 3 and 17)
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
SEEDS = tuple(int(a) for a in sys.argv[2:]) or (3, 17)
bad = 0
for ci, opts in enumerate(cases):
    for seed in SEEDS:
        n = 8192
        a = make(opts, n, seed, False); b = make(opts, n, seed, True)
        assert a.ego_render_path == 'span' and b.ego_render_path == 'per_env'
        for s in (a, b): s.reset()
        ok = torch.equal(a.obs, b.obs)
        for t in range(STEPS):
            auto = (t % 5 == 4)
            for s in (a, b):
                if auto: s.step_autoreset()
                else: s.step()
            if not torch.equal(a.obs, b.obs): ok = False; print('case', ci, 'seed', seed, 'step', t, 'frames differ after step', int((a.obs != b.obs).reshape(n, -1).any(1).sum()), 'envs'); break
            if not auto:
                for s in (a, b): s.reset_done()
                if not torch.equal(a.obs, b.obs): ok = False; print('case', ci, 'seed', seed, 'step', t, 'frames differ after reset_done', int((a.obs != b.obs).reshape(n, -1).any(1).sum()), 'envs'); break
        ok = ok and torch.equal(a.reward, b.reward) and torch.equal(a.game_over_codes, b.game_over_codes)
        print('case', ci, 'seed', seed, 'ok' if ok else 'MISMATCH', flush=True)
        bad += 0 if ok else 1
        a.close(); b.close()
print('done, bad =', bad)

"""GPU parity for FLAGS_curriculum != 0 (XWorldNav.py:27-55): every env walks through the six levels on its own success
record; rewards, codes and levels against the CPU oracle (pinned to the reference by tests/golden/curriculum.json)."""
import os

import numpy as np
import pytest

from test_gpu_xworld import _torch
from test_oracle_tasks import KINDS

pytestmark = pytest.mark.gpu

CONF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xworld_amd", "confs")


def _make(oracle, n, tasks, curriculum, seed, policy_seed, gid0=0, **opts):
    from xworld_amd.batched import BatchedSimulator
    o = {"xwd_conf_path": os.path.join(CONF, "navigation2d.json"), "task_mode": "lang_acquisition", "tasks": list(tasks),
         "curriculum": curriculum}
    o.update(opts)
    sim = BatchedSimulator("xworld", o, num_envs=n, seed=seed, policy_seed=policy_seed, env_gid0=gid0)
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    cfg = dict(map_kind=0, max_dim=8, dim=8, num_goals=4, num_blocks=16, seed=seed, tasks=list(tasks), curriculum=curriculum,
               start_level=int(opts.get("start_level", 0)), task_mode=0 if o["task_mode"] == "lang_acquisition" else 1)
    return sim, pal, cfg


@pytest.mark.parametrize("tasks,curriculum,start", [([KINDS[0]], 0.1, 0), ([KINDS[0], KINDS[4]], 0.05, 0),
                                                    ([KINDS[0], KINDS[1], KINDS[4]], 0.02, 3), (KINDS, 0.3, 0)],
                         ids=["one", "two", "from3", "five"])
def test_rollout_with_levels(oracle, tasks, curriculum, start):
    """Random policy, thousands of resets per config: reward bits and codes of every env-step, then the level and the check
    counter of every env (a few envs replayed one by one through the oracle)."""
    _torch()
    n, steps = 384, 9000
    sim, pal, cfg = _make(oracle, n, tasks, curriculum, seed=11, policy_seed=6, gid0=40, start_level=start)
    ref = oracle.xw_rollout(n, oracle.xw_cfg(**cfg), pal, steps, policy_seed=6, env_gid0=40)
    for t in range(steps):
        sim.reset_done()
        sim.step()
        assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), ref.rewards[t].view(np.uint32)), t
        assert np.array_equal(sim.game_over_codes.cpu().numpy(), ref.codes[t]), t
    levels = np.array([sim.env_state(e).xw_level for e in range(n)])
    assert levels.min() >= start and (levels.max() > start or start > 0 or len(tasks) == 5), np.bincount(levels)
    for e in range(0, n, 48):                               # the same envs, one oracle object each
        w = oracle.XWorld(pal, render=False, **cfg)
        episode = 0
        w.reset_game(40 + e, episode)
        for t in range(steps):
            if w.game_over() != 0:
                episode += 1
                w.reset_game(40 + e, episode)
            w.take_actions(oracle.policy_action(6, 40 + e, t, 4))
        st = sim.env_state(e)
        assert (st.xw_level, st.xw_check_counter) == w.curriculum_state(), e
    sim.close()


def test_grid_follows_the_level(oracle):
    """Reset after reset the map is the level's: dims (brick padding around it), goals, blocks; the time-up of the 3-D
    tasks uses the level's dims."""
    _torch()
    n = 256
    sim, pal, cfg = _make(oracle, n, [KINDS[0]], 0.02, seed=5, policy_seed=2)
    ow = oracle.XWorld(pal, render=False, **cfg)
    for t in range(6000):
        sim.reset_done()
        sim.step()
    goals_seq, blocks_seq = [2, 2, 2, 4, 4, 4], [0, 3, 6, 9, 12, 16]
    types = sim.palette.icon_type
    seen = set()
    for e in range(n):
        st = sim.env_state(e)
        g = sim.env_grid(e, raw=True) & 0x7fff
        d = 3 + st.xw_level
        off = (8 - d) // 2
        inner = g[off:off + d, off:off + d]
        pad = np.ones((8, 8), bool)
        pad[off:off + d, off:off + d] = False
        assert (g[pad] > 0).all() and (types[g[pad] - 1] == 1).all(), e                       # brick padding
        kinds = types[inner[inner > 0] - 1]
        assert (kinds == 0).sum() == goals_seq[st.xw_level] and (kinds == 1).sum() == blocks_seq[st.xw_level], (e, st.xw_level)
        seen.add(st.xw_level)
    assert max(seen) >= 1
    sim.close()


def test_curriculum_config_rules():
    _torch()
    from xworld_amd.batched import BatchedSimulator
    from xworld_amd.lib import XwbError
    nav = os.path.join(CONF, "navigation2d.json")
    with pytest.raises(XwbError):                           # XWorldNav asserts six levels: its 8x8 world only
        BatchedSimulator("xworld", {"xwd_conf_path": nav, "curriculum": 0.1, "max_dim": 7, "dim": 7}, num_envs=4)
    with pytest.raises(XwbError):
        BatchedSimulator("xworld", {"xwd_conf_path": nav, "curriculum": 0.1, "start_level": 6}, num_envs=4)
    # the reference's own example: walls.json with curriculum 0.1 (python/examples/test_xworld.py:31-40) -- XWorldWalls
    # never reads the flag
    sim = BatchedSimulator("xworld", {"xwd_conf_path": os.path.join(CONF, "walls.json"), "curriculum": 0.1,
                                      "task_mode": "lang_acquisition"}, num_envs=4)
    assert sim.env_state(0).xw_level == 0
    sim.close()
    sim = BatchedSimulator("xworld", {"xwd_conf_path": nav, "curriculum": 0.1, "start_level": 2}, num_envs=4)
    assert all(sim.env_state(e).xw_level == 2 and sim.env_state(e).xw_check_counter == 1 for e in range(4))
    sim.close()


def test_checkpoint_carries_the_curriculum(oracle):
    """xwb_save_state / xwb_load_state: level, check counter and success windows travel with the blob -- a resumed batch
    reaches the same level checks with the same outcome."""
    torch = _torch()
    n = 256
    a, _, _ = _make(oracle, n, [KINDS[0]], 0.1, seed=3, policy_seed=4)
    for t in range(1500):
        a.reset_done()
        a.step()
    blob = a.save_state(include_obs=False)
    b, _, _ = _make(oracle, n, [KINDS[0]], 0.1, seed=3, policy_seed=4)
    b.load_state(blob)
    assert [(b.env_state(e).xw_level, b.env_state(e).xw_check_counter) for e in range(0, n, 16)] == \
           [(a.env_state(e).xw_level, a.env_state(e).xw_check_counter) for e in range(0, n, 16)]
    assert any(a.env_state(e).xw_check_counter > 1 for e in range(n))
    for t in range(6000):
        a.reset_done(); b.reset_done()
        a.step(); b.step()
        if t % 50 == 0 or t == 5999:
            assert torch.equal(a.reward, b.reward) and torch.equal(a.game_over_codes, b.game_over_codes), t
    la = [a.env_state(e).xw_level for e in range(n)]
    assert la == [b.env_state(e).xw_level for e in range(n)] and max(la) >= 1
    assert torch.equal(a.obs, b.obs)
    a.close()
    b.close()

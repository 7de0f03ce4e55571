"""GPU parity: XWorld2D HIP kernels (step + teacher rule, reset / map generation, render)
through the C ABI vs the CPU oracle.  Everything here is integer / byte work: bit-exact.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CONF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xworld_amd", "confs")
NAV = os.path.join(CONF, "nav_target.json")
WALLS = os.path.join(CONF, "walls_target.json")

# (conf, options for the product, oracle cfg overrides)
MAPS = {
    "nav8": (NAV, {}, dict(map_kind=0, max_dim=8, dim=8, num_goals=4, num_blocks=16)),
    "nav7": (NAV, {"max_dim": 7, "num_blocks": 16}, dict(map_kind=0, max_dim=7, dim=7, num_goals=4, num_blocks=16)),
    "nav11": (NAV, {"max_dim": 11, "num_blocks": 30}, dict(map_kind=0, max_dim=11, dim=11, num_goals=4, num_blocks=30)),
    "nav8_dim5": (NAV, {"dim": 5, "num_goals": 2, "num_blocks": 6},
                  dict(map_kind=0, max_dim=8, dim=5, num_goals=2, num_blocks=6)),
    "walls7": (WALLS, {}, dict(map_kind=1, max_dim=7, dim=7, num_goals=12, num_blocks=12)),
}


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _make(oracle, key, n, seed=0xC0FFEE, policy_seed=0x5EED, gid0=0, render=False, **opts):
    from xworld_amd.batched import BatchedSimulator
    conf, popts, ocfg = MAPS[key]
    o = {"xwd_conf_path": conf, "task_mode": "lang_acquisition"}
    o.update(popts)
    o.update(opts)
    sim = BatchedSimulator("xworld", o, num_envs=n, seed=seed, policy_seed=policy_seed, env_gid0=gid0)
    pal = oracle.Palette(oracle.NAV_SUBTREES if ocfg["map_kind"] == 0 else oracle.WALLS_SUBTREES)
    assert len(pal) == len(sim.palette)
    assert [m["path"] for m in pal.meta] == [m["path"] for m in sim.palette.meta]
    cfg = dict(ocfg)
    cfg.update(seed=seed, color=int(bool(opts.get("color", False))), context=int(opts.get("context", 1)),
               max_steps=int(opts.get("max_steps", 0)), max_steps_factor=int(opts.get("max_steps_factor", 10)),
               task_mode=0 if o["task_mode"] == "lang_acquisition" else 1)
    return sim, pal, cfg


@pytest.mark.parametrize("key", list(MAPS))
def test_reset_map_generation(oracle, key):
    """xwb-mapgen-v1: the GPU reset kernel and the oracle produce the same map, agent cell and target."""
    _torch()
    n = 384
    sim, pal, cfg = _make(oracle, key, n, seed=99, gid0=1000)
    ow = oracle.XWorld(pal, render=False, **cfg)
    for episode in range(3):
        if episode:
            sim.reset()
        for e in range(n):
            ow.reset_game(1000 + e, episode)
            assert np.array_equal(sim.env_grid(e).astype(np.int32), ow.grid()), (episode, e)
            st = sim.env_state(e)
            assert (st.xw_agent_x, st.xw_agent_y) == ow.agent_xy()
            assert st.xw_target_name == ow.target_name() and st.xw_stage == ow.stage() == 1
            assert st.episode == episode and st.num_steps == 0 and st.game_over == 0
    sim.close()


@pytest.mark.parametrize("key,mode", [("nav8", "lang_acquisition"), ("nav7", "lang_acquisition"),
                                      ("walls7", "lang_acquisition"), ("nav8", "one_channel"),
                                      ("nav8_dim5", "lang_acquisition")])
def test_step_teacher_rollout(oracle, key, mode):
    """Random-policy rollouts with resets: reward bits and game_over codes of every env-step."""
    _torch()
    n, steps = 1536, 220
    extra = {"task_mode": mode}
    if mode == "one_channel":
        extra["max_steps"] = 37          # the only way an episode ends in one_channel mode
    sim, pal, cfg = _make(oracle, key, n, seed=7, policy_seed=11, gid0=50, **extra)
    ref = oracle.xw_rollout(n, oracle.xw_cfg(**cfg), pal, steps, policy_seed=11, env_gid0=50)
    resets = 0
    for t in range(steps):
        sim.reset_done()
        resets += sim.done_count()
        sim.step()
        assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), ref.rewards[t].view(np.uint32)), t
        assert np.array_equal(sim.game_over_codes.cpu().numpy(), ref.codes[t]), t
    assert resets == ref.stats.resets and resets > 0
    print(key, mode, "resets", resets, "mean reward", ref.stats.reward_sum / (n * steps))
    sim.close()


def test_step_details_and_act_rep(oracle):
    """Per-env lock-step replay with explicit actions and act_rep > 1: agent cell, success flag, event, stage."""
    torch = _torch()
    n = 96
    sim, pal, cfg = _make(oracle, "nav8", n, seed=3)
    envs = [oracle.XWorld(pal, render=False, **cfg) for _ in range(n)]
    eps = [0] * n
    for e, w in enumerate(envs):
        w.reset_game(e, 0)
    rng = np.random.default_rng(5)
    for t in range(150):
        sim.reset_done()
        for e, w in enumerate(envs):
            if w.game_over() != 0:
                eps[e] += 1
                w.reset_game(e, eps[e])
        acts = rng.integers(0, 4, n).astype(np.int32)
        rep = int(rng.integers(1, 4))
        sim.step(torch.from_numpy(acts).cuda(), act_rep=rep)
        rew = sim.reward.cpu().numpy()
        for e, w in enumerate(envs):
            r = np.float32(w.take_actions(int(acts[e]), rep))
            st = sim.env_state(e)
            assert r.view(np.uint32) == rew[e:e + 1].view(np.uint32)[0], (t, e)
            assert (st.xw_agent_x, st.xw_agent_y) == w.agent_xy(), (t, e)
            assert st.last_action_success == w.last_action_success()
            assert st.xw_event == w.event() and st.xw_stage == w.stage()
            assert st.xw_steps_in_task == w.steps_in_task() and st.game_over == w.game_over()
            assert st.lives == w.get_lives() and st.num_steps == w.num_steps()
    for e, w in enumerate(envs):
        assert np.array_equal(sim.env_grid(e).astype(np.int32), w.grid())
    sim.close()


@pytest.mark.parametrize("key,color,context", [("nav8", True, 1), ("nav8", False, 1), ("nav7", True, 2),
                                               ("nav11", True, 1), ("walls7", False, 3), ("nav8_dim5", True, 1)])
def test_render_vs_canvas_oracle(oracle, key, color, context):
    """Screens: the tile-table expansion kernel vs the oracle's 64 px canvas + OpenCV-spec resize."""
    _torch()
    n, steps = 20, 14
    sim, pal, cfg = _make(oracle, key, n, seed=21, policy_seed=2, color=color, context=context)
    envs = [oracle.XWorld(pal, render=True, **cfg) for _ in range(n)]
    eps = [0] * n
    for e, w in enumerate(envs):
        w.reset_game(e, 0)
    h, wd, c = sim.screen_dims
    assert (h, wd, c) == envs[0].dims
    for t in range(steps):
        sim.reset_done()
        for e, w in enumerate(envs):
            if w.game_over() != 0:
                eps[e] += 1
                w.reset_game(e, eps[e])
        obs = sim.obs.cpu().numpy()
        for e, w in enumerate(envs):
            assert np.array_equal(obs[e], w.state_screen()), (t, e)
        sim.step()
        acts = sim.actions.cpu().numpy()
        for e, w in enumerate(envs):
            w.take_actions(int(acts[e]))
    obs = sim.obs.cpu().numpy()
    for e, w in enumerate(envs):
        assert np.array_equal(obs[e], w.state_screen())
    sim.close()


@pytest.mark.parametrize("color", [True, False])
def test_tile_table_every_icon_resize_unpinned_by_reference(oracle, color):
    """Each icon's 12x12 tile == the oracle's full render of a map holding only that icon (next to the agent)."""
    _torch()
    sim, pal, cfg = _make(oracle, "nav8", 1, color=color)
    table = sim.tile_table()
    ow = oracle.XWorld(pal, render=True, **cfg)
    agent_icon = int(np.nonzero(pal.type_arr == 2)[0][0])
    for i in range(len(pal)):
        ents = [(2, 0, 0, agent_icon, 0, 0), (int(pal.type_arr[i]) if pal.type_arr[i] != 2 else 1, 3, 5, i, int(pal.name_arr[i]), 1)]
        # the goal list must not be empty for the teacher's idle stage; put one reachable goal at (1, 0)
        goal_icon = int(np.nonzero(pal.type_arr == 0)[0][0])
        ents.append((0, 1, 0, goal_icon, int(pal.name_arr[goal_icon]), 2))
        ow.load_map(ents, 8, target_pick=0)
        scr = ow.screen()
        cell = scr[:, 5 * 12:6 * 12, 3 * 12:4 * 12]
        assert np.array_equal(cell, table[i]), i
        assert (scr[:, 7 * 12:, 7 * 12:] == 255).all()         # empty cell = canvas fill
    sim.close()


class _OracleSample:
    """A scattered sample of a full-size batch's envs run by the oracle in lock step (its own 64 px canvas + cv::resize
    restatement, no tile table): the full-size frames are compared with ORACLE PIXELS, not only through the
    obs == tile_table[grid] property."""

    def __init__(self, oracle, pal, cfg, ids, gid0=0):
        self.ids = list(ids)
        self.gid0 = gid0
        self.envs = [oracle.XWorld(pal, render=True, **cfg) for _ in self.ids]
        self.ep = [0] * len(self.ids)
        for w, e in zip(self.envs, self.ids):
            w.reset_game(gid0 + e, 0)

    def check_frames(self, sim, where):
        import torch
        obs = sim.obs[torch.tensor(self.ids, device="cuda")].cpu().numpy()
        for k, w in enumerate(self.envs):
            exp = w.state_screen()
            assert np.array_equal(obs[k], exp), (where, self.ids[k], int((obs[k] != exp).sum()))

    def reset_done(self):
        for k, w in enumerate(self.envs):
            if w.game_over():
                self.ep[k] += 1
                w.reset_game(self.gid0 + self.ids[k], self.ep[k])

    def step(self, sim):
        acts = sim.actions.cpu().numpy()
        rew = sim.reward.cpu().numpy()
        for k, w in enumerate(self.envs):
            assert np.float32(w.take_actions(int(acts[self.ids[k]]))) == rew[self.ids[k]], self.ids[k]


def test_full_size_c4(oracle):
    """BASELINE config C4: 32 768 envs, 7x7, 84x84x3.  Rewards / codes of every env vs the oracle batch driver; screens of
    every env through the size-independent property obs == tile_table[grid] (table checked above), and of a scattered
    sample of 320 envs against the oracle's own pixels at every step (terminal frames and first frames included)."""
    torch = _torch()
    n, steps = 32768, 24
    sim, pal, cfg = _make(oracle, "nav7", n, seed=5, policy_seed=6, color=True)
    ref = oracle.xw_rollout(n, oracle.xw_cfg(**cfg), pal, steps, policy_seed=6)
    sample = _OracleSample(oracle, pal, cfg, list(range(7, n, 103))[:320])
    table = torch.from_numpy(sim.tile_table()).cuda()
    full = torch.cat([torch.full_like(table[:1], 255), table])          # index 0 = empty cell
    D = 7
    for t in range(steps):
        sim.reset_done()
        sample.reset_done()
        sample.check_frames(sim, ("after reset_done", t))
        sim.step()
        sample.step(sim)
        sample.check_frames(sim, ("after step", t))
        assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), ref.rewards[t].view(np.uint32)), t
        assert np.array_equal(sim.game_over_codes.cpu().numpy(), ref.codes[t]), t
        if t % 8 == 0 or t == steps - 1:
            for lo in range(0, n, 8192):                                # every env, in slabs
                G = sim.grid[lo:lo + 8192].to(torch.int64) & 0x7fff            # bit 15 = target-set flag
                exp = full[G]                                           # [m, D, D, C, 12, 12]
                exp = exp.permute(0, 3, 1, 4, 2, 5).reshape(G.shape[0], 3, 12 * D, 12 * D)
                assert torch.equal(sim.obs[lo:lo + 8192], exp), (t, lo)
    sim.close()


def test_full_size_c5_shard(oracle):
    """BASELINE config C5's per-GPU shard at the size it names: 32 768 envs x 11x11 x 132x132x3 (1.7 GB of frames), the five
    navigation2d.json tasks, env ids of shard 5 of 8 (global ids 163 840 ...).  Rewards / codes vs the oracle batch driver;
    screens through obs == tile_table[grid] over every env (the table itself is checked icon by icon above)."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    n, steps, D, gid0 = 32768, 16, 11, 5 * 32768
    opts = {"xwd_conf_path": os.path.join(CONF, "navigation2d.json"), "task_mode": "lang_acquisition", "max_dim": D,
            "num_blocks": 30, "color": True}
    sim = BatchedSimulator("xworld", opts, num_envs=n, seed=0xC0FFEE, policy_seed=0x5EED, env_gid0=gid0)
    assert sim.obs.shape == (n, 3, 132, 132) and sim.obs.numel() == n * 52272
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    cfg = dict(map_kind=0, max_dim=D, dim=D, num_goals=4, num_blocks=30, color=1, seed=0xC0FFEE, tasks=[0, 1, 2, 3, 4])
    ref = oracle.xw_rollout(n, oracle.xw_cfg(**cfg), pal, steps, policy_seed=0x5EED, env_gid0=gid0)
    sample = _OracleSample(oracle, pal, cfg, list(range(3, n, 257))[:96], gid0)      # oracle pixels for a scattered sample
    table = torch.from_numpy(sim.tile_table()).cuda()
    full = torch.cat([torch.full_like(table[:1], 255), table])          # index 0 = empty cell
    for t in range(steps):
        sim.reset_done()
        sample.reset_done()
        sample.check_frames(sim, ("after reset_done", t))               # the first frames of new episodes, as the C4 test
        sim.step()
        sample.step(sim)
        sample.check_frames(sim, ("after step", t))
        assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), ref.rewards[t].view(np.uint32)), t
        assert np.array_equal(sim.game_over_codes.cpu().numpy(), ref.codes[t]), t
        if t % 5 == 0 or t == steps - 1:
            for lo in range(0, n, 4096):                                # every env, in slabs
                G = sim.grid[lo:lo + 4096].to(torch.int64) & 0x7fff            # bit 15 = target-set flag
                exp = full[G].permute(0, 3, 1, 4, 2, 5).reshape(G.shape[0], 3, 12 * D, 12 * D)
                assert torch.equal(sim.obs[lo:lo + 4096], exp), (t, lo)
    sim.close()


def test_autoreset_then_reset_done_resets_once(oracle):
    """xwb_step_autoreset keeps the codes of the envs it already reset; a following xwb_reset_done only clears them."""
    torch = _torch()
    sim, pal, cfg = _make(oracle, "nav7", 2048, seed=2, policy_seed=8)
    seen = 0
    for t in range(60):
        sim.step_autoreset()
        codes = sim.game_over_codes.clone()
        ep = sim.episode.clone()
        grid = sim.grid.clone()
        sim.reset_done()
        seen += int((codes != 0).sum())
        assert int(sim.game_over_codes.sum()) == 0
        assert torch.equal(sim.episode, ep) and torch.equal(sim.grid, grid), t
    assert seen > 0
    sim.close()
    for name, opts in (("simple_game", {"array_size": 8}), ("simple_race", {"track_width": 20.0, "track_length": 100.0, "track_radius": 30.0})):
        from xworld_amd.batched import BatchedSimulator
        g = BatchedSimulator(name, opts, num_envs=512, policy_seed=3)
        seen = 0
        for t in range(80):
            g.step_autoreset()
            ep = g.episode.clone()
            seen += int((g.game_over_codes != 0).sum())
            g.reset_done()
            assert int(g.game_over_codes.sum()) == 0 and torch.equal(g.episode, ep), (name, t)
        assert seen > 0
        g.close()


def test_autoreset_equals_step_then_reset(oracle):
    torch = _torch()
    n = 4096
    a, _, _ = _make(oracle, "nav8", n, seed=8, policy_seed=9, color=True)
    b, _, _ = _make(oracle, "nav8", n, seed=8, policy_seed=9, color=True)
    for t in range(100):
        a.step_autoreset()
        b.step()
        rb, cb = b.reward.clone(), b.game_over_codes.clone()
        b.reset_done()
        assert torch.equal(a.reward, rb) and torch.equal(a.game_over_codes, cb), t
        assert torch.equal(a.num_steps, b.num_steps), t
        assert torch.equal(a.obs, b.obs), t
    a.close()
    b.close()


def test_sharding_invariance(oracle):
    """Results are keyed by global env id: a batch split in two shards equals the unsplit batch."""
    torch = _torch()
    n = 2048
    whole, _, _ = _make(oracle, "nav8", n, seed=4, policy_seed=4, color=True)
    lo, _, _ = _make(oracle, "nav8", n // 2, seed=4, policy_seed=4, gid0=0, color=True)
    hi, _, _ = _make(oracle, "nav8", n // 2, seed=4, policy_seed=4, gid0=n // 2, color=True)
    for t in range(40):
        for s in (whole, lo, hi):
            s.step_autoreset()
        assert torch.equal(whole.reward, torch.cat([lo.reward, hi.reward]))
        assert torch.equal(whole.game_over_codes, torch.cat([lo.game_over_codes, hi.game_over_codes]))
    assert torch.equal(whole.obs, torch.cat([lo.obs, hi.obs]))
    for s in (whole, lo, hi):
        s.close()


def test_bind_obs_and_masked_reset(oracle):
    torch = _torch()
    n = 512
    sim, pal, cfg = _make(oracle, "nav8", n, seed=12, color=False)
    buf = torch.zeros((2 * n,) + tuple(sim.obs.shape[1:]), dtype=torch.uint8, device="cuda")
    before = sim.obs.clone()
    sim.bind_obs(buf[n:])
    sim.step()
    assert (buf[:n] == 0).all() and not torch.equal(buf[n:], torch.zeros_like(buf[n:]))
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda")
    mask[::3] = 1
    ep0 = np.array([sim.env_state(e).episode for e in range(0, n, 50)])
    sim.reset_masked(mask)
    ep1 = np.array([sim.env_state(e).episode for e in range(0, n, 50)])
    exp = np.array([1 if e % 3 == 0 else 0 for e in range(0, n, 50)])
    assert np.array_equal(ep1 - ep0, exp)
    sim.close()


# ---------------------------------------------------------------- reference golden vectors ----
import json                                                                  # noqa: E402

GOLD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EVENTS = {"": 0, "correct_goal": 1, "wrong_goal": 2, "time_up": 3}
STAGES = {"idle": 0, "navigation_reward": 1, "terminal": 2}


def _grid_from_entities(ents, d):
    g = np.zeros((d, d), np.uint16)
    for t, x, y, icon, name, serial in ents:
        g[y, x] = icon + 1
    return g


@pytest.mark.parametrize("kind", ["nav", "walls"])
def test_reference_teacher_traces_through_product(kind):
    """The reference's own XWorld3DNavTarget task (Python, run in the build container on reference-generated
    maps, tests/golden/teacher.json) vs the HIP step kernel: same maps, same actions ->
    reward (float32 of the Python double), event, stage, agent cell, action success, game_over code."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    with open(os.path.join(GOLD_DIR, "teacher.json")) as f:
        runs = json.load(f)[kind]
    n = len(runs)
    conf = NAV if kind == "nav" else WALLS
    sim = BatchedSimulator("xworld", {"xwd_conf_path": conf, "task_mode": "lang_acquisition"}, num_envs=n)
    d = sim.cfg.max_dim
    for e, run in enumerate(runs):
        assert run["max_dim"] == d and run["dim"] == d
        agent = [x for x in run["entities"] if x[0] == 2][0]
        sim.load_map(e, _grid_from_entities(run["entities"], d), agent[1], agent[2], run["target_name"])
    T = max(len(r["trace"]) for r in runs)
    for t in range(T):
        acts = np.array([r["trace"][t][0] if t < len(r["trace"]) else 0 for r in runs], np.int32)
        sim.step(torch.from_numpy(acts).cuda())
        rew = sim.reward.cpu().numpy()
        codes = sim.game_over_codes.cpu().numpy()
        for e, r in enumerate(runs):
            if t >= len(r["trace"]):
                continue
            a, reward, event, stage, ax, ay, success = r["trace"][t]
            st = sim.env_state(e)
            assert rew[e] == np.float32(reward), (e, t)
            assert st.xw_event == EVENTS[event] and st.xw_stage == STAGES[stage], (e, t)
            assert (st.xw_agent_x, st.xw_agent_y) == (ax, ay) and st.last_action_success == success, (e, t)
            assert codes[e] == {0: 0, 1: 4, 2: 2, 3: 1}[EVENTS[event]]
    sim.close()


def test_reference_maps_render_through_product(oracle):
    """Reference-generated XWorldNav maps: product screen == oracle canvas render of the same entity list."""
    _torch()
    from xworld_amd.batched import BatchedSimulator
    with open(os.path.join(GOLD_DIR, "maps_nav.json")) as f:
        maps = json.load(f)[:16]
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    sim = BatchedSimulator("xworld", {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "color": True},
                           num_envs=len(maps))
    ow = oracle.XWorld(pal, render=True, map_kind=0, max_dim=8, dim=8, num_goals=4, color=1)
    for e, m in enumerate(maps):
        agent = [x for x in m["entities"] if x[0] == 2][0]
        sim.load_map(e, _grid_from_entities(m["entities"], 8), agent[1], agent[2], 0)
        ow.load_map([tuple(x) for x in m["entities"]], 8, target_pick=0)
        assert np.array_equal(sim.env_obs(e).reshape(3, 96, 96), ow.screen()), e
    sim.close()


def test_action_skip_and_reset_env(oracle):
    """One slot of a batch stepped / reset alone (context ring of the other envs must not move)."""
    torch = _torch()
    import ctypes as C
    from xworld_amd import lib
    n = 64
    sim, pal, cfg = _make(oracle, "nav8", n, seed=77, context=2)
    w = oracle.XWorld(pal, render=True, **cfg)
    w.reset_game(5, 0)
    before = sim.obs.clone()
    acts = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    for a in (0, 3, 3, 1, 2):
        acts[5] = a
        sim.step(acts)
        assert np.float32(float(sim.reward[5])) == np.float32(w.take_actions(a))
    assert np.array_equal(sim.obs[5].cpu().numpy(), w.state_screen())
    keep = torch.ones(n, dtype=torch.bool, device="cuda")
    keep[5] = False
    assert torch.equal(sim.obs[keep], before[keep])
    lib.check(sim.L.xwb_reset_env(sim.h, 5, None))
    w.reset_game(5, 1)
    assert np.array_equal(sim.obs[5].cpu().numpy(), w.state_screen())
    assert np.array_equal(sim.env_grid(5).astype(np.int32), w.grid())
    assert torch.equal(sim.obs[keep], before[keep])
    sim.close()


@pytest.mark.parametrize("key,color,context", [("nav7", True, 1), ("nav8", False, 1), ("nav11", True, 1), ("walls7", True, 3),
                                               ("nav8_dim5", False, 2)])
def test_float32_observations(oracle, key, color, context):
    """obs_format="float32" (SURVEY 8(d) variant): every frame equals the uint8 frame * float32(1/255), the product
    py_simulator.cpp:262-272 computes in get_state(); same trajectories, resets and context ring."""
    torch = _torch()
    n = 700
    a, _, _ = _make(oracle, key, n, seed=12, policy_seed=4, color=color, context=context)
    b, _, _ = _make(oracle, key, n, seed=12, policy_seed=4, color=color, context=context, obs_format="float32")
    assert b.obs.dtype == torch.float32 and b.obs.shape == a.obs.shape and b.obs_bytes_per_env == 4 * a.obs_bytes_per_env
    scale = torch.tensor(1 / 255.0, dtype=torch.float32, device="cuda")
    for t in range(60):
        assert torch.equal(b.obs, a.obs.to(torch.float32) * scale), t
        if t % 2:
            a.step_autoreset(); b.step_autoreset()
        else:
            a.step(); b.step()
            assert torch.equal(b.obs, a.obs.to(torch.float32) * scale), t
            a.reset_done(); b.reset_done()
        assert torch.equal(a.reward, b.reward) and torch.equal(a.game_over_codes, b.game_over_codes)
    e = n // 2
    assert np.array_equal(b.env_obs(e), a.env_obs(e).astype(np.float32) * np.float32(1 / 255.0))
    a.close()
    b.close()


@pytest.mark.parametrize("md,blocks", [(15, 60), (16, 70), (13, 40)])
def test_largest_maps(oracle, md, blocks):
    """max_dim up to 16 (four 64-bit mask words per board, the generic-dimension render): reset, rollout, frames."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    n, steps = 256, 120
    opts = {"xwd_conf_path": os.path.join(CONF, "navigation2d.json"), "task_mode": "lang_acquisition", "max_dim": md,
            "num_blocks": blocks, "color": True}
    sim = BatchedSimulator("xworld", opts, num_envs=n, seed=3, policy_seed=4)
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    cfg = dict(map_kind=0, max_dim=md, dim=md, num_goals=4, num_blocks=blocks, color=1, seed=3, tasks=[0, 1, 2, 3, 4])
    ow = oracle.XWorld(pal, render=True, **cfg)
    obs = sim.obs.cpu().numpy()
    for e in range(0, n, 5):
        ow.reset_game(e, 0)
        st = sim.env_state(e)
        assert np.array_equal(sim.env_grid(e).astype(np.int32), ow.grid()) and (st.xw_agent_x, st.xw_agent_y) == ow.agent_xy()
        assert st.xw_task == ow.task_kind()
        if e % 25 == 0:
            assert np.array_equal(obs[e], ow.state_screen()), e
    ref = oracle.xw_rollout(n, oracle.xw_cfg(**cfg), pal, steps, policy_seed=4)
    for t in range(steps):
        sim.reset_done()
        sim.step()
        assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), ref.rewards[t].view(np.uint32)), t
        assert np.array_equal(sim.game_over_codes.cpu().numpy(), ref.codes[t]), t
    sim.close()

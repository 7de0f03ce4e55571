"""GPU parity for the egocentric mode (visible_radius > 0): six first-person actions, headings, the teacher's rule
along the heading, and the egocentric frames (shadow casting, goal warps, view rotation, the two resizes) -- through
the C ABI against the CPU oracle and the reference's own task traces (tests/golden/tasks_ego.json)."""
import os

import numpy as np
import pytest

from test_gpu_xworld import MAPS, _torch
from test_oracle_ego import ego_runs
from test_oracle_tasks import EVENTS, KINDS, STAGES

pytestmark = pytest.mark.gpu

CONF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xworld_amd", "confs")
FACING = {0: 0.0, 1: np.pi / 2, 2: np.pi, 3: -np.pi / 2}


def _facing(yaw):
    eps = 1e-4
    return 0 if abs(yaw) < eps else (1 if abs(yaw - np.pi / 2) < eps else (2 if abs(yaw - np.pi) < eps else 3))


def _make(oracle, key, n, r, tasks=KINDS, seed=0xC0FFEE, policy_seed=0x5EED, gid0=0, **opts):
    from xworld_amd.batched import BatchedSimulator
    conf, popts, ocfg = MAPS[key]
    o = {"xwd_conf_path": conf, "task_mode": "lang_acquisition", "tasks": list(tasks), "visible_radius": r}
    o.update(popts)
    o.update(opts)
    sim = BatchedSimulator("xworld", o, num_envs=n, seed=seed, policy_seed=policy_seed, env_gid0=gid0)
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    cfg = dict(ocfg)
    cfg.update(seed=seed, tasks=list(tasks), visible_radius=r, color=int(bool(opts.get("color", False))),
               context=int(opts.get("context", 1)), max_steps=int(opts.get("max_steps", 0)))
    return sim, pal, cfg


@pytest.mark.parametrize("key,r", [("nav8", 3), ("nav7", 5), ("nav11", 7), ("nav8_dim5", 3), ("nav7", 1)])
def test_ego_reset_and_rollout(oracle, key, r):
    """Reset parity (map, heading, task, target sets) and random-policy rollouts with resets: reward bits and codes."""
    _torch()
    n, steps = 768, 200
    sim, pal, cfg = _make(oracle, key, n, r, seed=17, policy_seed=3, gid0=40)
    assert sim.num_actions == 6 and sim.screen_dims[:2] == (r * (84 // r), r * (84 // r))
    ow = oracle.XWorld(pal, render=False, **cfg)
    dirs = np.zeros(4, int)
    for e in range(0, n, 2):
        ow.reset_game(40 + e, 0)
        st = sim.env_state(e)
        raw = sim.env_grid(e, raw=True)
        assert np.array_equal((raw & 0x7fff).astype(np.int32), ow.grid()), e
        assert (st.xw_agent_x, st.xw_agent_y) == ow.agent_xy() and st.xw_task == ow.task_kind(), e
        assert st.xw_agent_dir == _facing(ow.agent_yaw()), e
        if st.xw_task != 3:
            assert np.array_equal((raw >> 15).astype(np.uint8), ow.target_cells()), (e, st.xw_task)
        dirs[st.xw_agent_dir] += 1
    assert dirs.min() > 50
    ref = oracle.xw_rollout(n, oracle.xw_cfg(**cfg), pal, steps, policy_seed=3, env_gid0=40)
    resets = 0
    for t in range(steps):
        sim.reset_done()
        resets += sim.done_count()
        sim.step()
        assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), ref.rewards[t].view(np.uint32)), t
        assert np.array_equal(sim.game_over_codes.cpu().numpy(), ref.codes[t]), t
    assert resets == ref.stats.resets and resets > 0
    sim.close()


@pytest.mark.parametrize("kind", KINDS)
def test_reference_ego_traces_through_product(oracle, kind):
    """tests/golden/tasks_ego.json: the reference's Python tasks with visible_radius = 3, replayed through the product's
    step kernel (map after the idle stage, heading and target set loaded)."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    runs = ego_runs(kind)
    md, dim = runs[0]["max_dim"], runs[0]["dim"]
    assert all(r["max_dim"] == md and r["dim"] == dim for r in runs)
    n = len(runs)
    sim = BatchedSimulator("xworld", {"xwd_conf_path": os.path.join(CONF, "nav_target.json"), "max_dim": md, "dim": dim,
                                      "task_mode": "lang_acquisition", "tasks": [kind], "visible_radius": 3}, num_envs=n)
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    for e, run in enumerate(runs):
        # the oracle replays the idle stage (pinned to the same file by test_oracle_ego.py) and tells what Direction's
        # (referent, direction) is
        w = oracle.XWorld(pal, render=False, map_kind=0, max_dim=md, dim=dim, num_goals=4, tasks=[kind], visible_radius=3)
        w.stage_poses(run["poses"])
        w.load_map_ex([tuple(x) for x in run["entities_before"]], dim, [0] + run["decisions"])
        g = np.zeros((md, md), np.uint16)
        agent = None
        for t, x, y, icon, name, serial in run["entities_after"]:
            g[y, x] = icon + 1
            if t == 2:
                agent = (x, y)
        if kind != "XWorld3DNavTargetDirection":
            for x, y in run["target_cells"]:
                g[y, x] |= 0x8000
        target = -1
        if kind == "XWorld3DNavTargetBetween":
            target = run["between"][1] * md + run["between"][0]
        elif kind == "XWorld3DNavTargetDirection":
            rx, ry, word = w.direction_target()
            target = (ry * md + rx) | (word << 8)
        sim.load_map(e, g, agent[0], agent[1], dim=dim, task=kind, target=target)
        sim.set_agent_dir(e, _facing(w.agent_yaw()))
    seen = set()
    T = max(len(r["trace"]) for r in runs)
    for t in range(T):
        acts = np.full(n, -1, np.int32)
        for e, run in enumerate(runs):
            if t < len(run["trace"]):
                acts[e] = run["trace"][t][0]
        sim.step(torch.from_numpy(acts).cuda())
        rew = sim.reward.cpu().numpy()
        for e, run in enumerate(runs):
            if t >= len(run["trace"]):
                continue
            a, reward, event, stage, ax, ay, success, yaw = run["trace"][t]
            st = sim.env_state(e)
            assert rew[e] == np.float32(reward), (run["py_seed"], t)
            assert st.xw_event == EVENTS[event] and st.xw_stage == STAGES[stage], (run["py_seed"], t)
            assert (st.xw_agent_x, st.xw_agent_y) == (ax, ay) and st.last_action_success == success
            assert st.xw_agent_dir == _facing(yaw), (run["py_seed"], t)
            seen.add(event)
    assert {"correct_goal", "wrong_goal"} <= seen
    sim.close()


@pytest.mark.parametrize("key,r,color,context", [("nav7", 3, True, 1), ("nav8", 5, False, 1), ("nav7", 7, True, 2),
                                                 ("nav11", 3, True, 1), ("nav8_dim5", 1, True, 1),
                                                 # 81 and 77 pixel edges: frames that are not whole 16-byte chunks
                                                 pytest.param("nav11", 9, True, 2, marks=pytest.mark.slow), ("nav11", 11, False, 1)])
def test_ego_frames_with_host_poses(oracle, key, r, color, context):
    """Frames bit for bit: maps from the oracle's generator are loaded into the product with the goal poses set through
    the host (same libm as the oracle), then both run the same action strings."""
    torch = _torch()
    n, steps = 48, 14
    sim, pal, cfg = _make(oracle, key, n, r, tasks=[KINDS[0]], seed=5, color=color, context=context)
    md = cfg["max_dim"]
    envs = []
    for e in range(n):
        w = oracle.XWorld(pal, render=True, **cfg)
        w.reset_game(e, 0)
        envs.append(w)
        g = w.grid().astype(np.uint16)
        ax, ay = w.agent_xy()
        tc = w.target_cells()
        g[tc != 0] |= 0x8000
        sim.load_map(e, g, ax, ay, dim=cfg["dim"], task=KINDS[0], target=w.target_name())
        sim.set_agent_dir(e, _facing(w.agent_yaw()))
        for i, ent in enumerate(w.entities()):
            if ent[0] == 0:
                yaw, scale, offset = w.get_pose(i)
                sim.set_goal_pose(e, ent[1], ent[2], yaw, scale, offset)
        sim.refresh_obs(e)
    rng = np.random.default_rng(1)
    for t in range(steps):
        obs = sim.obs.cpu().numpy()
        for e, w in enumerate(envs):
            exp = w.state_screen()
            assert np.array_equal(obs[e], exp), (t, e, int((obs[e] != exp).sum()))
        acts = rng.integers(0, 6, n).astype(np.int32)
        sim.step(torch.from_numpy(acts).cuda())
        for e, w in enumerate(envs):
            w.take_actions(int(acts[e]))
    sim.close()


def test_ego_frames_device_poses(oracle, trig):
    """The same with poses drawn by the reset kernel: its warp matrix comes from include/xwb_trig.h's cos / sin, the
    oracle's from the host's libm (trig = libm: a checker that shares no arithmetic with the kernel) or from the same
    header (trig = xwb_trig).  Frames are equal byte for byte either way -- the warp narrows the matrix to 1/1024-pixel
    fixed point --; differing envs / pixels are counted over the whole batch and reported."""
    _torch()
    n = 2048
    sim, pal, cfg = _make(oracle, "nav7", n, 3, seed=23, color=True)
    ow = oracle.XWorld(pal, render=True, **cfg)
    obs = sim.obs.cpu().numpy()
    bad_envs = bad_px = 0
    for e in range(n):
        ow.reset_game(e, 0)
        d = int((obs[e] != ow.state_screen()).sum())
        bad_envs += d > 0
        bad_px += d
    sim.close()
    assert bad_envs == 0, "trig=%s: %d of %d first frames differ (%d bytes)" % (trig, bad_envs, n, bad_px)


def test_ego_rollout_frames_device_poses(oracle, trig):
    """... and through a rollout with resets: every frame of 128 envs over 64 steps (each reset draws new goal poses on the
    device), rewards and codes, against the libm oracle and the xwb_trig one."""
    torch = _torch()
    n, steps = 128, 64
    sim, pal, cfg = _make(oracle, "nav7", n, 3, seed=29, policy_seed=6, color=True)
    envs = [oracle.XWorld(pal, render=True, **cfg) for _ in range(n)]
    ep = [0] * n
    for e, w in enumerate(envs):
        w.reset_game(e, 0)
    bad_frames = bad_px = resets = 0
    for t in range(steps):
        obs = sim.obs.cpu().numpy()
        for e, w in enumerate(envs):
            d = int((obs[e] != w.state_screen()).sum())
            bad_frames += d > 0
            bad_px += d
        sim.step()
        acts = sim.actions.cpu().numpy()
        rew = sim.reward.cpu().numpy()
        codes = sim.game_over_codes.cpu().numpy()
        for e, w in enumerate(envs):
            assert np.float32(w.take_actions(int(acts[e]))) == rew[e] and w.game_over() == codes[e], (t, e)
        sim.reset_done()
        for e, w in enumerate(envs):
            if codes[e]:
                ep[e] += 1
                resets += 1
                w.reset_game(e, ep[e])
    sim.close()
    assert resets > 8
    assert bad_frames == 0, "trig=%s: %d of %d frames differ (%d bytes)" % (trig, bad_frames, n * steps, bad_px)


def test_ego_curriculum_fewer_goals_than_levels_place(oracle):
    """FLAGS_curriculum with num_goals = 2 in the conf and visible_radius > 0: the levels place 2 or 4 goals whatever the
    option says (XWorldNav.py:27-34), so the per-env goal-image cache holds 4 slots; frames of envs at every level
    against the oracle (start_level 3 -> 4 goals at once)."""
    _torch()
    for start in (0, 3, 5):
        n = 96
        sim, pal, cfg = _make(oracle, "nav8", n, 3, tasks=[KINDS[0]], seed=31, color=True, num_goals=2, curriculum=0.1,
                              start_level=start)
        cfg.update(curriculum=0.1, start_level=start)
        ow = oracle.XWorld(pal, render=True, **cfg)
        obs = sim.obs.cpu().numpy()
        for e in range(n):
            ow.reset_game(e, 0)
            assert sim.env_state(e).xw_level == start
            assert np.array_equal(obs[e], ow.state_screen()), (start, e)
        sim.close()


def test_ego_autoreset_skip_and_context(oracle):
    """step_autoreset == step + reset_done also in egocentric mode (frames, rewards, codes, headings); XWB_ACTION_SKIP
    leaves an env and its frames untouched; the context ring shifts."""
    torch = _torch()
    n = 1024
    a, _, _ = _make(oracle, "nav7", n, 3, seed=4, policy_seed=2, color=True, context=2)
    b, _, _ = _make(oracle, "nav7", n, 3, seed=4, policy_seed=2, color=True, context=2)
    for t in range(60):
        a.step_autoreset()
        b.step()
        rb, cb = b.reward.clone(), b.game_over_codes.clone()
        b.reset_done()
        assert torch.equal(a.reward, rb) and torch.equal(a.game_over_codes, cb), t
        assert torch.equal(a.obs, b.obs) and torch.equal(a.num_steps, b.num_steps), t
    before = a.obs.clone()
    acts = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    acts[5] = 4                                            # one env turns left
    d0 = a.env_state(5).xw_agent_dir
    a.step(acts)
    assert a.env_state(5).xw_agent_dir == (d0 + 3) % 4 and a.env_state(5).last_action_success == 0
    keep = torch.ones(n, dtype=torch.bool, device="cuda")
    keep[5] = False
    assert torch.equal(a.obs[keep], before[keep])
    assert torch.equal(a.obs[5, :3], before[5, 3:]) and not torch.equal(a.obs[5, 3:], before[5, 3:])
    a.close()
    b.close()


@pytest.mark.parametrize("key,r,opts", [("nav7", 3, dict(color=True)), ("nav8", 3, dict(color=False, context=2)),
                                        ("nav7", 5, dict(color=True, obs_format="float32")), ("nav11", 7, dict(color=True, context=3))])
def test_ego_span_path_equals_per_env_path(oracle, key, r, opts):
    """The two egocentric renders (the span path -- cells, evaluated pixels, gather -- and one workgroup per env) draw the same
    frames through every verb: step + reset_done, step_autoreset, masked resets; context rings and float32 frames included."""
    torch = _torch()
    n = 700                                                # not a multiple of the kernels' env groups (64, 8)
    a, _, _ = _make(oracle, key, n, r, seed=11, policy_seed=3, **opts)
    b, _, _ = _make(oracle, key, n, r, seed=11, policy_seed=3, debug=["ego_no_span"], **opts)     # xwb_config.debug_flags
    assert a.ego_render_path == "span" and b.ego_render_path == "per_env"
    for sim in (a, b):
        sim.reset()
    assert torch.equal(a.obs, b.obs)
    mask = (torch.arange(n, device="cuda") % 7 == 3)
    for t in range(50):
        for sim in (a, b):
            if t % 3 == 2:
                sim.step_autoreset()
            else:
                sim.step()
        assert torch.equal(a.obs, b.obs), ("terminal / stepped frames", t)
        for sim in (a, b):
            if t % 3 != 2:
                sim.reset_done()
            if t == 20:
                sim.reset_masked(mask)
        assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward) and torch.equal(a.game_over_codes, b.game_over_codes), t
    a.close()
    b.close()


@pytest.mark.parametrize("key,r,steps", [("nav8_dim5", 3, 160), ("nav7", 5, 120)])
def test_ego_goal_lines_cached_across_revisits(oracle, key, r, steps):
    """Round 5: a goal's cache entry also holds the border lines that blend its image (its own first row / column, the first row
    of the square below, the first column of the square to the right, the crossing pixel below right), evaluated ONCE per (goal slot,
    view cell, heading) on the view of that moment and reused whenever the key comes back.  On a small map the agent keeps coming
    back to the same cells with goals next to each other, next to blocks and next to the map's edge: long episodes of the span path
    against the one-workgroup-per-env kernel, which evaluates every such pixel on every frame."""
    torch = _torch()
    n = 520
    a, _, _ = _make(oracle, key, n, r, seed=29, policy_seed=31, color=True, max_steps=400)
    b, _, _ = _make(oracle, key, n, r, seed=29, policy_seed=31, color=True, max_steps=400, debug=["ego_no_span"])
    assert a.ego_render_path == "span" and b.ego_render_path == "per_env"
    for sim in (a, b):
        sim.reset()
    for t in range(steps):
        for sim in (a, b):
            sim.step()
        assert torch.equal(a.obs, b.obs), ("stepped frames", t)
        for sim in (a, b):
            sim.reset_done()
        assert torch.equal(a.obs, b.obs), ("first frames", t)
    a.close()
    b.close()


def test_ego_render_path_by_geometry(oracle):
    """r = 3, 5, 7: the frame is r x r equal squares -> span path; r = 1 and r >= 9 (81, 77 pixel edges): one workgroup per env."""
    _torch()
    for key, r, want in (("nav7", 3, "span"), ("nav7", 5, "span"), ("nav11", 7, "span"), ("nav7", 1, "per_env"), ("nav11", 9, "per_env")):
        sim, _, _ = _make(oracle, key, 8, r)
        assert sim.ego_render_path == want, (key, r)
        sim.close()


def test_ego_config_errors():
    _torch()
    from xworld_amd.batched import BatchedSimulator
    from xworld_amd.lib import XwbError
    nav = os.path.join(CONF, "nav_target.json")
    with pytest.raises(XwbError):                          # xmap.cpp:277: must be odd
        BatchedSimulator("xworld", {"xwd_conf_path": nav, "visible_radius": 4}, num_envs=4)
    with pytest.raises(XwbError):                          # the reference asserts for maps without maze generation
        BatchedSimulator("xworld", {"xwd_conf_path": os.path.join(CONF, "walls_target.json"), "visible_radius": 3}, num_envs=4)
    with pytest.raises(XwbError):                          # clamped to the map (8): even -> CHECK fails
        BatchedSimulator("xworld", {"xwd_conf_path": nav, "visible_radius": 99}, num_envs=4)
    sim = BatchedSimulator("xworld", {"xwd_conf_path": nav, "visible_radius": 99, "max_dim": 7}, num_envs=4)
    assert sim.cfg.visible_radius == 99 and sim.screen_dims[:2] == (84, 84) and sim.num_actions == 6   # clamped to 7 inside
    sim.close()


@pytest.mark.parametrize("key,r", [("nav7", 3), ("nav11", 9)])
def test_ego_float32_frames(oracle, key, r):
    """obs_format="float32" in egocentric mode: every frame = the uint8 frame * float32(1/255), through resets and the ring."""
    torch = _torch()
    n = 300
    a, _, _ = _make(oracle, key, n, r, seed=6, policy_seed=5, color=True, context=2)
    b, _, _ = _make(oracle, key, n, r, seed=6, policy_seed=5, color=True, context=2, obs_format="float32")
    assert b.obs.dtype == torch.float32 and b.obs_bytes_per_env == 4 * a.obs_bytes_per_env
    scale = torch.tensor(1 / 255.0, dtype=torch.float32, device="cuda")
    for t in range(40):
        assert torch.equal(b.obs, a.obs.to(torch.float32) * scale), t
        if t % 2:
            a.step_autoreset(); b.step_autoreset()
        else:
            a.step(); b.step()
            assert torch.equal(b.obs, a.obs.to(torch.float32) * scale), t
            a.reset_done(); b.reset_done()
    a.close()
    b.close()


def test_ego_without_wall_shadow(oracle):
    """FLAGS_wall_shadow = false (xmap.cpp:19,170): cells behind walls stay visible; frames against the oracle, and they
    differ from the shadowed ones where a wall hides something."""
    torch = _torch()
    n = 64
    a, pal, cfg = _make(oracle, "nav7", n, 3, tasks=[KINDS[0]], seed=5, color=True, wall_shadow=False)
    b, _, _ = _make(oracle, "nav7", n, 3, tasks=[KINDS[0]], seed=5, color=True)
    cfg["no_wall_shadow"] = 1
    envs = []
    for e in range(n):
        w = oracle.XWorld(pal, render=True, **cfg)
        w.reset_game(e, 0)
        envs.append(w)
        g = w.grid().astype(np.uint16)
        ax, ay = w.agent_xy()
        g[w.target_cells() != 0] |= 0x8000
        for sim in (a, b):
            sim.load_map(e, g, ax, ay, dim=cfg["dim"], task=KINDS[0], target=w.target_name())
            sim.set_agent_dir(e, _facing(w.agent_yaw()))
            for i, ent in enumerate(w.entities()):
                if ent[0] == 0:
                    sim.set_goal_pose(e, ent[1], ent[2], *w.get_pose(i))
            sim.refresh_obs(e)
    rng = np.random.default_rng(2)
    differ = 0
    for t in range(10):
        oa, ob = a.obs.cpu().numpy(), b.obs.cpu().numpy()
        for e, w in enumerate(envs):
            assert np.array_equal(oa[e], w.state_screen()), (t, e)
        differ += int((oa != ob).any(axis=(1, 2, 3)).sum())
        acts = rng.integers(0, 6, n).astype(np.int32)
        for sim in (a, b):
            sim.step(torch.from_numpy(acts).cuda())
        for e, w in enumerate(envs):
            w.take_actions(int(acts[e]))
    assert differ > 0
    a.close()
    b.close()

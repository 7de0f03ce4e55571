"""Edge cases the reference's behaviour defines but its tests never reach: ragged batch sizes (not multiples of
the 64-lane wavefront / 256-thread workgroup), one-env batches, the smallest and largest maps, deep context
rings, gray + colour, every map class -- all bit-exact against the oracle through the C ABI."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAV = os.path.join(ROOT, "xworld_amd", "confs", "nav_target.json")
WALLS = os.path.join(ROOT, "xworld_amd", "confs", "walls_target.json")
GOLD = -7046029254386353131


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("n", [1, 2, 63, 65, 255, 257, 1000])
def test_ragged_batches_simple_games(oracle, n):
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    steps = 90
    ref = oracle.sg_rollout(n, 12, steps, policy_seed=3, env_gid0=9, context=2)
    sim = BatchedSimulator("simple_game", {"array_size": 12, "context": 2}, num_envs=n, policy_seed=3, env_gid0=9)
    w = torch.arange(1, 25, dtype=torch.int64, device="cuda") * GOLD
    for t in range(steps):
        sim.reset_done()
        ck = (sim.obs.reshape(n, 24).to(torch.int64) * w[None, :]).sum(1)
        assert np.array_equal(ck.cpu().numpy().view(np.uint64), ref.obs_ck[t])
        sim.step()
        assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), ref.rewards[t].view(np.uint32))
        assert np.array_equal(sim.game_over_codes.cpu().numpy(), ref.codes[t])
    sim.close()
    cfg = oracle.race_cfg(random=1, track_type=1)
    ref = oracle.race_rollout(n, cfg, seed=5, steps=steps, policy_seed=6)
    sim = BatchedSimulator("simple_race", {"track_type": "circle", "random": True, "track_width": 20.0,
                                           "track_length": 100.0, "track_radius": 30.0}, num_envs=n, seed=5, policy_seed=6)
    for t in range(steps):
        sim.reset_done()
        sim.step()
        assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), ref.rewards[t].view(np.uint32))
        assert np.array_equal(sim.game_over_codes.cpu().numpy(), ref.codes[t])
    sim.close()


def _xw(oracle, n, conf, popts, ocfg, seed, policy_seed, **opts):
    from xworld_amd.batched import BatchedSimulator
    o = {"xwd_conf_path": conf, "task_mode": "lang_acquisition"}
    o.update(popts)
    o.update(opts)
    sim = BatchedSimulator("xworld", o, num_envs=n, seed=seed, policy_seed=policy_seed)
    pal = oracle.Palette(oracle.NAV_SUBTREES if ocfg["map_kind"] == 0 else oracle.WALLS_SUBTREES)
    cfg = dict(ocfg)
    cfg.update(seed=seed, color=int(bool(opts.get("color", False))), context=int(opts.get("context", 1)))
    return sim, pal, cfg


XW_CASES = [
    # n, conf, product opts, oracle cfg, extra
    (1, NAV, {}, dict(map_kind=0, max_dim=8, dim=8, num_goals=4, num_blocks=16), dict(color=True)),
    (3, NAV, {"max_dim": 3, "num_goals": 2, "num_blocks": 0}, dict(map_kind=0, max_dim=3, dim=3, num_goals=2, num_blocks=0), dict(context=4)),
    (65, NAV, {"max_dim": 16, "num_goals": 9, "num_blocks": 60}, dict(map_kind=0, max_dim=16, dim=16, num_goals=9, num_blocks=60), dict(color=False)),
    (17, NAV, {"max_dim": 13, "num_goals": 4, "num_blocks": 40}, dict(map_kind=0, max_dim=13, dim=13, num_goals=4, num_blocks=40), dict(color=True, context=2)),
    (130, NAV, {"max_dim": 9, "dim": 6, "num_goals": 3, "num_blocks": 10}, dict(map_kind=0, max_dim=9, dim=6, num_goals=3, num_blocks=10), dict(color=True)),
    (33, WALLS, {"max_dim": 10, "num_goals": 16, "num_blocks": 14}, dict(map_kind=1, max_dim=10, dim=10, num_goals=16, num_blocks=14), dict(color=True)),
    (257, WALLS, {}, dict(map_kind=1, max_dim=7, dim=7, num_goals=12, num_blocks=12), dict(context=2)),
]


@pytest.mark.parametrize("case", XW_CASES, ids=[str(i) for i in range(len(XW_CASES))])
def test_xworld_sizes_and_maps(oracle, case):
    """Maps from 3x3 to 16x16 (1, 2 and 4 mask words in the reset kernel; templated and generic render), padded
    curriculum dims, both map classes; reset, rollout rewards/codes and the final screens."""
    _torch()
    n, conf, popts, ocfg, extra = case
    sim, pal, cfg = _xw(oracle, n, conf, popts, ocfg, seed=31, policy_seed=32, **extra)
    steps = 60
    ref = oracle.xw_rollout(n, oracle.xw_cfg(**cfg), pal, steps, policy_seed=32)
    ow = oracle.XWorld(pal, render=False, **cfg)
    for e in range(min(n, 40)):
        ow.reset_game(e, 0)
        assert np.array_equal(sim.env_grid(e).astype(np.int32), ow.grid()), e
    for t in range(steps):
        sim.reset_done()
        sim.step()
        assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), ref.rewards[t].view(np.uint32)), t
        assert np.array_equal(sim.game_over_codes.cpu().numpy(), ref.codes[t]), t
    # screens of a few envs after the rollout: replay those envs in the canvas oracle
    acts_hist = None
    sim.close()
    sim, pal, cfg = _xw(oracle, n, conf, popts, ocfg, seed=31, policy_seed=32, **extra)
    envs = {e: oracle.XWorld(pal, render=True, **cfg) for e in sorted({0, n // 2, n - 1})}
    eps = {e: 0 for e in envs}
    for e, w in envs.items():
        w.reset_game(e, 0)
    for t in range(12):
        sim.reset_done()
        for e, w in envs.items():
            if w.game_over():
                eps[e] += 1
                w.reset_game(e, eps[e])
        sim.step()
        acts = sim.actions.cpu().numpy()
        for e, w in envs.items():
            w.take_actions(int(acts[e]))
    obs = sim.obs.cpu().numpy()
    for e, w in envs.items():
        assert np.array_equal(obs[e], w.state_screen()), e
    sim.close()


def test_create_rejects_impossible_configs():
    _torch()
    from xworld_amd import lib
    from xworld_amd.batched import BatchedSimulator
    with pytest.raises(lib.XwbError, match="too many blocks"):
        BatchedSimulator("xworld", {"xwd_conf_path": NAV, "max_dim": 7, "num_blocks": 19}, num_envs=4)
    with pytest.raises(lib.XwbError, match="max_dim"):
        BatchedSimulator("xworld", {"xwd_conf_path": NAV, "max_dim": 17}, num_envs=4)
    with pytest.raises(lib.XwbError, match="num_envs"):
        BatchedSimulator("simple_game", {"array_size": 6}, num_envs=0)
    with pytest.raises(lib.XwbError, match="context"):
        BatchedSimulator("simple_game", {"array_size": 6, "context": 0}, num_envs=1)


def test_user_stream_and_state_packet(oracle):
    """Calls on a non-default HIP stream; get_state of one env as reference wire bytes."""
    torch = _torch()
    import ctypes as C
    from xworld_amd.batched import BatchedSimulator
    st = torch.cuda.Stream()
    sim = BatchedSimulator("simple_game", {"array_size": 16}, num_envs=512, policy_seed=1)
    ref = oracle.sg_rollout(512, 16, 50, policy_seed=1)
    for t in range(50):
        sim.reset_done(stream=st)
        sim.step(stream=st)
        st.synchronize()
        assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), ref.rewards[t].view(np.uint32))
    raw = sim.state_packet(env=7, reward=1.25, stream=st)
    L = oracle.lib()
    buf = np.frombuffer(raw, np.uint8).copy()
    out = (oracle.PacketField * 4)()
    assert L.orc_packet_decode(oracle.ptr(buf, oracle.u8p), len(raw), out, 4) == 2
    keys = {out[i].key: out[i] for i in range(2)}
    assert set(keys) == {b"reward", b"screen"}
    assert keys[b"reward"].has_reals and keys[b"reward"].reals[0] == 1.25
    scr = keys[b"screen"]
    assert scr.has_pixels and scr.n_pixels == 16
    assert [scr.pixels[i] for i in range(16)] == sim.env_obs(7).tolist()
    sim.close()


@pytest.mark.parametrize("game,opts", [("simple_game", {"array_size": 16}),
                                        ("simple_race", {"track_width": 20.0, "track_length": 100.0, "track_radius": 30.0}),
                                        ("xworld", None)])
def test_bind_results_packed_output(game, opts):
    """xwb_bind_results: every step also writes (reward, game_over code) of the stepped envs into one caller buffer."""
    import os
    import torch
    assert torch.cuda.is_available()
    from xworld_amd.batched import BatchedSimulator
    if opts is None:
        conf = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xworld_amd", "confs", "navigation2d.json")
        opts = {"xwd_conf_path": conf, "task_mode": "lang_acquisition"}
    n = 2048
    sim = BatchedSimulator(game, opts, num_envs=n, policy_seed=3)
    bufs = [torch.full((n, 2), -7.0, device="cuda") for _ in range(2)]
    for t in range(60):
        sim.bind_results(bufs[t & 1])
        sim.step()
        assert torch.equal(bufs[t & 1][:, 0], sim.reward) and torch.equal(bufs[t & 1][:, 1], sim.game_over_codes.float()), t
        sim.reset_done()
        assert torch.equal(bufs[t & 1][:, 1] != 0, bufs[t & 1][:, 1] != 0)
    sim.bind_results(None)
    keep = bufs[1].clone()
    sim.step()
    assert torch.equal(bufs[1], keep)
    sim.close()


def test_batch_copy_out_functions():
    """xwb_get_obs / xwb_get_reward / xwb_get_done: the whole batch's outputs into caller-owned host or device memory."""
    import ctypes as C
    import torch
    from xworld_amd.batched import BatchedSimulator
    sim = BatchedSimulator("simple_game", {"array_size": 16, "context": 2}, num_envs=512)
    for _ in range(5):
        sim.step()
    n, b = sim.num_envs, sim.obs_bytes_per_env
    host = np.empty(n * b, np.uint8)
    assert sim.L.xwb_get_obs(sim.h, host.ctypes.data, host.size, None) == 0
    assert np.array_equal(host, sim.obs.cpu().numpy().reshape(-1))
    dev = torch.empty(n * b, dtype=torch.uint8, device="cuda")
    assert sim.L.xwb_get_obs(sim.h, C.c_void_p(dev.data_ptr()), n * b, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(dev, sim.obs.reshape(-1))
    r = np.empty(n, np.float32)
    d = np.empty(n, np.uint8)
    assert sim.L.xwb_get_reward(sim.h, r.ctypes.data, None) == 0 and sim.L.xwb_get_done(sim.h, d.ctypes.data, None) == 0
    assert np.array_equal(r, sim.reward.cpu().numpy()) and np.array_equal(d, sim.game_over_codes.cpu().numpy())
    assert sim.L.xwb_get_obs(sim.h, host.ctypes.data, host.size - 1, None) != 0
    sim.close()


@pytest.mark.parametrize("extra", [pytest.param("", id="full"), pytest.param(", 'visible_radius': 3", id="ego", marks=pytest.mark.slow)])
def test_queue_sync_modes_agree_and_pmc_env_falls_back(extra):
    """The step loop's two queues hand over through device-side epochs by default and through events when a tool that
    serialises kernels is in sight (rocprofv3 --pmc sets ROCPROF_COUNTER_COLLECTION): both give the same rollout --
    step + reset_done and step_autoreset, full observation and the egocentric span path (frames hashed every step)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, os, hashlib; sys.path.insert(0, %r); import torch\n"
            "from xworld_amd.batched import BatchedSimulator\n"
            "conf = os.path.join(%r, 'xworld_amd', 'confs', 'navigation2d.json')\n"
            "sim = BatchedSimulator('xworld', {'xwd_conf_path': conf, 'task_mode': 'lang_acquisition', 'max_dim': 7, 'color': True" + extra + "}, num_envs=1024, seed=3, policy_seed=4)\n"
            "h = hashlib.sha256()\n"
            "for t in range(120):\n"
            "    if t %% 4 == 3:\n"
            "        sim.step_autoreset(); h.update(sim.obs.cpu().numpy().tobytes())\n"
            "    else:\n"
            "        sim.step(); h.update(sim.obs.cpu().numpy().tobytes())\n"
            "    h.update(sim.reward.cpu().numpy().tobytes()); h.update(sim.game_over_codes.cpu().numpy().tobytes())\n"
            "    if t %% 4 != 3: sim.reset_done()\n"
            "h.update(sim.obs.cpu().numpy().tobytes()); assert sim.check_errors() == 0; print(h.hexdigest())\n") % (root, root)
    outs = []
    for env in ({"XWB_QUEUE_SYNC": "epochs"}, {"XWB_QUEUE_SYNC": "events"}, {"ROCPROF_COUNTER_COLLECTION": "1"}):
        e = dict(os.environ)
        e.pop("XWB_QUEUE_SYNC", None)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] == outs[2], outs

"""Gather the state, not the pixels (include/xwb.h xwb_xw_pack_grids / xwb_xw_render_grids / xwb_gather_grids_begin): under
full observation a frame is a pure function of the env's cell codes, so the root of a sharded batch can draw every shard's
frames from 2 * D^2 + 1 bytes per env instead of receiving 144 * c * D^2 bytes of pixels.  What must hold: the tensor the
root draws equals, byte for byte, the frames the shards drew themselves (= an unsharded batch) -- after every verb, on every
step path (pre-generated / lazy / classic), for uint8 and float32 frames, with context rings.  Replaces the reference's
scale-out point simulator_interface.cpp:270-283 (one TCP round trip per env per step)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAV = os.path.join(ROOT, "xworld_amd", "confs", "navigation2d.json")
WALLS = os.path.join(ROOT, "xworld_amd", "confs", "walls_target.json")


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _bufs(torch, sim, n=None):
    n = sim.num_envs if n is None else n
    d = sim.cfg.max_dim
    return (torch.zeros((n, d * d), dtype=torch.int16, device="cuda"), torch.zeros((n,), dtype=torch.uint8, device="cuda"),
            torch.zeros((n,) + tuple(sim.obs.shape[1:]), dtype=sim.obs.dtype, device="cuda"))


CASES = {
    "nav7_color": {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 7, "color": True},
    "nav7_f32": {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 7, "color": True, "obs_format": "float32"},
    "walls7_ctx3": {"xwd_conf_path": WALLS, "task_mode": "one_channel", "context": 3, "max_steps": 30},
    "nav8_ctx2_color": {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "context": 2, "color": True},
    "nav11_gray": {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 11, "num_blocks": 30},
    "nav8_curriculum": {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "curriculum": 0.1, "color": True},
}


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("loop", ["step+reset_done", "step_autoreset", "skips"])
def test_pack_then_render_reproduces_the_frames(case, loop):
    """One batch: after EVERY frame-drawing verb the draw state is packed and drawn again into a second tensor, which must
    equal the batch's own observation buffer (context rings included: the second tensor only ever sees packed states)."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    n = 640
    sim = BatchedSimulator("xworld", CASES[case], num_envs=n, seed=21, policy_seed=4)
    grids, flags, mirror = _bufs(torch, sim)
    ctx1 = sim.cfg.context == 1

    def sync_mirror(tag):
        sim.pack_grids(grids, None if (ctx1 and tag % 2) else flags)       # flags are optional without a ring
        sim.render_grids(grids, flags, mirror)
        torch.cuda.synchronize()
        assert torch.equal(mirror, sim.obs), (case, loop, tag, int((mirror != sim.obs).sum()))

    sync_mirror(-1)                                                         # the first frames (xwb_create resets)
    rng = np.random.default_rng(3)
    resets = 0
    for t in range(70):
        if loop == "step+reset_done":
            sim.step()
            sync_mirror(2 * t)                                              # terminal frames of the envs that finished
            resets += int((sim.game_over_codes != 0).sum())
            sim.reset_done()
            sync_mirror(2 * t + 1)                                          # first frames of their next episodes
        elif loop == "step_autoreset":
            sim.step_autoreset()
            resets += int((sim.game_over_codes != 0).sum())
            sync_mirror(t)
        else:                                                               # some envs sit a step out; masked resets in between
            acts = torch.from_numpy(rng.integers(-1, sim.num_actions, n).astype(np.int32)).cuda()
            sim.step(acts)
            sync_mirror(3 * t)
            if t % 5 == 4:
                sim.reset_masked(torch.from_numpy((rng.random(n) < 0.1).astype(np.uint8)).cuda())
                sync_mirror(3 * t + 1)
            sim.reset_done()
            sync_mirror(3 * t + 2)
    assert loop == "skips" or resets > 0
    if not ctx1:
        # a ring is replayed one draw at a time: two draws without a pack in between cannot be packed
        sim.step()
        sim.reset_done()
        with pytest.raises(Exception, match="EVERY verb"):
            sim.pack_grids(grids, flags)
    sim.close()


def test_render_grids_draws_any_number_of_envs():
    """The root draws the WHOLE sharded batch with its own shard's kernel: n_envs is the caller's (here 3x the batch)."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    opts = CASES["nav7_color"]
    small = BatchedSimulator("xworld", opts, num_envs=100, seed=5, policy_seed=1, env_gid0=200)
    big = BatchedSimulator("xworld", opts, num_envs=300, seed=5, policy_seed=1)
    for _ in range(9):
        big.step()
        big.reset_done()
    g, f, out = _bufs(torch, big)
    big.pack_grids(g, f)
    small.render_grids(g, None, out)                                        # 300 frames through the 100-env batch
    torch.cuda.synchronize()
    assert torch.equal(out, big.obs)
    with pytest.raises(Exception):
        small.render_grids(g, None, out, n_envs=0)
    ego = BatchedSimulator("xworld", dict(opts, visible_radius=3), num_envs=8)
    with pytest.raises(Exception, match="egocentric"):
        ego.pack_grids(g, f)
    sg = BatchedSimulator("simple_game", {"array_size": 8}, num_envs=8)
    with pytest.raises(Exception, match="xworld"):
        sg.pack_grids(g, f)
    for s in (small, big, ego, sg):
        s.close()


SHARDED = [("c4", {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 7, "color": True}, [20000, 12768], 12),
           ("c5_shard", {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 11, "num_blocks": 30, "color": True}, [16384, 16384], 8),
           ("ragged_ctx2", {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "context": 2}, [384, 200, 0, 57], 30),
           ("single", {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 7, "color": True}, [3000], 4)]


@pytest.mark.parametrize("name,opts,counts,steps", SHARDED, ids=[s[0] for s in SHARDED])
def test_grids_gather_on_a_loopback_communicator(name, opts, counts, steps):
    """xwb_gather_grids_begin with real ncclSend / ncclRecv on one GPU: the shards of the batch live on one rank of a world-size-1
    communicator and post their halves as one group; the tensor the root draws equals the unsharded batch's frames AND what
    xwb_gather_screens_begin delivers (C4-sized batch; two C5 per-GPU shards; ragged shards incl. an empty one with a ring)."""
    torch = _torch()
    from xworld_amd import lib
    from xworld_amd import sharding
    from xworld_amd.batched import BatchedSimulator
    L = lib.load()
    comm = sharding.LibComm(0, 1, 0)
    total, ns = sum(counts), len(counts)
    whole = BatchedSimulator("xworld", opts, num_envs=total, seed=6, policy_seed=2)
    shards, g0 = [], 0
    for c in counts:
        shards.append(BatchedSimulator("xworld", opts, num_envs=c, seed=6, policy_seed=2, env_gid0=g0) if c else None)
        g0 += c
    live = [i for i in range(ns) if counts[i]]
    root = live[0]
    c_counts = (C.c_int32 * ns)(*counts)
    peers = (C.c_int32 * ns)(*([0] * ns))
    shape, dt = tuple(whole.obs.shape[1:]), whole.obs.dtype
    by_grids = [torch.zeros((total,) + shape, dtype=dt, device="cuda") for _ in range(2)]
    by_screens = torch.zeros((total,) + shape, dtype=dt, device="cuda")
    ring = whole.cfg.context > 1

    def gather(begin, dst):
        lib.check(L.xwb_comm_group_start(comm.h))
        try:
            for i in live:
                lib.check(begin(shards[i].h, comm.h, C.c_void_p(dst.data_ptr()) if i == root else None, c_counts, peers, ns, i, root, None))
        finally:                                                            # (an open group would poison RCCL for the whole process)
            lib.check(L.xwb_comm_group_end(comm.h))

    def check(t, k):
        gather(L.xwb_gather_grids_begin, by_grids[k])
        lib.check(L.xwb_comm_mark(comm.h, k))
        if not ring:                                                        # the pixels themselves, for comparison
            gather(L.xwb_gather_screens_begin, by_screens)
        lib.check(L.xwb_comm_wait(comm.h, k, None))
        lib.check(L.xwb_gather_screens_end(comm.h, None))
        torch.cuda.synchronize()
        assert torch.equal(by_grids[k], whole.obs), (name, t, int((by_grids[k] != whole.obs).sum()))
        assert ring or torch.equal(by_screens, whole.obs), (name, t)

    if ring:                                                                # a ring needs every draw: both buffers follow every step
        by_grids = [by_grids[0], by_grids[0]]
    check(-1, 0)
    for t in range(steps):
        for s in [whole] + [shards[i] for i in live]:
            s.step()
        if ring or t % 2:
            check(t, t & 1)                                                 # terminal frames
        for s in [whole] + [shards[i] for i in live]:
            s.reset_done()
        check(t, (t + 1) & 1)
    # outside a group a shard on the root's rank cannot reach it
    if len(live) > 1:
        assert L.xwb_gather_grids_begin(shards[live[1]].h, comm.h, None, c_counts, peers, ns, live[1], root, None) != 0
        assert b"group" in L.xwb_last_error()
    # a third gather of one batch inside ONE open group would reuse a slab whose transfer is only queued: refused, and the
    # refusal leaves the rotation alone -- the next well-formed gather is still exact
    if not ring and len(live) == 1:
        i = live[0]
        dstp = C.c_void_p(by_grids[0].data_ptr())
        lib.check(L.xwb_comm_group_start(comm.h))
        lib.check(L.xwb_gather_grids_begin(shards[i].h, comm.h, dstp, c_counts, peers, ns, i, root, None))
        lib.check(L.xwb_gather_grids_begin(shards[i].h, comm.h, dstp, c_counts, peers, ns, i, root, None))
        assert L.xwb_gather_grids_begin(shards[i].h, comm.h, dstp, c_counts, peers, ns, i, root, None) != 0
        assert b"more than two gathers" in L.xwb_last_error()
        assert L.xwb_comm_release_sim(comm.h, shards[i].h) != 0 and b"open group" in L.xwb_last_error()
        lib.check(L.xwb_comm_group_end(comm.h))
        lib.check(L.xwb_gather_screens_end(comm.h, None))
        check(steps, 1)
    # the slabs a communicator keeps for a batch can be handed back before the batch goes (they are re-made on demand)
    for i in live:
        lib.check(L.xwb_comm_release_sim(comm.h, shards[i].h))
    if not ring:                                                            # (a ring may only be packed once per draw)
        check(steps + 1, 0)
        for i in live:
            lib.check(L.xwb_comm_release_sim(comm.h, shards[i].h))
    for s in [whole] + [shards[i] for i in live]:
        s.close()
    comm.close()


def test_lib_gather_class_grids_mode_world_of_one():
    """sharding.LibScreensGather(mode="grids") -- what bench.py --exchange lib --gather grids runs on every rank -- and the
    torch.distributed-free GridsGather protocol pieces, one rank: pipelined (depth 2), on a non-default stream."""
    torch = _torch()
    from xworld_amd import sharding
    from xworld_amd.batched import BatchedSimulator
    opts = CASES["nav7_color"]
    sim = BatchedSimulator("xworld", opts, num_envs=512, seed=1)
    ref = BatchedSimulator("xworld", opts, num_envs=512, seed=1)
    comm = sharding.LibComm(0, 1, 0)
    st = torch.cuda.Stream()
    sg = sharding.LibScreensGather(sim, comm, [512], 0, mode="grids", stream=st)
    assert sg.depth == 2
    prev = None
    for t in range(12):
        sg.bind_next()
        sim.step(stream=st); ref.step()
        sim.reset_done(stream=st); ref.reset_done()
        sg.start()
        got = sg.latest()
        if prev is not None:
            st.synchronize()
            assert torch.equal(got, prev), t
        prev = ref.obs.clone()
    out = sg.drain()
    st.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(out, ref.obs)
    sim.close(); ref.close(); comm.close()


def test_gather_results_with_an_empty_shard():
    """counts[i] == 0 used to leave unmatched receives behind; an empty shard now sends nothing and still receives."""
    torch = _torch()
    from xworld_amd import lib, sharding
    L = lib.load()
    comm = sharding.LibComm(0, 1, 0)
    counts = [5, 0, 3]
    c_counts = (C.c_int32 * 3)(*counts)
    peers = (C.c_int32 * 3)(0, 0, 0)
    packed = [torch.arange(c * 2, dtype=torch.float32, device="cuda").view(c, 2) + 100 * i for i, c in enumerate(counts)]
    packed[1] = torch.zeros((1, 2), device="cuda")                          # an empty shard still passes a valid pointer
    allres = torch.full((8, 2), -1.0, device="cuda")
    lib.check(L.xwb_comm_group_start(comm.h))
    for i in range(3):
        lib.check(L.xwb_gather_results(comm.h, C.c_void_p(packed[i].data_ptr()), C.c_void_p(allres.data_ptr()), c_counts, peers, 3, i, None))
    lib.check(L.xwb_comm_group_end(comm.h))
    torch.cuda.synchronize()
    assert torch.equal(allres[:5], packed[0]) and torch.equal(allres[5:], packed[2])
    assert L.xwb_comm_group_end(comm.h) != 0                                # unbalanced end
    comm.close()


@pytest.mark.parametrize("case", ["nav7_color", "walls7_ctx3", "nav7_f32", "nav8_curriculum"])
@pytest.mark.parametrize("loop", ["step+reset_done", "step_autoreset"])
def test_a_batch_that_does_not_draw_feeds_a_renderer(case, loop):
    """xwb_xw_set_draw(sim, 0): the verbs store no pixels, the draw state still leaves through xwb_xw_pack_grids, and a
    renderer fed with it alone reproduces, frame for frame, the observation buffer of an identical batch that does draw --
    rewards, codes and state included (the renders' bookkeeping -- epochs, installs, flags -- still runs)."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    n = 768
    a = BatchedSimulator("xworld", CASES[case], num_envs=n, seed=13, policy_seed=2)
    b = BatchedSimulator("xworld", CASES[case], num_envs=n, seed=13, policy_seed=2)
    grids, flags, mirror = _bufs(torch, b)
    b.pack_grids(grids, flags)                                            # the first frames, drawn by xwb_create
    b.render_grids(grids, flags, mirror)
    b.set_draw(False)
    stale = b.obs.clone()

    def check(tag):
        b.pack_grids(grids, flags)
        b.render_grids(grids, flags, mirror)
        torch.cuda.synchronize()
        assert torch.equal(mirror, a.obs), (case, loop, tag, int((mirror != a.obs).sum()))
        assert torch.equal(b.reward, a.reward) and torch.equal(b.game_over_codes, a.game_over_codes), tag
        assert torch.equal(b.grid, a.grid) and torch.equal(b.num_steps, a.num_steps), tag

    for t in range(60):
        for s in (a, b):
            if loop == "step_autoreset":
                s.step_autoreset()
            else:
                s.step()
        check(2 * t)
        if loop != "step_autoreset":
            a.reset_done(); b.reset_done()
            check(2 * t + 1)
    assert torch.equal(b.obs, stale)                                       # nothing was drawn into the batch's own buffer
    assert a.task_performance() == b.task_performance() and b.check_errors() == 0
    b.set_draw(True)
    for s in (a, b):
        s.reset()
    assert torch.equal(a.obs, b.obs)                                       # a verb that draws every env makes the buffer current again
    ego = BatchedSimulator("xworld", dict(CASES["nav7_color"], visible_radius=3), num_envs=8)
    with pytest.raises(Exception, match="egocentric"):
        ego.set_draw(False)
    for s in (a, b, ego):
        s.close()

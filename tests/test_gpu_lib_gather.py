"""The multi-GPU exchange issued by libxwb.so itself (include/xwb.h: xwb_comm_*, xwb_gather_*; RCCL below Python) on ONE
GPU: RCCL refuses two ranks on one device, so the two shards of the batch live on one rank of a world-size-1 communicator
(a loopback: `peers` = [0, 0]) and post their halves of the exchange -- ncclRecv into the root's slice, ncclSend of the
other shard's frames -- as one group.  Gathered screens and results must equal an unsharded batch, step for step."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONF = os.path.join(ROOT, "xworld_amd", "confs", "navigation2d.json")


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


GAMES = [("xworld", {"xwd_conf_path": CONF, "task_mode": "lang_acquisition", "max_dim": 7, "color": True}),
         ("xworld", {"xwd_conf_path": CONF, "task_mode": "lang_acquisition", "max_dim": 7, "visible_radius": 3}),
         ("simple_game", {"array_size": 16}),
         ("simple_race", {"track_width": 20.0, "track_length": 100.0, "track_radius": 30.0, "random": True})]


@pytest.mark.parametrize("game,opts", GAMES, ids=["xworld", "ego", "simple_game", "simple_race"])
def test_two_shards_on_a_loopback_communicator(game, opts):
    torch = _torch()
    from xworld_amd import lib, sharding
    from xworld_amd.batched import BatchedSimulator
    L = lib.load()
    comm = sharding.LibComm(0, 1, 0)
    assert comm.version >= 20000
    counts = [384, 200]                                       # ragged on purpose
    total = sum(counts)
    whole = BatchedSimulator(game, opts, num_envs=total, seed=6, policy_seed=2)
    shards = [BatchedSimulator(game, opts, num_envs=counts[0], seed=6, policy_seed=2, env_gid0=0),
              BatchedSimulator(game, opts, num_envs=counts[1], seed=6, policy_seed=2, env_gid0=counts[0])]
    c_counts = (C.c_int32 * 2)(*counts)
    peers = (C.c_int32 * 2)(0, 0)
    shape = tuple(whole.obs.shape[1:])
    # double buffered: the root renders straight into its slice of the destination, the other shard into its own buffers
    full = [torch.zeros((total,) + shape, dtype=whole.obs.dtype, device="cuda") for _ in range(2)]
    other = [torch.zeros((counts[1],) + shape, dtype=whole.obs.dtype, device="cuda") for _ in range(2)]
    packed = [torch.zeros((c, 2), device="cuda") for c in counts]
    allres = torch.full((total, 2), -9.0, device="cuda")
    for s, p in zip(shards, packed):
        s.bind_results(p)
    prev_pending = None
    for t in range(40):
        k = t & 1
        shards[0].bind_obs(full[k][:counts[0]])
        shards[1].bind_obs(other[k])
        whole.step()
        for s in shards:
            s.step()
        # results: each shard hands its rows over (shards on one rank only copy: nothing crosses RCCL)
        for i, s in enumerate(shards):
            lib.check(L.xwb_gather_results(comm.h, C.c_void_p(packed[i].data_ptr()), C.c_void_p(allres.data_ptr()), c_counts, peers, 2, i, None))
        torch.cuda.synchronize()
        assert torch.equal(allres[:, 0], whole.reward) and torch.equal(allres[:, 1], whole.game_over_codes.float()), t
        for s in shards + [whole]:
            s.reset_done()
        # screens (the frames the next policy step sees): root posts the receive, the other shard the send, one group; the
        # transfer runs on the communicator's stream, beside the next iteration's step (which renders into the other buffers)
        lib.check(L.xwb_comm_group_start(comm.h))
        lib.check(L.xwb_gather_screens_begin(shards[0].h, comm.h, C.c_void_p(full[k].data_ptr()), c_counts, peers, 2, 0, 0, None))
        lib.check(L.xwb_gather_screens_begin(shards[1].h, comm.h, None, c_counts, peers, 2, 1, 0, None))
        lib.check(L.xwb_comm_group_end(comm.h))
        want = whole.obs.clone()
        if t % 3 == 2:                                        # sometimes waited for at once, usually one step later
            lib.check(L.xwb_gather_screens_end(comm.h, None))
            torch.cuda.synchronize()
            assert torch.equal(full[k], want), t
            pending = None
        else:
            pending = (k, want)
        if prev_pending is not None:                          # the transfer begun one iteration ago: its buffers come up next
            lib.check(L.xwb_gather_screens_end(comm.h, None))
            torch.cuda.synchronize()
            assert torch.equal(full[prev_pending[0]], prev_pending[1]), t
        prev_pending = pending
    # one shard per rank, equal shards: the all-gather path of a world of one
    one = (C.c_int32 * 1)(counts[0])
    out = torch.zeros((counts[0], 2), device="cuda")
    lib.check(L.xwb_gather_results(comm.h, C.c_void_p(packed[0].data_ptr()), C.c_void_p(out.data_ptr()), one, None, 1, 0, None))
    torch.cuda.synchronize()
    assert torch.equal(out, packed[0])
    # argument checks
    assert L.xwb_gather_screens_begin(shards[0].h, comm.h, None, c_counts, peers, 2, 0, 0, None) != 0          # root without destination
    bad = (C.c_int32 * 2)(0, 3)
    assert L.xwb_gather_screens_begin(shards[1].h, comm.h, None, c_counts, bad, 2, 1, 0, None) != 0            # peer outside the world
    for s in shards + [whole]:
        s.close()
    comm.close()


def test_lib_screens_gather_class_world_of_one():
    """sharding.LibScreensGather / lib_gather_results (what bench.py --exchange lib runs on every rank) with one rank."""
    torch = _torch()
    from xworld_amd import sharding
    from xworld_amd.batched import BatchedSimulator
    sim = BatchedSimulator("xworld", {"xwd_conf_path": CONF, "task_mode": "lang_acquisition", "max_dim": 7, "color": True}, num_envs=512, seed=1)
    ref = BatchedSimulator("xworld", {"xwd_conf_path": CONF, "task_mode": "lang_acquisition", "max_dim": 7, "color": True}, num_envs=512, seed=1)
    comm = sharding.LibComm(0, 1, 0)
    sg = sharding.LibScreensGather(sim, comm, [512], 0)
    packed = torch.zeros((512, 2), device="cuda")
    out = torch.zeros((512, 2), device="cuda")
    sim.bind_results(packed)
    prev = None
    for t in range(12):
        sg.bind_next()
        sim.step(); ref.step()
        sharding.lib_gather_results(comm, packed, out, [512], 0)
        sim.reset_done(); ref.reset_done()
        sg.start()
        got = sg.latest()
        if prev is not None:
            assert torch.equal(got, prev), t
        prev = ref.obs.clone()
        torch.cuda.synchronize()
        assert torch.equal(out[:, 0], ref.reward)
    assert torch.equal(sg.drain(), ref.obs)
    sim.close(); ref.close(); comm.close()


@pytest.mark.parametrize("game,opts,want_epoch", [("xworld", {"xwd_conf_path": os.path.join(ROOT, "xworld_amd", "confs", "navigation2d.json"),
                                                              "task_mode": "lang_acquisition", "max_dim": 7, "color": True}, True),
                                                  ("xworld", {"xwd_conf_path": os.path.join(ROOT, "xworld_amd", "confs", "navigation2d.json"),
                                                              "task_mode": "lang_acquisition", "max_dim": 7, "visible_radius": 3}, False),
                                                  ("simple_game", {"array_size": 16}, False)], ids=["xworld", "ego", "simple_game"])
def test_results_gathered_beside_the_step_loop(game, opts, want_epoch):
    """xwb_gather_results_beside (LibResultGather): the last step's rows of the results ring, all-gathered on the communicator's
    stream behind that step's kernel -- by the step's epoch on a full-observation xworld batch (nothing enqueued on the caller's
    stream), by an event otherwise -- equal the batch's own reward / codes every step, whether the caller waits (finish) or lets
    the exchange go (release) and reads after a drain."""
    torch = _torch()
    from xworld_amd import sharding
    from xworld_amd.batched import BatchedSimulator
    n = 4096
    sim = BatchedSimulator(game, opts, num_envs=n, seed=12, policy_seed=5)
    comm = sharding.LibComm(0, 1, 0)
    # before a ring is bound / before the first step: refused
    rg = sharding.LibResultGather(sim, comm, [n], 0)
    with pytest.raises(Exception, match="results ring"):
        rg.start()
    ring = torch.zeros((4, n, 2), dtype=torch.float32, device="cuda")
    sim.bind_results_ring(ring)
    rg = sharding.LibResultGather(sim, comm, [n], 0)
    with pytest.raises(Exception, match="no step"):
        rg.start()
    mode = sim.queue_sync_mode()[0] if game == "xworld" else "events"
    kept = []
    for t in range(24):
        sim.step()
        rew, codes = sim.reward.clone(), sim.game_over_codes.float().clone()
        fused = sim.step_path()["path"] == "lazy_fused"
        late = t % 4 >= 2                                           # the exchange behind reset_done (bench.py's order) or in front of it
        if late:
            sim.reset_done()
        rg.start()
        if t % 3 == 0:
            r, c = rg.finish(convert=False)
            torch.cuda.synchronize()
            assert torch.equal(r, rew) and torch.equal(c, codes), t
        else:
            rg.release()
            kept.append((t, rg.out[rg.slot ^ 1], rew, codes))
        # (a fused step + render launch has no kernel behind its step blocks that could publish the step's epoch before
        # xwb_reset_done's list render is queued: an exchange started in between is handed that call's rows by an event)
        assert rg.by_epoch == (want_epoch and mode == "epochs" and (late or not fused)), (t, rg.by_epoch, mode, fused, late)
        if not late:
            sim.reset_done()
        if len(kept) == 2:                                          # the two buffers alternate: read them before they come round
            rg.drain()
            torch.cuda.synchronize()
            for (tt, out, rr, cc) in kept:
                assert torch.equal(out[:, 0], rr) and torch.equal(out[:, 1], cc), tt
            kept = []
    rg.drain()
    assert sim.check_errors() == 0
    sim.close()
    comm.close()

"""The N > 1 path on CPU: world_size 2 (and 3, ragged) over gloo -- shard ranges, per-step result gather,
screen-slab gather into one contiguous tensor on rank 0."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from xworld_amd import sharding


def test_shard_ranges_cover_the_batch():
    for total in (1, 7, 32768, 262144, 100003):
        for world in (1, 2, 3, 4, 8):
            spans = [sharding.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        counts = sharding.shard_counts(total, world)
        start, n = sharding.shard_range(total, world, rank)
        # what a rank's simulator would hold: values are functions of the GLOBAL env id
        gid = torch.arange(start, start + n)
        reward = gid.to(torch.float32) * 0.5 - 3.0
        done = (gid % 5 == 0).to(torch.uint8) * 4
        obs = (gid[:, None] * 7 + torch.arange(12)[None, :]).to(torch.uint8).reshape(n, 1, 3, 4)
        rg = sharding.ResultGather(counts, rank, torch.device("cpu"))
        for step in range(3):
            r_all, d_all = rg(reward + step, done)
            out = torch.zeros((total, 1, 3, 4), dtype=torch.uint8) if rank == 0 else None
            got = sharding.gather_slabs(obs, out, counts, rank)
            if rank == 0:
                g = torch.arange(total)
                assert torch.equal(r_all, g.to(torch.float32) * 0.5 - 3.0 + step)
                assert torch.equal(d_all, (g % 5 == 0).to(torch.uint8) * 4)
                exp = (g[:, None] * 7 + torch.arange(12)[None, :]).to(torch.uint8).reshape(total, 1, 3, 4)
                assert torch.equal(got, exp) and got.is_contiguous()
            else:
                assert r_all is None and got is None
        # pipelined form with the simulator writing straight into the exchange buffer (bind_results): the gather of
        # step t is finished only after step t + 1 was produced
        prev = None
        for step in range(4):
            buf = rg.next_buffer()                      # what BatchedSimulator.bind_results would be given
            buf[:, 0] = reward + 10 * step
            buf[:, 1] = done.to(torch.float32)
            got_prev = rg.finish()
            rg.start()
            if rank == 0 and prev is not None:
                g = torch.arange(total)
                assert torch.equal(got_prev[0], g.to(torch.float32) * 0.5 - 3.0 + 10 * prev)
                assert torch.equal(got_prev[1], (g % 5 == 0).to(torch.uint8) * 4)
            prev = step
        rg.finish()
        # the pipelined form shipping caller-owned rows of a per-step record (bench.py: xwb_bind_results_ring)
        ring = torch.zeros((3, n, 2))
        for step in range(5):
            ring[step % 3, :, 0] = reward + 100 * step
            ring[step % 3, :, 1] = done.to(torch.float32)
            got_prev = rg.finish()
            rg.start(packed=ring[step % 3])
            if rank == 0 and step > 0:
                g = torch.arange(total)
                assert torch.equal(got_prev[0], g.to(torch.float32) * 0.5 - 3.0 + 100 * (step - 1))
        rg.finish()
        # double-buffered screens gather: the simulator renders step t into the free buffer pair, the transfer of step t
        # is waited for when its buffers come up again (context 1) or at once (context ring: one buffer)
        for context in (1, 2):
            class FakeSim:
                class cfg:
                    pass
                def __init__(self):
                    self.cfg.context = context
                    self.obs = torch.zeros((n, context, 3, 4), dtype=torch.uint8)
                def bind_obs(self, t):
                    self.obs = t
            fs = FakeSim()
            sg = sharding.ScreensGather(fs, counts, rank)
            assert sg.depth == (2 if context == 1 else 1)
            g = torch.arange(total)
            for step in range(5):
                sg.bind_next()
                fs.obs.copy_(((gid[:, None] * 3 + step + torch.arange(12 * context)[None, :]) % 251).to(torch.uint8).reshape(n, context, 3, 4))
                sg.start()
                got = sg.latest()
                if rank == 0:
                    want_step = step - 1 if context == 1 else step
                    if want_step >= 0:
                        exp = ((g[:, None] * 3 + want_step + torch.arange(12 * context)[None, :]) % 251).to(torch.uint8).reshape(total, context, 3, 4)
                        assert got is not None and got.is_contiguous() and torch.equal(got, exp), (context, step)
                    else:
                        assert got is None
                else:
                    assert got is None
            final = sg.drain()
            if rank == 0:
                exp = ((g[:, None] * 3 + 4 + torch.arange(12 * context)[None, :]) % 251).to(torch.uint8).reshape(total, context, 3, 4)
                assert torch.equal(final, exp)
        # several shards per rank (gather_shards: shard i lives on rank peers[i]), ragged, one of them empty -- the layout the
        # one-GPU tests of the nccl branch use (tests/test_gpu_nccl_branch.py), here over gloo with two shards per rank
        ns = 2 * world
        sc = [5, 0, 3, 7, 2, 4][:ns]
        peers = [i // 2 for i in range(ns)]
        offs = [sum(sc[:i]) for i in range(ns)]
        mine = {i: (torch.arange(offs[i], offs[i] + sc[i])[:, None] * 3 + torch.arange(6)[None, :]).to(torch.int32) for i in range(ns) if peers[i] == rank}
        for root_shard in (0, ns - 1):
            holder = peers[root_shard]
            out = torch.full((sum(sc), 6), -1, dtype=torch.int32) if rank == holder else None
            got = sharding.gather_shards(mine, out, sc, peers, rank, dst=root_shard)
            if rank == holder:
                exp = (torch.arange(sum(sc))[:, None] * 3 + torch.arange(6)[None, :]).to(torch.int32)
                assert torch.equal(got, exp), (root_shard, got)
            else:
                assert got is None
        rgf = sharding.ResultGather([4], 0, torch.device("cpu"), force_collective=True) if world == 1 else None
        assert rgf is None
        assert sharding.backend_info() == {"world_size": world, "backend": "gloo", "version": None}
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:                               # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 64), (2, 33), (3, 100)])
def test_gather_over_gloo(world, total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res

"""Sharding the env batch over GPUs: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Envs are independent (the reference's own scale-out is one OS process per env,
examples/demo_interface.cpp:67-95), so the batch is split into contiguous ranges of *global* env ids;
every RNG stream is keyed by the global id, hence results do not depend on the number of shards.
The only exchange is the per-step gather of results to rank 0:
  * gather_results: (reward f32, game_over u8) of every shard, a few bytes per env;
  * gather_screens: every shard's observation slab into one contiguous tensor on rank 0.  Each remote
    shard crosses its one direct xGMI link to the root, so this is link-bound (DESIGN.md "multi-GPU").
These helpers only move tensors; they work with any backend (tests use gloo on CPU).
"""
import torch
import torch.distributed as dist


def shard_range(total_envs, world_size, rank):
    """Contiguous split of [0, total_envs): (first global env id, number of envs) of `rank`."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(total_envs, world_size)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def shard_counts(total_envs, world_size):
    return [shard_range(total_envs, world_size, r)[1] for r in range(world_size)]


class ResultGather:
    """Per-step gather of (reward, game_over) to `dst`.  Equal shard sizes -> one all_gather_into_tensor of a packed
    [n, 2] float tensor; ragged shards -> point-to-point into slices.

    `start()` enqueues the exchange of the step that just finished and returns at once; `finish()` waits for
    it (stream-side) and hands out the gathered tensors.  Calling finish() for step t only after step t+1 has
    been enqueued lets the (latency-bound, few-hundred-KB) collective run beside the next step's kernels --
    two packed/out buffers alternate so the in-flight exchange is never overwritten.  `__call__` = start + finish."""

    def __init__(self, counts, rank, device, dst=0, group=None):
        self.counts, self.rank, self.dst, self.group = list(counts), rank, dst, group
        self.world = len(counts)
        self.equal = len(set(counts)) == 1
        n = counts[rank]
        self.total = sum(counts)
        self.packed = [torch.empty((n, 2), dtype=torch.float32, device=device) for _ in range(2)]
        # equal shards use all_gather_into_tensor (the plainest RCCL collective; 8 B/env, so the extra copies on the
        # other ranks are noise): every rank owns an output buffer; ragged shards gather point-to-point into `dst` only
        self.out = [torch.empty((self.total, 2), dtype=torch.float32, device=device) if (rank == dst or self.equal) else None
                    for _ in range(2)]
        self.slot = 0
        self.pending = None          # (work handle or None, slot)

    def next_buffer(self):
        """The packed [n, 2] buffer the NEXT start() ships: bind it as the simulator's results output
        (BatchedSimulator.bind_results) and call start() without arguments -- no packing kernels at all."""
        return self.packed[self.slot]

    def start(self, reward=None, game_over=None):
        if self.pending is not None:
            raise RuntimeError("ResultGather.start() called twice without finish()")
        k = self.slot
        self.slot ^= 1
        packed, out = self.packed[k], self.out[k]
        if reward is not None:
            packed[:, 0] = reward
            packed[:, 1] = game_over.to(torch.float32)
        work = None
        if self.world == 1:
            out.copy_(packed)
        elif self.equal:
            work = dist.all_gather_into_tensor(out, packed, group=self.group, async_op=True)
        else:
            gather_slabs(packed, out, self.counts, self.rank, self.dst, self.group)
        self.pending = (work, k)

    def finish(self):
        if self.pending is None:
            return None, None
        work, k = self.pending
        self.pending = None
        if work is not None:
            work.wait()
        if self.rank != self.dst:
            return None, None
        out = self.out[k]
        return out[:, 0], out[:, 1].to(torch.uint8)

    def __call__(self, reward, game_over):
        self.start(reward, game_over)
        return self.finish()


def gather_slabs(local, out_root, counts, rank, dst=0, group=None):
    """Every rank's `local` [count_r, ...] slab into out_root[offset_r : offset_r + count_r] on `dst`
    (the root's own slab is copied unless it already aliases its slice)."""
    world = len(counts)
    offsets = [sum(counts[:r]) for r in range(world)]
    if rank == dst:
        mine = out_root[offsets[dst]:offsets[dst] + counts[dst]]
        if mine.data_ptr() != local.data_ptr():
            mine.copy_(local)
        ops = [dist.P2POp(dist.irecv, out_root[offsets[r]:offsets[r] + counts[r]], r, group=group)
               for r in range(world) if r != dst and counts[r] > 0]
    else:
        ops = [dist.P2POp(dist.isend, local, dst, group=group)] if counts[rank] > 0 else []
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return out_root if rank == dst else None

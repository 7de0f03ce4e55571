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
    """Per-step gather of (reward, game_over) to `dst`.  Equal shard sizes -> one dist.gather of a packed
    [n, 2] float tensor; ragged shards -> point-to-point into slices."""

    def __init__(self, counts, rank, device, dst=0, group=None):
        self.counts, self.rank, self.dst, self.group = list(counts), rank, dst, group
        self.world = len(counts)
        self.equal = len(set(counts)) == 1
        n = counts[rank]
        self.packed = torch.empty((n, 2), dtype=torch.float32, device=device)
        self.total = sum(counts)
        self.offsets = [sum(counts[:r]) for r in range(self.world)]
        self.out = torch.empty((self.total, 2), dtype=torch.float32, device=device) if rank == dst else None

    def __call__(self, reward, game_over):
        self.packed[:, 0] = reward
        self.packed[:, 1] = game_over.to(torch.float32)
        if self.world == 1:
            self.out.copy_(self.packed)
        elif self.equal:
            n = self.counts[0]
            lst = [self.out[r * n:(r + 1) * n] for r in range(self.world)] if self.rank == self.dst else None
            dist.gather(self.packed, lst, dst=self.dst, group=self.group)
        else:
            gather_slabs(self.packed, self.out, self.counts, self.rank, self.dst, self.group)
        if self.rank != self.dst:
            return None, None
        return self.out[:, 0], self.out[:, 1].to(torch.uint8)


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

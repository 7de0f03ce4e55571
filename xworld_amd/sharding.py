"""Sharding the env batch over GPUs: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Envs are independent (the reference's own scale-out is one OS process per env,
examples/demo_interface.cpp:67-95), so the batch is split into contiguous ranges of *global* env ids;
every RNG stream is keyed by the global id, hence results do not depend on the number of shards.
The only exchange is the per-step gather of results to rank 0:
  * ResultGather: (reward f32, game_over u8) of every shard, a few bytes per env;
  * ScreensGather: every shard's observation slab into one contiguous tensor on rank 0.  Each remote
    shard crosses its one direct xGMI link to the root, so this is link-bound (DESIGN.md "multi-GPU");
    it is double-buffered: the transfer of step t runs beside the kernels of step t + 1.
These helpers only move tensors; they work with any backend.  The CPU tests use gloo; the single-GPU test of
the N > 1 path runs several ranks on one device over gloo, whose device-tensor collectives are staged through
host memory here (`_Staged`) -- RCCL refuses two ranks on one device.
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


def backend_info(group=None):
    """What the exchange runs on: {"world_size", "backend", "version"} (version = RCCL's for backend nccl)."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"world_size": 1, "backend": "none", "version": None}
    be = dist.get_backend(group)
    ver = None
    if be == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                                   # pragma: no cover
            ver = None
    return {"world_size": dist.get_world_size(group), "backend": be, "version": ver}


def _needs_staging(t, group=None):
    # gloo has no device-tensor send / recv / all_gather: stage through host memory (test transport only)
    return t.is_cuda and dist.get_backend(group) == "gloo"


class _Staged:
    """A finished-on-wait handle for an exchange that went through host staging buffers."""

    def __init__(self, copies):
        self.copies = copies          # [(device destination, host source)]

    def wait(self):
        for dst, src in self.copies:
            dst.copy_(src)


def _p2p_gather(locals_by_shard, out_root, counts, peers, rank, dst, group):
    """isend / irecv of every shard's slab into its slice of out_root on the holder of shard `dst`; returns the work handles.
    Shard i lives on rank peers[i]; `locals_by_shard` = {shard: slab} of the shards THIS rank holds (one shard per rank in
    production: {rank: local}, peers = 0 .. world - 1; several shards on one rank -- a loopback on one GPU -- take the very
    same lines: the sends to the own rank and the matching receives leave in one batch)."""
    n_shards = len(counts)
    offsets = [sum(counts[:i]) for i in range(n_shards)]
    root_rank = peers[dst]
    some = next(iter(locals_by_shard.values()))
    if _needs_staging(some, group):
        handles = []
        if rank == root_rank:
            copies = []
            for i in range(n_shards):
                if i == dst or counts[i] == 0 or peers[i] == rank:
                    continue
                host = torch.empty((counts[i],) + tuple(out_root.shape[1:]), dtype=out_root.dtype)
                handles.append(dist.irecv(host, peers[i], group=group))
                copies.append((out_root[offsets[i]:offsets[i] + counts[i]], host))
            for i, t in locals_by_shard.items():            # (gloo: shards on the root's own rank are plain copies)
                if i != dst and counts[i] > 0:
                    copies.append((out_root[offsets[i]:offsets[i] + counts[i]], t))
            handles.append(_Staged(copies))     # waited last: the receives above are complete by then
            return _Ordered(handles)
        for i, t in locals_by_shard.items():
            if counts[i] > 0:
                handles.append(dist.isend(t.cpu(), root_rank, group=group))
        return _Ordered(handles)
    ops, local_copies = [], []
    # shards on the root's own rank: RCCL matches a send to the own rank with a receive of the same batch (what the one-GPU
    # test of this branch relies on); other backends copy
    self_p2p = dist.get_backend(group) == "nccl"
    if rank == root_rank:
        ops += [dist.P2POp(dist.irecv, out_root[offsets[i]:offsets[i] + counts[i]], peers[i], group=group)
                for i in range(n_shards) if i != dst and counts[i] > 0 and (self_p2p or peers[i] != rank)]
    for i, t in locals_by_shard.items():
        if i == dst or counts[i] == 0:
            continue
        if rank == root_rank and not self_p2p:
            local_copies.append((out_root[offsets[i]:offsets[i] + counts[i]], t))
        else:
            ops.append(dist.P2POp(dist.isend, t, root_rank, group=group))
    for d, t in local_copies:
        d.copy_(t)
    global _P2P_BROKEN
    if not _P2P_BROKEN:
        try:
            return _Ordered(dist.batch_isend_irecv(ops) if ops else [])
        except Exception as e:                              # pragma: no cover  (never seen; first contact with a new node)
            # every rank runs the same code on the same backend: a refusal is expected to hit all of them alike
            import warnings
            warnings.warn("batched point-to-point failed (%s): screens travel by all_gather from now on" % e)
            _P2P_BROKEN = True
    if len(locals_by_shard) != 1 or list(peers) != list(range(n_shards)):
        raise RuntimeError("the all_gather fallback needs one shard per rank")
    return _allgather_fallback(some, out_root, counts, rank, dst, group)


_P2P_BROKEN = False


def _allgather_fallback(local, out_root, counts, rank, dst, group):
    """Plan B for a backend that refuses batched send / recv: all_gather of equal (padded) slabs; only `dst` keeps them."""
    world = len(counts)
    m = max(counts)
    pad = local if counts[rank] == m else torch.cat([local, local.new_zeros((m - counts[rank],) + tuple(local.shape[1:]))])
    tmp = local.new_empty((world * m,) + tuple(local.shape[1:]))
    if _needs_staging(local, group):
        host = torch.empty(tuple(tmp.shape), dtype=tmp.dtype)
        dist.all_gather_into_tensor(host, pad.contiguous().cpu(), group=group)
        tmp.copy_(host)
    else:
        dist.all_gather_into_tensor(tmp, pad.contiguous(), group=group)
    if rank == dst:
        off = 0
        for r in range(world):
            if r != dst and counts[r] > 0:
                out_root[off:off + counts[r]].copy_(tmp[r * m:r * m + counts[r]])
            off += counts[r]
    return _Ordered([])


class _Ordered:
    def __init__(self, handles):
        self.handles = list(handles)

    def wait(self):
        for h in self.handles:
            h.wait()


def gather_shards(locals_by_shard, out_root, counts, peers, rank, dst=0, group=None, async_op=False):
    """Every shard's slab into out_root[offset_i : offset_i + count_i] on the rank that holds shard `dst` (that shard's own
    slab is copied unless it already aliases its slice).  `locals_by_shard` = {shard: [count_i, ...] slab} of the shards this
    rank holds, shard i living on rank peers[i].  async_op: returns a handle whose wait() orders the current stream behind
    the transfer instead of waiting here."""
    if dst in locals_by_shard:
        off = sum(counts[:dst])
        mine = out_root[off:off + counts[dst]]
        if mine.data_ptr() != locals_by_shard[dst].data_ptr():
            mine.copy_(locals_by_shard[dst])
    work = _p2p_gather(locals_by_shard, out_root, counts, list(peers), rank, dst, group) if len(counts) > 1 else _Ordered([])
    if async_op:
        return work
    work.wait()
    return out_root if dst in locals_by_shard else None


def gather_slabs(local, out_root, counts, rank, dst=0, group=None, async_op=False):
    """gather_shards with one shard per rank (shard r on rank r): what every N > 1 run uses."""
    return gather_shards({rank: local}, out_root, counts, range(len(counts)), rank, dst, group, async_op)


def nccl_init_kwargs(device):
    """Keyword arguments for dist.init_process_group("nccl", ...) on a rank that also steps a batch: bind the device and put
    the communicator's internal stream at high priority.  Why: HIP maps streams of one priority onto a small pool of hardware
    queues (4), in creation order; a communication stream that lands on the hardware queue of the batch's internal stream
    queues the per-step exchange IN FRONT of the next map generator and behind the event it waits for -- the generator then
    follows the render instead of running beside it (kernel trace, one MI355X, C4, world of one: 0.209 ms per step against
    0.113 without an exchange; with a high-priority communication stream, which comes from another pool: 0.128)."""
    kw = {"device_id": device}
    try:
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
        kw["pg_options"] = opts
    except Exception:                                    # a build without the option: the defaults still work, only slower
        pass
    return kw


def init_nccl(device, **kw):
    """dist.init_process_group("nccl", ...) with nccl_init_kwargs; a torch build whose init_process_group does not take the
    options falls back to the plain call (same results, the slower queue layout nccl_init_kwargs describes)."""
    extra = nccl_init_kwargs(device)
    try:
        return dist.init_process_group("nccl", **extra, **kw)
    except TypeError:
        return dist.init_process_group("nccl", device_id=device, **kw)


class ResultGather:
    """Per-step gather of (reward, game_over) to `dst`.  Equal shard sizes -> one all_gather_into_tensor of a packed
    [n, 2] float tensor; ragged shards -> point-to-point into slices.

    `start()` enqueues the exchange of the step that just finished and returns at once; `finish()` waits for
    it (stream-side) and hands out the gathered tensors.  Calling finish() for step t only after step t+1 has
    been enqueued lets the (latency-bound, few-hundred-KB) collective run beside the next step's kernels --
    two packed/out buffers alternate so the in-flight exchange is never overwritten.  `start(packed=t)` ships a
    caller-owned [n, 2] tensor instead (e.g. one row of a per-step record the simulator wrote through
    bind_results); the caller then keeps it untouched until finish().  `__call__` = start + finish."""

    def __init__(self, counts, rank, device, dst=0, group=None, force_collective=False):
        self.counts, self.rank, self.dst, self.group = list(counts), rank, dst, group
        self.world = len(counts)
        # a world of one needs no exchange; force_collective issues the collective all the same (a single-GPU run of the
        # very RCCL call the N > 1 run makes: tests/test_gpu_nccl_branch.py, bench.py --force-exchange)
        self.force = bool(force_collective)
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

    def start(self, reward=None, game_over=None, packed=None):
        if self.pending is not None:
            raise RuntimeError("ResultGather.start() called twice without finish()")
        k = self.slot
        self.slot ^= 1
        out = self.out[k]
        if packed is None:
            packed = self.packed[k]
            if reward is not None:
                packed[:, 0] = reward
                packed[:, 1] = game_over.to(torch.float32)
        work = None
        if self.world == 1 and not self.force:
            out.copy_(packed)
        elif self.equal and not _needs_staging(packed, self.group):
            work = dist.all_gather_into_tensor(out, packed, group=self.group, async_op=True)
        elif self.equal:
            host_out = torch.empty(tuple(out.shape), dtype=out.dtype)
            h = dist.all_gather_into_tensor(host_out, packed.cpu(), group=self.group, async_op=True)
            work = _Ordered([h, _Staged([(out, host_out)])])
        else:
            work = gather_slabs(packed, out, self.counts, self.rank, self.dst, self.group, async_op=True)
        self.pending = (work, k)

    def release(self):
        """Let go of the exchange started last WITHOUT ordering the caller's stream behind it: nobody on this stream reads its
        result (a rollout whose policy acts on device-resident shards; the root's trainer would call finish()).  The collective
        still runs, on the backend's own stream, in order with the ones before and after it -- so the two output buffers are
        reused safely -- and `last()` names the newest buffer that is certainly complete: the one released two starts ago.
        What this saves is the barrier packet `finish()` puts on the caller's stream every step."""
        if self.pending is None:
            return
        work, k = self.pending
        self.pending = None
        self._released = getattr(self, "_released", [])
        self._released.append((work, k))
        del self._released[:-2]                              # (keep the handles of the two exchanges that may still be in flight)

    def drain(self):
        """Waits (stream-side) for every released exchange: call before reading `out`, or before tearing the group down."""
        for work, _ in getattr(self, "_released", []):
            if work is not None:
                work.wait()
        self._released = []
        return self.finish(convert=False)

    def finish(self, convert=True):
        """-> (reward, code) of the exchange started last, on `dst` (None, None elsewhere).  convert=False: the codes stay the
        float column the shards wrote (no conversion kernel on the root's stream)."""
        if self.pending is None:
            return None, None
        work, k = self.pending
        self.pending = None
        if work is not None:
            work.wait()
        if self.rank != self.dst:
            return None, None
        out = self.out[k]
        return out[:, 0], (out[:, 1].to(torch.uint8) if convert else out[:, 1])

    def __call__(self, reward, game_over):
        self.start(reward, game_over)
        return self.finish()


class ScreensGather:
    """Per-step gather of every shard's screens into ONE contiguous [total_envs, ...] tensor on `dst`, double-buffered
    (SURVEY 8(e)): two destination tensors on the root and two observation buffers on every rank alternate, so the
    transfer of step t (async point-to-point over each remote GPU's direct xGMI link to the root) runs beside the
    kernels of step t + 1 and is only waited for when its buffers come up for reuse, two steps later.

        sg = ScreensGather(sim, counts, rank)
        loop:  sg.bind_next()            # the simulator renders this step into the free buffer pair
               sim.step(); sim.reset_done()
               sg.start()                # ship this step's screens
               screens = sg.latest()     # (root) the newest COMPLETE gathered tensor: the previous step's; None at first
        sg.drain()                       # -> the last step's gathered tensor

    The root renders straight into its slice of the destination (BatchedSimulator.bind_obs), so its own slab is never
    copied.  A context ring (context > 1) shifts frames in place and therefore needs ONE observation buffer: the gather
    then is waited for before the next step (`depth` = 1).

    The two buffers start zeroed.  With depth 2 every env must be drawn by every step: an env left out of a step
    (XWB_ACTION_SKIP, whose observation the kernels leave untouched) would show its frame of two steps ago -- callers that
    skip envs pass depth=1."""

    def __init__(self, sim, counts, rank, dst=0, group=None, depth=None):
        self.sim, self.counts, self.rank, self.dst, self.group = sim, list(counts), rank, dst, group
        self.world = len(counts)
        self.depth = 2 if sim.cfg.context == 1 else 1
        if depth is not None:
            assert depth == 1 or (depth == 2 and sim.cfg.context == 1), "depth 2 needs context == 1"
            self.depth = int(depth)
        shape = tuple(sim.obs.shape[1:])
        dtype, device = sim.obs.dtype, sim.obs.device
        n = counts[rank]
        self.off = sum(counts[:rank])
        if rank == dst:
            self.full = [torch.zeros((sum(counts),) + shape, dtype=dtype, device=device) for _ in range(self.depth)]
            self.local = [f[self.off:self.off + n] for f in self.full]
        else:
            self.full = [None] * self.depth
            self.local = [torch.zeros((n,) + shape, dtype=dtype, device=device) for _ in range(self.depth)]
        self.work = [None] * self.depth
        self.k = self.depth - 1          # buffer pair of the current step
        self.done_k = None               # newest pair whose gather was waited for

    def bind_next(self):
        self.k = (self.k + 1) % self.depth
        if self.work[self.k] is not None:             # its previous transfer must be over before it is rendered into
            self.work[self.k].wait()
            self.work[self.k] = None
            self.done_k = self.k
        self.sim.bind_obs(self.local[self.k])

    def start(self):
        self.work[self.k] = gather_slabs(self.local[self.k], self.full[self.k], self.counts, self.rank, self.dst,
                                         self.group, async_op=True)
        if self.depth == 1:
            self.work[self.k].wait()
            self.work[self.k] = None
            self.done_k = self.k

    def latest(self):
        """root: newest gathered tensor that is complete in stream order (with depth 2: the previous step's)."""
        k = self.k if self.depth == 1 else (self.k + 1) % self.depth
        if self.depth == 2 and self.work[k] is not None:
            self.work[k].wait()
            self.work[k] = None
            self.done_k = k
        return self.full[k] if (self.rank == self.dst and self.done_k is not None) else None

    def drain(self):
        for i in range(self.depth):
            k = (self.k + 1 + i) % self.depth
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None
        self.done_k = self.k
        return self.full[self.k] if self.rank == self.dst else None


# ------------------------------------------------------------------------------------------------------------------
# The same two exchanges issued by libxwb.so itself (include/xwb.h "multi-GPU": xwb_comm_*, xwb_gather_*): RCCL directly,
# below Python, so that C / C++ holders of a xwb_sim shard the same way.  torch.distributed is only used to hand the
# communicator's unique id from rank 0 to the others.
class LibComm:
    """An RCCL communicator made through the library (ncclGetUniqueId on rank 0 -> broadcast -> ncclCommInitRank)."""

    def __init__(self, rank, world, device, group=None):
        import ctypes as C
        from . import lib
        self.L = lib.load()
        self.rank, self.world, self.device = rank, world, device
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            lib.check(self.L.xwb_comm_unique_id(ident))
        if world > 1:
            on_gpu = dist.get_backend(group) == "nccl"
            t = torch.tensor(list(ident), dtype=torch.uint8, device="cuda:%d" % device if on_gpu else "cpu")
            dist.broadcast(t, 0, group=group)
            ident = (C.c_uint8 * 128)(*t.cpu().tolist())
        h = C.c_void_p()
        lib.check(self.L.xwb_comm_init_rank(ident, world, rank, device, C.byref(h)))
        self.h = h
        v = C.c_int32()
        lib.check(self.L.xwb_comm_version(C.byref(v)))
        self.version = v.value

    def close(self):
        if getattr(self, "h", None):
            self.L.xwb_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LibResultGather:
    """ResultGather through the library, BESIDE the step loop (xwb_gather_results_beside): the rows the last step call wrote into
    the batch's results ring (BatchedSimulator.bind_results_ring) are all-gathered on the communicator's own stream, ordered
    behind that step's kernel by the batch's epoch hand-over -- on a full-observation xworld batch nothing is enqueued on the
    caller's stream at all (the torch path's collective records an event there every step: ~6 us of idle GPU per step).  Same
    protocol as ResultGather: start() after the step call, then release() (nobody on this stream reads the result) or finish()
    (orders the current stream behind the exchange, -> (reward, code) columns on every rank).  Two output buffers alternate;
    marks 2 / 3 of the communicator are this object's (LibScreensGather uses 0 / 1).  Equal shards, one shard per rank.
    The results ring needs at least two slots (a one-slot ring's row is rewritten by the very next step while the exchange may
    still read it): with `slots` of them the row exchange i reads is rewritten by step i + slots, and released exchanges are
    never waited for, so every (slots - 2)-th start() orders the caller's stream behind the PREVIOUS exchange -- long complete
    in a healthy run, and the guarantee that no step overwrites a row an exchange still reads, however far the host runs ahead."""

    def __init__(self, sim, comm, counts, rank, stream=None):
        import ctypes as C
        from . import lib
        self.C, self.lib, self.L = C, lib, comm.L
        self.sim, self.comm, self.rank = sim, comm, rank
        self.counts = list(counts)
        self.c_counts = (C.c_int32 * len(counts))(*counts)
        self.total = sum(counts)
        dev = sim.obs.device
        self.out = [torch.empty((self.total, 2), dtype=torch.float32, device=dev) for _ in range(2)]
        self.slot = 0
        self.pending = None
        self.stream = stream
        self.by_epoch = None                                  # True once an exchange was ordered by the step's epoch
        ring = getattr(sim, "_results", None)
        self.ring_slots = int(ring.shape[0]) if ring is not None and ring.dim() == 3 else (1 if ring is not None else 0)
        self.started = 0

    def _st(self):
        s = self.stream if self.stream is not None else torch.cuda.current_stream()
        return self.C.c_void_p(int(s.cuda_stream))

    def start(self):
        if self.pending is not None:
            raise RuntimeError("LibResultGather.start() called twice without finish() / release()")
        ring = getattr(self.sim, "_results", None)
        slots = int(ring.shape[0]) if ring is not None and ring.dim() == 3 else (1 if ring is not None else 0)
        if slots == 1:
            raise RuntimeError("LibResultGather needs a results ring of at least two slots (bind_results_ring): the next step rewrites a "
                               "one-slot ring's row while the exchange may still read it")
        k = self.slot
        if slots >= 2 and self.started and self.started % max(1, slots - 2) == 0:
            self.lib.check(self.L.xwb_comm_wait(self.comm.h, 2 + (k ^ 1), self._st()))    # (see the class comment)
        self.started += 1
        self.slot ^= 1
        flag = self.C.c_int32()
        self.lib.check(self.L.xwb_gather_results_beside(self.sim.h, self.comm.h, self.C.c_void_p(self.out[k].data_ptr()), self.c_counts, None,
                                                        len(self.counts), self.rank, self._st(), self.C.byref(flag)))
        self.lib.check(self.L.xwb_comm_mark(self.comm.h, 2 + k))
        self.by_epoch = bool(flag.value)
        self.pending = k

    def release(self):
        self.pending = None

    def finish(self, convert=True):
        if self.pending is None:
            return None, None
        k = self.pending
        self.pending = None
        self.lib.check(self.L.xwb_comm_wait(self.comm.h, 2 + k, self._st()))
        out = self.out[k]
        return out[:, 0], (out[:, 1].to(torch.uint8) if convert else out[:, 1])

    def drain(self):
        """Orders the current stream behind every exchange issued so far (before reading `out`, or tearing down)."""
        self.pending = None
        for k in (0, 1):
            self.lib.check(self.L.xwb_comm_wait(self.comm.h, 2 + k, self._st()))
        return None, None


class LibScreensGather:
    """ScreensGather through the library (xwb_gather_screens_begin, or mode="grids": xwb_gather_grids_begin -- every shard
    ships its cell codes, 2 * max_dim^2 + 1 bytes per env, and the root draws all frames itself): same protocol (bind_next /
    start / latest / drain), same double buffering; the transfers run on the communicator's own stream inside the library.
    Every destination buffer has its own completion mark (xwb_comm_mark / xwb_comm_wait): waiting for buffer k does not wait
    for the transfer begun after it, so transfer t really runs beside the kernels of step t + 1.  `stream`: the stream the
    simulator's verbs are issued on (None = the default stream); begin and the waits are ordered on it."""

    def __init__(self, sim, comm, counts, rank, dst=0, depth=None, mode="screens", stream=None):
        import ctypes as C
        from . import lib
        assert mode in ("screens", "grids")
        self.C, self.lib, self.mode = C, lib, mode
        self.sim, self.comm, self.counts, self.rank, self.dst = sim, comm, list(counts), rank, dst
        self.stream = stream
        self.depth = 2 if sim.cfg.context == 1 else 1
        if depth is not None:
            self.depth = int(depth)
        shape = tuple(sim.obs.shape[1:])
        dtype, device = sim.obs.dtype, sim.obs.device
        n = counts[rank]
        self.off = sum(counts[:rank])
        if rank == dst:
            self.full = [torch.zeros((sum(counts),) + shape, dtype=dtype, device=device) for _ in range(self.depth)]
        else:
            self.full = [None] * self.depth
        if mode == "grids":
            # every shard keeps drawing into its own buffer; the root's tensor is drawn by the root from the gathered codes
            self.local = [None] * self.depth
        elif rank == dst:
            self.local = [f[self.off:self.off + n] for f in self.full]
        else:
            self.local = [torch.zeros((n,) + shape, dtype=dtype, device=device) for _ in range(self.depth)]
        self.c_counts = (C.c_int32 * len(counts))(*counts)
        self.busy = [False] * self.depth
        self.k = self.depth - 1
        self.done_k = None

    def _sp(self):
        s = self.stream
        return None if s is None else self.C.c_void_p(int(getattr(s, "cuda_stream", s)))

    def _end(self, k):
        if self.busy[k]:
            self.lib.check(self.comm.L.xwb_comm_wait(self.comm.h, k, self._sp()))
            self.busy[k] = False
            self.done_k = k

    def bind_next(self):
        self.k = (self.k + 1) % self.depth
        self._end(self.k)
        if self.local[self.k] is not None:
            self.sim.bind_obs(self.local[self.k])

    def start(self):
        C = self.C
        dst = self.full[self.k]
        begin = self.comm.L.xwb_gather_grids_begin if self.mode == "grids" else self.comm.L.xwb_gather_screens_begin
        self.lib.check(begin(self.sim.h, self.comm.h, C.c_void_p(dst.data_ptr()) if dst is not None else None,
                             self.c_counts, None, len(self.counts), self.rank, self.dst, self._sp()))
        self.lib.check(self.comm.L.xwb_comm_mark(self.comm.h, self.k))
        self.busy[self.k] = True
        if self.depth == 1:
            self._end(self.k)

    def latest(self):
        k = self.k if self.depth == 1 else (self.k + 1) % self.depth
        self._end(k)
        return self.full[k] if (self.rank == self.dst and self.done_k is not None) else None

    def drain(self):
        for i in range(self.depth):
            self._end((self.k + 1 + i) % self.depth)
        self.done_k = self.k
        return self.full[self.k] if self.rank == self.dst else None


class GridsGather:
    """The grids gather over torch.distributed (any backend): every shard packs its draw state (BatchedSimulator.pack_grids),
    the packed rows travel with gather_slabs, the root draws the whole batch (render_grids) into ONE contiguous tensor.
    Synchronous per step (the packed rows are a few MB: C5 7.9 MB per shard); the library path (LibScreensGather
    mode="grids") is the pipelined one.  context > 1: call after every verb that draws frames."""

    def __init__(self, sim, counts, rank, dst=0, group=None):
        self.sim, self.counts, self.rank, self.dst, self.group = sim, list(counts), rank, dst, group
        d = sim.cfg.max_dim
        n, device = counts[rank], sim.obs.device
        self.total = sum(counts)
        # one row per env: cell codes as 2 * d * d bytes, then the ring flag, then a pad byte (rows stay 2-byte aligned)
        self.row = 2 * d * d + 2
        self.cells = d * d
        self.grids = torch.zeros((n, d * d), dtype=torch.int16, device=device)
        self.flags = torch.zeros((n,), dtype=torch.uint8, device=device)
        self.rows = torch.zeros((n, self.row), dtype=torch.uint8, device=device)
        if rank == dst:
            self.all_rows = torch.zeros((self.total, self.row), dtype=torch.uint8, device=device)
            self.full = torch.zeros((self.total,) + tuple(sim.obs.shape[1:]), dtype=sim.obs.dtype, device=device)
        else:
            self.all_rows = self.full = None

    # the ScreensGather protocol (bench.py drives every gather through it); nothing is pipelined here: depth 1
    depth = 1

    def bind_next(self):
        pass

    def start(self):
        self()

    def latest(self):
        return self.full

    def drain(self):
        return self.full

    def __call__(self):
        self.sim.pack_grids(self.grids, self.flags)
        self.rows[:, :2 * self.cells] = self.grids.view(torch.uint8).view(-1, 2 * self.cells)
        self.rows[:, 2 * self.cells] = self.flags
        gather_slabs(self.rows, self.all_rows, self.counts, self.rank, self.dst, self.group)
        if self.rank != self.dst:
            return None
        g = self.all_rows[:, :2 * self.cells].contiguous().view(torch.int16).view(self.total, self.cells)
        f = self.all_rows[:, 2 * self.cells].contiguous()
        self.sim.render_grids(g, f, self.full)
        return self.full


def lib_gather_results(comm, packed, out, counts, rank):
    """xwb_gather_results: every shard's [n, 2] (reward, code) rows into out [total, 2] on every rank, on the current stream."""
    import ctypes as C
    from . import lib
    c_counts = (C.c_int32 * len(counts))(*counts)
    lib.check(comm.L.xwb_gather_results(comm.h, C.c_void_p(packed.data_ptr()), C.c_void_p(out.data_ptr()), c_counts, None,
                                        len(counts), rank, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out

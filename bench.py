#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched XWorld simulator on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload xworld7|xworld7_f32|xworld7_ego3|xworld8_ego5|xworld7_ego7|xworld8|xworld11|simple_game|simple_race]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over the whole batch with inputs resident in HBM:
SimulatorInterface::take_actions(act_rep=1) for every env under the built-in uniform random policy
(actions drawn on device), observation of every env materialised in HBM, then the reference example
loop's `if game_over: reset_game()` for the envs that finished (wavefront-ballot compaction, map
generation, re-render).  Default workload = BASELINE.json config C4 (the configuration the north-star
target is quoted on): XWorld2D 7x7, 84x84x3 uint8 planar BGR, 32 768 envs per GPU.

Timing: W untimed warm-up steps, then the same loop is spun (untimed) until the clocks are warm
(--spin-seconds, default 0.3 s), then EXACTLY K steps are timed between two barrier + synchronize fences,
max over ranks -- R times over (--repeats); `ms_per_step` / `value` are the MEDIAN region, the spread is
reported (`regions`).  A 20-step region of the default workload is 2.5 ms of GPU work: one region alone
measures the box's clock ramp, not the code.

Parity gate ("parity gates reported with every perf number"): every step of the whole run writes
(reward, game_over) of every env into a device-side record (xwb_bind_results_ring, no extra launches); after
the timed regions the record of a slab of envs is compared, bit for bit, with the CPU oracle's rollout of the
same envs from reset (the oracle is test infrastructure: it is only used here, outside every timed region, and
by the cpu_baseline leg).

Multi-GPU: the env batch is sharded by global env id (weak scaling: 32 768 envs per GPU).  `value` = screens
left device-resident, one RCCL all-gather of (reward, game_over) per step; `screens_gather` = the same loop
with every shard's screens gathered into one contiguous tensor on rank 0 (double-buffered: the transfer of step
t runs beside step t + 1; xGMI-link bound), with the link-bound ceiling beside it.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
XGMI_LINK_GBS = 153.6       # one xGMI link, per direction (MI355X_MICROARCH.md); every remote shard has ONE link to the root

WORKLOADS = {
    # name: (game, opts, envs per GPU)
    "xworld7": ("xworld", {"max_dim": 7, "num_blocks": 16, "color": True}, 32768),
    "xworld7_f32": ("xworld", {"max_dim": 7, "num_blocks": 16, "color": True, "obs_format": "float32"}, 32768),
    "xworld7_ego3": ("xworld", {"max_dim": 7, "num_blocks": 16, "color": True, "visible_radius": 3}, 32768),
    "xworld8_ego5": ("xworld", {"color": True, "visible_radius": 5}, 32768),                       # 80x80x3 frames
    "xworld7_ego7": ("xworld", {"max_dim": 7, "num_blocks": 16, "color": True, "visible_radius": 7}, 32768),
    # a geometry the span path cannot take (81 x 81 frames: include/xwb.h xwb_ego_render_path): the one-workgroup-per-env kernel
    "xworld11_ego9": ("xworld", {"max_dim": 11, "num_blocks": 30, "color": True, "visible_radius": 9}, 32768),
    "xworld8": ("xworld", {"color": True}, 32768),
    "xworld11": ("xworld", {"max_dim": 11, "num_blocks": 30, "color": True}, 32768),
    "simple_game": ("simple_game", {"array_size": 64}, 65536),
    "simple_race": ("simple_race", {"track_width": 20.0, "track_length": 100.0, "track_radius": 30.0}, 65536),
}
POLICY_SEED = 0x5EED
REC_BYTES_CAP = 6 << 30      # the per-step (reward, game_over) record: a ring of at most this many bytes


def make_sim(workload, n_envs, device, gid0, seed=0xC0FFEE, **extra):
    from xworld_amd.batched import BatchedSimulator
    game, opts, _ = WORKLOADS[workload]
    opts = dict(opts)
    opts.update(extra)
    if game == "xworld":
        opts["xwd_conf_path"] = os.path.join(ROOT, "xworld_amd", "confs", "navigation2d.json")       # the five XWorld3DNav tasks
        opts["task_mode"] = "lang_acquisition"
    return BatchedSimulator(game, opts, num_envs=n_envs, device=device, env_gid0=gid0,
                            seed=seed, policy_seed=POLICY_SEED)


def algorithmic_bytes(workload, sim):
    """logical bytes per env-step, and per env per launch of the dominant kernel."""
    game = WORKLOADS[workload][0]
    if game == "simple_game":
        a = sim.cfg.array_size
        return 27 + a, 27 + a, "sg_kernel"
    if game == "simple_race":
        return 57, 57, "race_kernel"
    d = sim.cfg.max_dim
    c = sim.screen_dims[2]
    if sim.cfg.visible_radius:
        # egocentric: the frame is (r * (84 / r))^2 pixels; its algorithmic bytes are the frame written + the grid read.  The
        # whole-batch render is four launches on the span path (timed together, on the stream they run on: cell table,
        # evaluated pixels, terminal frames, gather -- the gather alone moves ~ all the bytes), one otherwise
        obs = c * sim.screen_dims[0] * sim.screen_dims[1]
        name = ("xw_ego_cells_kernel + xw_ego_eval_kernel + xw_ego_gather_list_kernel + xw_ego_gather_kernel"
                if sim.ego_render_path == "span" else "xw_render_ego_kernel")
        return 33 + 2 * d * d + obs, 2 * d * d + obs, name
    obs = c * 144 * d * d * (4 if sim.obs_is_float else 1)      # float32 variant: obs term x 4
    return 33 + 2 * d * d + obs, 2 * d * d + obs, "xw_render_all_kernel"


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in separate runs, gfx950 corrections applied by tools/summarize_prof.py); None if not profiled.
    The number is a STORED measurement, not something this run measured: the line says which file, which commit and which
    source fingerprint it comes from, and `traffic_stale` when the sources this run executes are not those."""
    best = None
    for d in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        f = os.path.join(ROOT, "profiles", d, "traffic_%s.json" % workload)
        if os.path.exists(f):
            best = f
    if not best:
        return None, {"traffic_source": None}
    with open(best) as fh:
        t = json.load(fh)
    from xworld_amd import build
    now = build.source_fingerprint()
    return t["traffic_bytes_per_launch"], {"traffic_source": os.path.relpath(best, ROOT), "traffic_commit": t.get("commit", "unknown"),
                                           "traffic_source_sha16": t.get("source_sha16"), "source_sha16": now,
                                           "traffic_stale": t.get("source_sha16") != now}


def write_ceiling(buf, reps=24):
    """The bandwidth anchor of THIS run: a pure write stream of the observation batch's size (hipMemsetAsync through
    torch.Tensor.zero_, and a fill kernel), timed with events on the current stream, in this process, on this box -- what
    `roofline.achieved` can be read against besides the 8 TB/s spec.  It runs AFTER the timed and the event regions (round 4
    ran it between the spin and the timed regions: the first regions then paid for whatever it disturbed and the driver's
    20-step median landed in that ramp), on a buffer allocated before the warm-up and kept until the process ends."""
    import torch
    nbytes = buf.numel()
    out = {}
    for name, fn in (("memset", lambda: buf.zero_()), ("fill_kernel", lambda: buf.fill_(7))):
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / reps * 1e3
        out[name] = {"us": us, "GBps": nbytes / us / 1e3}
    return out


# ---- N > 1: a hung collective must cost a block of the line, not the line.  Every measurement behind the main one runs as a
# PHASE with its own deadline; a daemon thread on every rank watches it.  When a phase overruns, rank 0 prints the line as far
# as it got (the overrun phase and everything not reached recorded as {"error": ...}) and every rank leaves with os._exit(0):
# a process group whose RCCL kernels hang cannot be torn down politely.
WATCH = {"line": None, "phase": None, "deadline": None, "rank": 0, "fired": False, "pending": []}


def _watchdog_loop():
    import threading
    while True:
        time.sleep(0.5)
        dl = WATCH["deadline"]
        if dl is None or time.perf_counter() < dl:
            continue
        WATCH["fired"] = True
        if WATCH["rank"] == 0 and WATCH["line"] is not None:
            line = dict(WATCH["line"])
            msg = "watchdog: phase '%s' exceeded its %.0f s" % (WATCH["phase"], WATCH.get("budget", 0.0))
            line["watchdog"] = {"error": msg, "not_reached": list(WATCH["pending"])}
            sys.stdout.write(json.dumps(line) + "\n")
            sys.stdout.flush()
        elif WATCH["rank"] == 0 and not WATCH.get("printed"):
            sys.stderr.write("bench.py watchdog: phase '%s' overran before the main measurement was complete\n" % WATCH["phase"])
        os._exit(0 if (WATCH["line"] is not None or WATCH.get("printed")) else 3)


def watch_start(rank):
    import threading
    WATCH["rank"] = rank
    threading.Thread(target=_watchdog_loop, daemon=True).start()


class phase:
    """with phase("name", seconds): ... -- the block's deadline for the watchdog (None: no deadline)"""

    def __init__(self, name, seconds):
        self.name, self.seconds = name, seconds

    def __enter__(self):
        WATCH["phase"], WATCH["budget"] = self.name, self.seconds or 0.0
        WATCH["deadline"] = None if not self.seconds else time.perf_counter() + self.seconds
        self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        WATCH["deadline"] = None
        WATCH.setdefault("phase_seconds", {})[self.name] = round(time.perf_counter() - self.t0, 2)
        return False


def region_trend(regions):
    """(median of the last third - median of the first third) / median of all: a settled run is within +-1 %."""
    k = max(1, len(regions) // 3)
    med = statistics.median(regions)
    return (statistics.median(regions[-k:]) - statistics.median(regions[:k])) / med if med > 0 else 0.0


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    return O


def oracle_rollout(workload, n, steps, gid0, seed, render):
    """The CPU restatement's rollout of envs gid0 .. gid0 + n - 1 from reset (same RNG keys, same policy)."""
    O = _oracle()
    game, sim_opts, _ = WORKLOADS[workload]
    if game == "simple_game":
        return O.sg_rollout(n, sim_opts["array_size"], steps, POLICY_SEED, env_gid0=gid0)
    if game == "simple_race":
        return O.race_rollout(n, O.race_cfg(), seed, steps, POLICY_SEED, env_gid0=gid0)
    pal = O.Palette(O.NAV_SUBTREES)
    d = sim_opts.get("max_dim", 8)
    cfg = O.xw_cfg(map_kind=0, max_dim=d, dim=d, num_goals=4, num_blocks=sim_opts.get("num_blocks", 16),
                   color=1, seed=seed, tasks=[0, 1, 2, 3, 4], visible_radius=sim_opts.get("visible_radius", 0))
    return O.xw_rollout(n, cfg, pal, steps, POLICY_SEED, env_gid0=gid0, render=render)


def parity_gate(workload, rec, calls, slots, fused, gid0, seed, slab, calls_before):
    """Compare the device's per-step record of envs [0, slab) with the oracle's rollout of the same envs.
    rec: [slots, n, 2] ring written by the step kernels; `calls` step calls were recorded, each `fused` steps long,
    after `calls_before` unrecorded ones (the probe)."""
    import numpy as np
    steps = (calls_before + calls) * fused
    ref = oracle_rollout(workload, slab, steps, gid0, seed, render=False)
    first = max(0, calls - slots)                     # oldest call still in the ring
    got = rec[:, :slab, :].cpu().numpy()              # [slots, slab, 2]
    mism = 0
    for k in range(first, calls):
        t = (calls_before + k + 1) * fused - 1        # a fused call keeps its last step
        row = got[k % slots]
        mism += int(np.count_nonzero(row[:, 0].view(np.uint32) != ref.rewards[t].view(np.uint32)))
        mism += int(np.count_nonzero(row[:, 1].astype(np.uint8) != ref.codes[t]))
    return {"checked_env_steps": (calls - first) * slab, "mismatches": mism, "envs": slab, "scope": "a slab of envs, every recorded step: reward bits + game_over code (not the frames, not the whole batch)",
            "step_calls": [calls_before + first, calls_before + calls], "against": "oracle/liboracle.so rollout from reset, reward bits + game_over code"}


def cpu_baseline(workload, seed, seconds_target=8.0):
    """The oracle (CPU restatement of the reference path, kind = "port") timed on this box's host cores on a bounded
    sample of the same workload (same loop: game_over? -> reset; get_state; random action; take_actions incl. screen):
    first one thread (calibration, also reported), then one independent env batch per thread on every core (ctypes
    releases the GIL; the oracle keeps no global state) -- `value` / `cores` are the all-core figures."""
    import threading

    def rollout(n, steps, gid0):
        oracle_rollout(workload, n, steps, gid0, seed, render=True)

    # one thread: grow the sample until a call takes about two seconds
    n, steps = 8, 50
    while True:
        t0 = time.perf_counter()
        rollout(n, steps, 0)
        dt = time.perf_counter() - t0
        if dt >= 1.5 or n >= 1 << 22:
            break
        n *= 4 if dt < 0.4 else 2
    single = n * steps / dt
    # every core: each thread keeps running batches of its own (bounded memory) until the time budget is used up
    cores = min(os.cpu_count() or 1, 64)
    n_thr = min(n, 16384)
    counts = [0] * cores

    def worker(k):
        t_end = time.perf_counter() + seconds_target
        i = 0
        while time.perf_counter() < t_end:
            rollout(n_thr, steps, 1000003 * (k + 1) + 7919 * i)
            counts[k] += n_thr * steps
            i += 1
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(cores)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    wall = time.perf_counter() - t0
    done = sum(counts)
    return {"value": done / wall, "unit": "env-steps/s", "cores": cores, "kind": "port", "single_thread_value": single,
            "sample": "%d env-steps of %s through oracle/liboracle.so (reset, step, teacher, 64px-canvas render) on %d threads "
                      "in %.1f s; one thread alone: %d env-steps in %.1f s" % (done, workload, cores, wall, n * steps, dt)}


def gather_block(sim, mode, lib_comm, depth, world, n_local, steps, sg_regions):
    """One screens_gather block from its timed regions: what crossed the links, the bound that applies, the ceilings."""
    shard_bytes = n_local * sim.obs_bytes_per_env
    sg_med = statistics.median(sg_regions)
    if mode == "screens":
        link_bytes = shard_bytes
        bound = {"bound": "xGMI link (every remote shard has ONE link to the root)"}
    else:
        mode_note = {"local_render": mode != "grids_nodraw"}
        link_bytes = n_local * (2 * sim.cfg.max_dim ** 2 + 1)
        # the root writes every frame of the whole batch once more: its HBM write stream bounds the step
        root_s = world * shard_bytes / (HBM_PEAK_GBS * 1e9)
        bound = {"bound": "the root's HBM write stream (it draws all %d frames from the gathered cell codes)" % (n_local * world),
                 "root_render_bound_ms_per_step": root_s * 1e3, "root_render_bound_ceiling": n_local * world / root_s, **mode_note}
    link_s = link_bytes / (XGMI_LINK_GBS * 1e9)
    if lib_comm is not None:
        by = "libxwb.so (xwb_gather_%s_begin + xwb_comm_mark / _wait: ncclSend / ncclRecv on the communicator's stream)" % mode.split("_")[0]
    elif mode.startswith("grids"):
        by = "torch.distributed batch_isend_irecv of the packed cell codes + xwb_xw_render_grids on the root (sharding.GridsGather)"
    else:
        by = "torch.distributed batch_isend_irecv"
    return {"mode": mode, "ms_per_step": sg_med / steps * 1e3, "value": n_local * world * steps / sg_med, "unit": "env-steps/s",
            "bytes_into_root_per_step": link_bytes * (world - 1), "link_bound_ms_per_step": link_s * 1e3,
            "link_bound_ceiling": n_local * world / link_s, "link_GBps_assumed": XGMI_LINK_GBS,
            "achieved_GBps_per_link": link_bytes / (sg_med / steps) / 1e9,
            "overlap": "double-buffered: transfer of step t beside the kernels of step t+1" if depth == 2 else
                       ("none (context ring)" if sim.cfg.context > 1 else "none (the few MB of cell codes are gathered synchronously)"),
            "issued_by": by, "regions_ms_per_step": {"min": min(sg_regions) / steps * 1e3, "max": max(sg_regions) / steps * 1e3},
            **bound}


def make_gather(sim, mode, lib_comm, counts, rank):
    """mode: screens | grids | grids_nodraw (= grids with every shard's own pixel stores off: xwb_xw_set_draw(sim, 0))"""
    from xworld_amd import sharding
    m = "grids" if mode.startswith("grids") else "screens"
    sim.set_draw(mode != "grids_nodraw") if mode.startswith("grids") else None
    if lib_comm is not None:
        return sharding.LibScreensGather(sim, lib_comm, counts, rank, mode=m)
    return sharding.GridsGather(sim, counts, rank) if m == "grids" else sharding.ScreensGather(sim, counts, rank)


def gather_regions(sim, mode, lib_comm, counts, rank, world, n_local, K, R, args, set_screens, timed_region, fence):
    """R timed regions of the loop with every step's frames gathered on rank 0 by `mode`; returns (block, gather object)."""
    g = make_gather(sim, mode, lib_comm, counts, rank)
    set_screens(g)
    try:
        for _ in range(2 * K):
            set_screens.one_step()
        regs = [timed_region() for _ in range(R)]
        fence()
    finally:
        set_screens(None)
        if mode == "grids_nodraw":                           # back to a batch that draws, its own buffer made current from its draw state
            import torch
            sim.set_draw(True)
            d = sim.cfg.max_dim
            gr = torch.empty((n_local, d * d), dtype=torch.int16, device=sim.obs.device)
            fl = torch.empty((n_local,), dtype=torch.uint8, device=sim.obs.device)
            sim.pack_grids(gr, fl)
            sim.render_grids(gr, fl, sim.obs)
    return gather_block(sim, mode, lib_comm, g.depth, world, n_local, args.steps, regs), g


def c5_block(args, world, rank, local_rank, dev, K, lib_comm_main=None):
    """BASELINE.json config C5 -- XWorld2D 11x11, 32 768 envs per GPU (262 144 over 8), "sharded 8 x MI355X with RCCL gather
    of screens" -- as a sub-object of the N > 1 line: the same loop as the main measurement on the xworld11 workload, once
    with the screens left device-resident (value) and once with every shard's screens gathered into one tensor on rank 0
    (double-buffered), beside the xGMI link ceiling.  Fewer regions than the main line (3): it is a second measurement."""
    import torch
    import torch.distributed as dist
    from xworld_amd import sharding
    n_local = args.envs_per_gpu or WORKLOADS["xworld11"][2]
    sim = make_sim("xworld11", n_local, local_rank, rank * n_local, args.seed)
    counts = [n_local] * world
    # (eight slots: with the exchanges released, a slot is rewritten eight steps after it was shipped)
    packed = torch.zeros((8, n_local, 2), dtype=torch.float32, device=dev)
    sim.bind_results_ring(packed)
    # (the per-step results as in the main measurement: through the library's communicator when it is up)
    results = sharding.LibResultGather(sim, lib_comm_main, counts, rank) if lib_comm_main is not None else sharding.ResultGather(counts, rank, dev)
    state = {"screens": None, "calls": 0}

    def one_step():
        if state["screens"] is not None:
            state["screens"].bind_next()
        state["calls"] += 1
        sim.step()
        sim.reset_done()
        (results.finish(convert=False) if args.results_wait else results.release())
        results.start(packed=packed[(state["calls"] - 1) % 8])
        if state["screens"] is not None:
            state["screens"].start()

    def fence():
        results.drain()
        if state["screens"] is not None:
            state["screens"].drain()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def region():
        fence()
        t0 = time.perf_counter()
        for _ in range(K):
            one_step()
        fence()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    for _ in range(max(10, args.warmup)):
        one_step()
    t_end = time.perf_counter() + args.spin_seconds
    while True:                                          # (every rank spins the same number of steps: the count is agreed on)
        for _ in range(50):
            one_step()
        flag = torch.tensor([1 if time.perf_counter() < t_end else 0], device=dev)
        dist.broadcast(flag, 0)
        if not int(flag.item()):
            break
    dev_regions = [region() for _ in range(3)]
    lib_comm = sharding.LibComm(rank, world, local_rank) if args.exchange == "lib" else None
    gathers = {}
    for mode in (("screens", "grids", "grids_nodraw") if args.gather == "both" else (args.gather,)):
        try:
            g = make_gather(sim, mode, lib_comm, counts, rank)
            state["screens"] = g
            for _ in range(4):
                one_step()
            regs = [region() for _ in range(3)]
            fence()
            gathers[mode] = gather_block(sim, mode, lib_comm, g.depth, world, n_local, K, regs)
        except Exception as e:
            gathers[mode] = {"error": "%s: %s" % (type(e).__name__, e)}
        state["screens"] = None
    errs = sim.check_errors()
    d_med = statistics.median(dev_regions)
    first = "screens" if "screens" in gathers else "grids"
    out = {"workload": "xworld11", "config": "BASELINE C5: 11x11, 132x132x3 u8, %d envs per GPU, %d in all" % (n_local, n_local * world),
           "value": n_local * world * K / d_med, "unit": "env-steps/s", "ms_per_step": d_med / K * 1e3,
           "exchange": "all_gather(reward,done) per step (%s), screens device-resident" % ("libxwb.so, beside the step loop" if lib_comm_main is not None else "torch.distributed"),
           "regions": 3, "steps_per_region": K,
           "action_errors": errs, "screens_gather": gathers[first]}
    if first == "screens" and "grids" in gathers:
        out["screens_gather"] = dict(gathers["screens"], grids=gathers["grids"])
        if "grids_nodraw" in gathers:
            out["screens_gather"]["grids_no_local_render"] = gathers["grids_nodraw"]
    sim.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=9, help="timed K-step regions; the median one is reported")
    ap.add_argument("--spin-seconds", type=float, default=0.3, help="untimed run of the same loop before the timed regions (clock ramp)")
    ap.add_argument("--seed", type=lambda v: int(v, 0), default=0xC0FFEE, help="env RNG seed (xwb-rng-v1 key word 0)")
    ap.add_argument("--workload", default="xworld7", choices=list(WORKLOADS))
    ap.add_argument("--envs-per-gpu", type=int, default=0)
    ap.add_argument("--no-screens-gather", action="store_true", help="N > 1: skip the screens-gather-inclusive regions")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--parity-envs", type=int, default=4096, help="envs whose whole per-step record is checked against the oracle "
                    "(a slab of the batch, not all of it: `parity.envs` of `config.envs_per_gpu`)")
    ap.add_argument("--frame-envs", type=int, default=64, help="envs whose frames are checked against the oracle's renderer ...")
    ap.add_argument("--frame-steps", type=int, default=24, help="... over this many steps from reset (untimed, before the warm-up)")
    ap.add_argument("--autoreset", action="store_true", help="use the fused step+reset+single-render call")
    ap.add_argument("--fused", type=int, default=1, help="simple games only: steps per launch (xwb_step_n); --steps must be "
                    "a multiple; every step still writes its reward / code / observation")
    ap.add_argument("--exchange", default="auto", choices=["auto", "torch", "lib"], help="N > 1: who issues the exchanges.  torch: "
                    "torch.distributed for everything.  lib: libxwb.so's own RCCL calls for everything (results beside the step loop: "
                    "xwb_gather_results_beside; screens: xwb_gather_screens_begin / grids; backend nccl only).  auto (default): the "
                    "per-step results through the library when its communicator comes up on every rank within 90 s (else torch), the "
                    "screens gathers through torch.distributed")
    ap.add_argument("--gather", default="both", choices=["screens", "grids", "grids_nodraw", "both"], help="N > 1, full observation: what crosses "
                    "the links per step -- every shard's pixels (screens), or its cell codes with the root drawing all frames "
                    "(grids: xwb_gather_grids_begin, needs --exchange lib), or one set of regions each (both; grids only with --exchange lib)")
    ap.add_argument("--c5", action="store_true", help="N > 1: add the BASELINE C5 block (xworld11); on by itself at N = 8")
    ap.add_argument("--force-exchange", action="store_true", help="N = 1: initialise torch.distributed (world size 1) and issue the "
                    "N > 1 run's exchanges all the same -- the RCCL all-gather of results, the gather objects -- so that no line of "
                    "the multi-GPU path runs for the first time on the 8-GPU box")
    ap.add_argument("--results-wait", action="store_true", help="N > 1: order the step's stream behind every step's all-gather of "
                    "(reward, done) (ResultGather.finish one step late) instead of letting it go (release): what a trainer that reads "
                    "the results on the root every step pays")
    ap.add_argument("--phase-timeout", type=float, default=240.0, help="N > 1: seconds a secondary measurement (screens gather mode, "
                    "C5 block, step_autoreset) may take before the watchdog prints the line without it; the main measurement gets 3x")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo with every rank "
                    "on the visible GPUs modulo their count only exercises the N > 1 code path on a smaller box)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            from xworld_amd.sharding import init_nccl
            init_nccl(torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(0)
        if args.force_exchange:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            from xworld_amd.sharding import nccl_init_kwargs
            kw = nccl_init_kwargs(torch.device("cuda", 0)) if args.backend == "nccl" else {}
            if "MASTER_ADDR" in os.environ and "MASTER_PORT" in os.environ:
                dist.init_process_group(args.backend, world_size=1, rank=0, **kw)
            else:
                import socket
                sk = socket.socket()
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
                sk.close()
                dist.init_process_group(args.backend, init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0, **kw)
    dev = torch.device("cuda", local_rank)
    guarded = world > 1 or args.force_exchange
    PT = args.phase_timeout if guarded else None
    watch_start(rank)
    WATCH["pending"] = ["main", "step_autoreset", "screens_gather", "c5", "parity"]
    WATCH["phase"], WATCH["budget"] = "main (set-up, frame gate, warm-up)", 3 * (PT or 0.0)
    WATCH["deadline"] = None if not PT else time.perf_counter() + 3 * PT
    n_local = args.envs_per_gpu or WORKLOADS[args.workload][2]
    sim = make_sim(args.workload, n_local, local_rank, rank * n_local, args.seed)
    per_step, per_launch, kernel_name = algorithmic_bytes(args.workload, sim)
    is_xworld = WORKLOADS[args.workload][0] == "xworld"
    fused = args.fused if not is_xworld else 1
    assert fused == 1 or args.steps % fused == 0, "--steps must be a multiple of --fused"
    K, W, R = args.steps // fused, -(-args.warmup // fused), max(1, args.repeats)   # in step CALLS (warm-up rounded up)

    from xworld_amd import sharding
    counts = [n_local] * world
    forced = world == 1 and args.force_exchange
    # --exchange lib: the library's own communicator carries the per-step results too, beside the step loop (no packet on the
    # step's stream: xwb_gather_results_beside); the torch path stays the default
    lib_comm_main, lib_note = None, None
    if (world > 1 or forced) and args.exchange in ("lib", "auto") and args.backend == "nccl":
        # the library's communicator is made in a thread with a deadline: a second ncclCommInitRank that hangs on some box must
        # cost the faster exchange, not the run; every rank then agrees (over the torch group, which is up) on what to use
        import threading
        box = {}

        def make_comm():
            try:
                box["comm"] = sharding.LibComm(rank, world, local_rank)
            except Exception as e:                       # noqa: BLE001
                box["err"] = "%s: %s" % (type(e).__name__, e)
        th = threading.Thread(target=make_comm, daemon=True)
        th.start()
        th.join(90.0)
        ok = torch.tensor([1 if "comm" in box else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()):
            lib_comm_main = box["comm"]
        else:
            lib_note = box.get("err", "timed out after 90 s" if th.is_alive() else "another rank failed")
            if args.exchange == "lib":
                raise RuntimeError("--exchange lib: the library's communicator did not come up: %s" % lib_note)
    if lib_comm_main is not None:
        results = sharding.LibResultGather(sim, lib_comm_main, counts, rank)
    else:
        results = sharding.ResultGather(counts, rank, dev, force_collective=forced) if (world > 1 or forced) else None
    with_screens = (world > 1 or forced) and not args.no_screens_gather
    screens = None                                   # ScreensGather while the screens regions run

    loop = {"autoreset": args.autoreset}             # which call sequence one_step() issues
    calls = [0]                                      # step calls so far == the record slot counter
    rec = [None]

    def exchange_results():
        if results is None:
            return
        # finish the gather of the previous step (it ran beside this step's kernels), start this step's: the step
        # kernel wrote (reward, code) straight into the record's slot, no packing kernels
        (results.finish(convert=False) if args.results_wait else results.release())
        results.start(packed=rec[0][(calls[0] - 1) % rec[0].shape[0]])

    def one_step():
        if screens is not None:
            screens.bind_next()
        calls[0] += 1
        if fused > 1:                                    # `fused` steps in one launch (built-in policy, auto-reset)
            sim.step_n(fused)
            return
        if loop["autoreset"]:
            sim.step_autoreset()
            exchange_results()
        else:
            sim.step()
            sim.reset_done()
            # this step's results: the (reward, code) rows the step wrote into the record, which reset_done leaves alone.  Behind
            # reset_done, whose list render publishes the epoch of a fused step + render launch: the exchange waits for that
            # epoch on its own stream instead of an event recorded on this one
            exchange_results()
        if screens is not None:
            screens.start()                              # the frames the next policy step would see

    def fence():
        if results is not None:
            results.drain()
        if screens is not None:
            screens.drain()
        torch.cuda.synchronize()
        if world > 1 or forced:
            dist.barrier()
        torch.cuda.synchronize()

    def set_screens(g):
        nonlocal screens
        screens = g
    set_screens.one_step = one_step

    def bcast_int(v):
        if world == 1:
            return int(v)
        t = torch.tensor([int(v)], dtype=torch.int64, device=dev)
        dist.broadcast(t, 0)
        return int(t.item())

    # ---- plan the run: the record ring must exist before the first step; its length needs the step count, which needs
    # the spin length, which needs a step time: a short untimed probe on a throw-away ring gives it ----
    probe = torch.zeros((2, n_local, 2), dtype=torch.float32, device=dev)
    sim.bind_results_ring(probe)
    rec[0] = probe
    # ---- frame gate (untimed, before anything else runs): the frame every policy step sees, for a slab of rank 0's envs over
    # the first steps, against the oracle's own renderer (position-weighted checksums, oracle/oracle.h orc_obs_checksum) ----
    frames = None
    if not args.no_parity and fused == 1 and not (is_xworld and sim.obs_is_float):      # (float32 frames: the oracle renders uint8)
        import numpy as np
        fe, fs = min(args.frame_envs, n_local), args.frame_steps
        got = []
        for _ in range(fs):
            fence()
            got.append(sim.obs[:fe].contiguous().view(torch.uint8).reshape(fe, -1).cpu().numpy() if rank == 0 else None)
            one_step()
        fence()
        if rank == 0:
            ref = oracle_rollout(args.workload, fe, fs, 0, args.seed, render=True)
            O = _oracle()
            bad = sum(int(np.count_nonzero(O.obs_checksum_np(got[t]) != ref.obs_ck[t])) for t in range(fs))
            frames = {"checked_frames": fe * fs, "mismatches": bad, "envs": fe, "steps": fs,
                      "against": "oracle/liboracle.so renderer (64 px canvas + cv::resize restatement), checksum of every byte of the frame"}
    for _ in range(3):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(10):
        one_step()
    fence()
    est = (time.perf_counter() - t0) / 10
    spin_calls = bcast_int(min(200000, math.ceil(max(0.0, args.spin_seconds) / max(est, 1e-7))))
    screens_regions = R if with_screens else 0
    total_calls = 13 + W + spin_calls + 24 * K + 2 * R * K + screens_regions * K + (2 * K if with_screens else 0) + 4 * K
    slots = int(max(2, min(total_calls, REC_BYTES_CAP // (n_local * 8))))
    rec[0] = torch.zeros((slots, n_local, 2), dtype=torch.float32, device=dev)
    # the probe's 13 calls happened with another ring: restart the slot counter with the library's (bind resets it)
    sim.bind_results_ring(rec[0])
    probe_calls = calls[0]
    calls[0] = 0
    # the write-ceiling anchor's buffer: allocated before the warm-up, freed with the process (nothing is allocated or freed
    # between the spin and the last timed region)
    ceiling_buf = torch.empty(int(n_local * sim.obs_bytes_per_env), dtype=torch.uint8, device=dev)

    for _ in range(W):
        one_step()
    # clocks warm, caches and allocator settled: spin by WALL TIME (a count derived from the cold probe above is too short on a
    # fresh box: its first steps run several times slower than the settled loop), every rank the same number of steps ...
    spun = 0
    t_end = time.perf_counter() + max(0.0, args.spin_seconds)
    while spun < 400000:
        for _ in range(50):
            one_step()
        spun += 50
        fence() if world > 1 else torch.cuda.synchronize()
        if not bcast_int(1 if time.perf_counter() < t_end else 0):
            break
    spin_calls = spun

    host_issue = []                                      # seconds the host spent issuing the K steps of each region (before the fence)

    def timed_region():
        fence()
        t0 = time.perf_counter()
        for _ in range(K):
            one_step()
        host_issue.append(time.perf_counter() - t0)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    # ... then unreported K-step regions until three in a row agree within 1 % (at most 24: a short region on a box that has
    # just been handed over measures the box settling, not the code)
    WATCH["deadline"] = None if not PT else time.perf_counter() + 3 * PT
    WATCH["phase"], WATCH["budget"] = "main", 3 * (PT or 0.0)
    settle = []
    while len(settle) < 24:
        settle.append(timed_region())
        ok = len(settle) >= 3 and max(settle[-3:]) <= 1.01 * min(settle[-3:])
        if bcast_int(1 if ok else 0):
            break

    # ---- the timed regions: exactly K steps each between two barrier + synchronize fences, max over ranks ----
    del host_issue[:]
    regions = [timed_region() for _ in range(R)]
    dt_med = statistics.median(regions)
    host_us_per_step = statistics.median(host_issue) / args.steps * 1e6

    # ---- R more regions with hipEvents around every launch of the dominant kernel (on its launch stream, recorded
    # inside libxwb) to get that kernel's average duration for the roofline ----
    sim.profile_begin()
    ev_regions = [timed_region() for _ in range(R)]
    kern = "render" if is_xworld else "step"
    kern_us, kern_n = sim.profile_end(kern)
    # every kernel of the step, not only the dominant one (each timed on the stream it runs on; the map generator runs on the
    # batch's internal queue BESIDE the render, so the sum is not the step time)
    kernels_us = {}
    for kname in (("step", "render", "reset", "list") if is_xworld else ("step", "reset")):
        us, nl = sim.profile_end(kname)
        if nl:
            kernels_us[kname] = {"avg_us": us, "launches": nl}
    sim.profile_stop()
    path_default = sim.step_path()
    # the run's own bandwidth anchor (rank-local, clocks warm), behind everything the headline and the roofline are read from
    ceiling = write_ceiling(ceiling_buf)
    for _ in range(K):                                   # and the loop itself warm again before the secondary measurements
        one_step()
    WATCH["deadline"] = None
    # N ranks on N devices, in the record itself: every rank's device as torch names it, and RCCL's own count of the communicator
    ranks_seen = {}
    if world > 1 or forced:
        mine = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.get_device_name(dev), "index": torch.cuda.current_device(),
                "uuid": str(getattr(torch.cuda.get_device_properties(dev), "uuid", ""))}
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        ranks_seen = {"ranks": everyone, "ranks_seen": len({r["rank"] for r in everyone}),
                      "devices_seen": len({r["uuid"] or r["index"] for r in everyone})}

    def core_line():
        total_envs = n_local * world
        value = total_envs * args.steps / dt_med
        # algorithmic bytes of one launch = per-step bytes x the steps that launch runs
        achieved = n_local * per_launch * fused / (kern_us * 1e-6) / 1e9 if kern_us > 0 else 0.0
        traffic, traffic_info = measured_traffic(args.workload) if n_local == WORKLOADS[args.workload][2] else (None, {"traffic_source": None})
        line = {
            "metric": "env-steps/sec (batched random policy)",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_med / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32 (f64 trig)" if WORKLOADS[args.workload][0] == "simple_race" else
                      ("u8 state, f32 frames (pixel * 1/255)" if sim.obs_is_float else "u8")),
            "data": "synthetic",
            "config": {"workload": args.workload, "envs_per_gpu": n_local, "total_envs": total_envs,
                       "obs": list(sim.obs.shape[1:]), "seed": args.seed, "policy": "uniform random, drawn on device",
                       "loop": ("step_n(%d): %d steps per launch, auto-reset" % (fused, fused)) if fused > 1 else
                               ("step_autoreset" if args.autoreset else "step + reset_done"),
                       "exchange": ("all_gather(reward,done) per step (%s; %s), screens device-resident" %
                                    ("the step's stream waits for it one step late" if args.results_wait else "released: no reader on the step's stream",
                                     ("libxwb.so on its communicator's stream, ordered by the step's %s" % ("epoch" if results.by_epoch else "event"))
                                     if lib_comm_main is not None else "torch.distributed"))
                                   if (world > 1 or forced) else "none",
                       "parallelism": "env-sharded x%d" % world},
            "regions": {"repetitions": R, "statistic": "median", "steps_per_region": args.steps,
                        "ms_per_step_min": min(regions) / args.steps * 1e3, "ms_per_step_max": max(regions) / args.steps * 1e3,
                        "ms_per_step_all": [r / args.steps * 1e3 for r in regions],
                        "trend": region_trend(regions),
                        "untimed_before": {"warmup_steps": args.warmup, "spin_steps": spin_calls * fused,
                                           "spin_seconds_target": args.spin_seconds, "probe_steps": probe_calls * fused,
                                           "settle_regions": len(settle)}},
            "path": path_default,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kernel_name,
                         "write_ceiling_GBps": ceiling["memset"]["GBps"], "write_ceiling": ceiling,
                         "frac_of_write_ceiling": achieved / ceiling["memset"]["GBps"] if ceiling["memset"]["GBps"] else None,
                         "kernels_us": kernels_us,
                         "kernel_avg_us": kern_us, "kernel_launches": kern_n,
                         "algorithmic_bytes_per_launch": n_local * per_launch * fused,
                         "algorithmic_bytes_per_env_step": per_step,
                         "step_loop_GBps": total_envs * per_step * args.steps / dt_med / 1e9,
                         "step_loop_frac": total_envs * per_step * args.steps / dt_med / 1e9 / HBM_PEAK_GBS / world,
                         **traffic_info},
            "timed_with_events_ms_per_step": statistics.median(ev_regions) / args.steps * 1e3,
            "host_us_per_step": host_us_per_step,
            "rccl": dict(sharding.backend_info(), **ranks_seen,
                         **({"results_exchange": "libxwb.so" if lib_comm_main is not None else "torch.distributed",
                             "results_exchange_note": lib_note} if (world > 1 or forced) else {})),
        }
        return line

    if rank == 0:
        WATCH["line"] = core_line()
        WATCH["pending"] = [x for x in WATCH["pending"] if x != "main"]

    # ---- a second, shorter measurement in the same run: the fused call xwb_step_autoreset (the reference example loop's
    # `if game_over: reset_game()` inside the step: a finished env's observation is the first frame of its next episode,
    # its terminal frame is not materialised -- NOT the loop `value` is quoted on) ----
    ar_line = None
    if fused == 1 and not args.autoreset:
        loop["autoreset"] = True
        with phase("step_autoreset", PT):
            for _ in range(K):
                one_step()
            ar_regions = [timed_region() for _ in range(3)]
        ar_path = sim.step_path()
        loop["autoreset"] = False
        ar_med = statistics.median(ar_regions)
        ar_line = {"loop": "step_autoreset (terminal frames of finished envs not materialised)" if is_xworld else
                           "step_autoreset (one launch per step: the step kernel resets the envs it finishes)", "regions": 3,
                   "ms_per_step": ar_med / args.steps * 1e3, "value": n_local * world * args.steps / ar_med, "unit": "env-steps/s",
                   "step_loop_frac": n_local * per_step * args.steps / ar_med / 1e9 / HBM_PEAK_GBS, "path": ar_path}
        one_step()                                       # back in the default loop before anything else is measured
        if rank == 0:
            WATCH["line"]["step_autoreset"] = ar_line
    WATCH["pending"] = [x for x in WATCH["pending"] if x != "step_autoreset"]

    # ---- the OTHER path of the default loop (weak point of round 3: "a trainer can end up on a path the bench never timed"):
    # a second batch of the same workload held on the classic kernel sequence (xwb_config.debug_flags no_pregen = what a batch
    # runs after three foreign resets, or with a curriculum / minstd / exclusive groups), timed with the same loop, 3 regions ----
    classic_line = None
    if is_xworld and world == 1 and fused == 1 and not args.autoreset and path_default["path"] == "lazy":
        sim2 = make_sim(args.workload, n_local, local_rank, 0, args.seed, debug=["no_pregen"])

        def classic_region():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                sim2.step()
                sim2.reset_done()
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        for _ in range(max(W, 20)):
            sim2.step()
            sim2.reset_done()
        t_end = time.perf_counter() + min(args.spin_seconds, 0.2)
        while time.perf_counter() < t_end:
            classic_region()
        c_med = statistics.median([classic_region() for _ in range(3)])
        classic_line = {"loop": "step + reset_done on the classic path (terminal snapshots, map generator beside the render, list render)",
                        "regions": 3, "ms_per_step": c_med / args.steps * 1e3, "value": n_local * args.steps / c_med, "unit": "env-steps/s",
                        "step_loop_frac": n_local * per_step * args.steps / c_med / 1e9 / HBM_PEAK_GBS, "path": sim2.step_path()}
        assert sim2.check_errors() == 0
        sim2.close()

    # ---- N > 1: the same loop with the screens of every shard gathered into one tensor on rank 0: the pixels themselves
    # (link-bound), and -- full observation, library exchange -- the cell codes with the root drawing every frame ----
    sg_line = None
    if with_screens:
        lib_comm = lib_comm_main if args.exchange == "lib" else None
        grids_ok = is_xworld and not sim.cfg.visible_radius
        modes = [m for m in (("screens", "grids", "grids_nodraw") if args.gather == "both" else (args.gather,)) if m == "screens" or grids_ok]
        blocks = {}
        for mode in modes:
            try:
                with phase("screens_gather:" + mode, PT):
                    blocks[mode], screens = gather_regions(sim, mode, lib_comm, counts, rank, world, n_local, K, R, args, set_screens, timed_region, fence)
            except Exception as e:                       # a second measurement must not take the line down with it
                blocks[mode] = {"error": "%s: %s" % (type(e).__name__, e)}
            screens = None                               # (the batch keeps the buffer it is bound to alive)
            if rank == 0:                                # (what the watchdog would print if a later mode hangs)
                WATCH["line"].setdefault("screens_gather", {"mode": modes[0]})
                if mode == modes[0]:
                    WATCH["line"]["screens_gather"] = dict(blocks[mode], mode=mode)
                else:
                    WATCH["line"]["screens_gather"][{"grids": "grids", "grids_nodraw": "grids_no_local_render"}.get(mode, mode)] = blocks[mode]
        sg_line = dict(blocks[modes[0]])
        sg_line["mode"] = modes[0]
        if len(modes) > 1:
            sg_line["grids"] = blocks["grids"]
            if "grids_nodraw" in blocks:
                sg_line["grids_no_local_render"] = blocks["grids_nodraw"]
        elif args.gather != "screens" and not grids_ok:
            sg_line["grids"] = {"skipped": "needs a full-observation xworld workload (a frame must be a function of the cell codes)"}
    WATCH["pending"] = [x for x in WATCH["pending"] if x != "screens_gather"]
    errs = sim.check_errors()
    assert errs == 0
    # ---- N > 1: BASELINE C5 (xworld11, 8 x 32 768 envs, RCCL gather of screens) as a block of the same line ----
    c5 = None
    if world > 1 and (args.c5 or world == 8) and args.workload != "xworld11":
        try:
            with phase("c5", 2 * PT if PT else None):
                c5 = c5_block(args, world, rank, local_rank, dev, K, lib_comm_main)
        except Exception as e:                           # the main line must not die with its second measurement
            c5 = {"error": "%s: %s" % (type(e).__name__, e)}
        if rank == 0:
            WATCH["line"]["c5"] = c5
    WATCH["pending"] = [x for x in WATCH["pending"] if x != "c5"]

    if rank == 0:
        line = WATCH["line"]
        line["phase_seconds"] = WATCH.get("phase_seconds", {})
        if ar_line is not None:
            line["step_autoreset"] = ar_line
        if classic_line is not None:
            line["classic_path"] = classic_line
        if sg_line is not None:
            line["screens_gather"] = sg_line
        if c5 is not None:
            line["c5"] = c5
        if world > 1:
            line["multi_gpu_note"] = ("value / screens_gather / c5 are this run's measurements; DESIGN.md carries no 8-GPU number of "
                                      "its own until a SCALE record exists")
        if not args.no_parity:
            # the checker leg: nothing above this line touched the oracle
            line["parity"] = parity_gate(args.workload, rec[0], calls[0], slots, fused, 0, args.seed,
                                         min(args.parity_envs, n_local), probe_calls)
            if frames is not None:
                line["parity"]["frames"] = frames
        if forced:
            line["forced_exchange"] = "world size 1: the N > 1 run's collectives were issued all the same (they move nothing between GPUs)"
        if not args.no_cpu_baseline and world == 1:          # rank 0, N = 1 only
            line["cpu_baseline"] = cpu_baseline(args.workload, args.seed)
        print(json.dumps(line))
        sys.stdout.flush()
    WATCH["printed"] = True
    sim.close()
    if world > 1 or forced:
        with phase("teardown", PT):                          # (the line is out: a hang here only costs the exit)
            WATCH["line"] = None
            dist.barrier()
            dist.destroy_process_group()


if __name__ == "__main__":
    main()

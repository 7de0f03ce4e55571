#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched XWorld simulator on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload xworld7|xworld8|xworld11|xworld7_ego3|...|simple_game|simple_race]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over the whole batch with inputs resident in HBM: SimulatorInterface::take_actions
(act_rep = 1) for every env under the built-in uniform random policy (actions drawn on device), observation of every env
materialised in HBM, then the reference example loop's `if game_over: reset_game()` for the envs that finished.  Default
workload = BASELINE.json config C4 (the configuration the north-star target is quoted on): XWorld2D 7x7, 84x84x3 uint8 planar
BGR, 32 768 envs per GPU.

Timing: W untimed warm-up steps, the same loop spun by wall time (--spin-seconds), unreported K-step regions until three in a
row agree within 1 %, then EXACTLY K steps between two barrier + synchronize fences, max over ranks, R times over (--repeats):
`ms_per_step` / `value` are the MEDIAN region, the spread is reported (`regions`).

Parity gate: every step of the run writes (reward, game_over) of every env into a device-side record
(xwb_bind_results_ring, no extra launches); after the timed regions the record of a slab of envs is compared, bit for bit, with
the CPU oracle's rollout of the same envs from reset; a frame gate checks the first steps' frames.  The oracle is test
infrastructure: bench_lib/parity.py is its only user here, outside every timed region.

N = 1 also times, behind the main measurement, `step_autoreset`, `classic_path` and `secondary` (egocentric mode, SimpleGame,
SimpleRace).  N > 1: the env batch is sharded by global env id (weak scaling); `value` = screens device-resident, one RCCL
all-gather of (reward, game_over) per step; `screens_gather` / `c5` = secondary blocks, each under a watchdog phase
(bench_lib/watchdog.py).  `--dry-run` prints the planned phases and their budgets without touching a GPU.

Prints ONE JSON line (rank 0).  The parts live in bench_lib/ (workloads, plan, dist, loop, parity, watchdog, gather, c5, secondary).
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

from bench_lib import gather, secondary, watchdog                                               # noqa: E402
from bench_lib.c5 import c5_block                                                               # noqa: E402
from bench_lib.dist import init_distributed, library_comm                                       # noqa: E402
from bench_lib.loop import Loop, region_trend                                                   # noqa: E402
from bench_lib.plan import SECONDARY, plan                                                      # noqa: E402
from bench_lib.parity import cpu_baseline, frame_gate, parity_gate                              # noqa: E402
from bench_lib.watchdog import WATCH, phase                                                     # noqa: E402
from bench_lib.workloads import (HBM_PEAK_GBS, REC_BYTES_CAP, WORKLOADS, algorithmic_bytes,     # noqa: E402
                                 dominant_kernel_name, make_sim, measured_traffic, write_ceiling)

def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=9, help="timed K-step regions; the median one is reported")
    ap.add_argument("--spin-seconds", type=float, default=0.3, help="untimed run of the same loop before the timed regions")
    ap.add_argument("--seed", type=lambda v: int(v, 0), default=0xC0FFEE, help="env RNG seed (xwb-rng-v1 key word 0)")
    ap.add_argument("--workload", default="xworld7", choices=list(WORKLOADS))
    ap.add_argument("--envs-per-gpu", type=int, default=0)
    ap.add_argument("--no-screens-gather", action="store_true", help="N > 1: skip the screens-gather-inclusive regions")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="N = 1: skip the `secondary` workloads behind the main measurement")
    ap.add_argument("--parity-envs", type=int, default=4096, help="envs whose whole per-step record is checked against the oracle")
    ap.add_argument("--frame-envs", type=int, default=64, help="envs whose frames are checked against the oracle's renderer ...")
    ap.add_argument("--frame-steps", type=int, default=24, help="... over this many steps from reset (untimed, before the warm-up)")
    ap.add_argument("--autoreset", action="store_true", help="use the fused step+reset+single-render call")
    ap.add_argument("--fused", type=int, default=1, help="simple games only: steps per launch (xwb_step_n); --steps must be a multiple")
    ap.add_argument("--exchange", default="auto", choices=["auto", "torch", "lib"],
                    help="N > 1: who issues the exchanges.  torch: torch.distributed for everything.  lib: libxwb.so's own RCCL calls "
                         "for everything (backend nccl only).  auto: the per-step results through the library when its communicator "
                         "comes up on every rank within 90 s (else torch), the screens gathers through torch.distributed")
    ap.add_argument("--gather", default="both", choices=["screens", "grids", "grids_nodraw", "both"],
                    help="N > 1, full observation: what crosses the links per step -- every shard's pixels (screens), or its cell "
                         "codes with the root drawing all frames (grids), or one set of regions each (both)")
    ap.add_argument("--c5", action="store_true", help="N > 1: add the BASELINE C5 block (xworld11); on by itself at N = 8")
    ap.add_argument("--force-exchange", action="store_true",
                    help="N = 1: initialise torch.distributed (world size 1) and issue the N > 1 run's exchanges all the same")
    ap.add_argument("--results-wait", action="store_true",
                    help="N > 1: order the step's stream behind every step's all-gather of (reward, done) instead of releasing it")
    ap.add_argument("--phase-timeout", type=float, default=240.0,
                    help="N > 1: seconds a secondary measurement may take before the watchdog prints the line without it; main gets 3x")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo exercises the code path)")
    ap.add_argument("--dry-run", action="store_true", help="print the planned phases and their watchdog budgets; touches no GPU")
    return ap.parse_args()


def main():
    args = parse_args()
    # (a dry run without a launcher plans for --gpus ranks; a real run is as wide as its launcher made it)
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus) if args.dry_run else "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "workload": args.workload, "phases": plan(args, world)}))
        return
    import torch
    import torch.distributed as dist
    from xworld_amd import sharding
    local_rank = init_distributed(args, world, local_rank)
    dev = torch.device("cuda", local_rank)
    forced = world == 1 and args.force_exchange
    guarded = world > 1 or forced
    PT = args.phase_timeout if guarded else None
    watchdog.watch_start(rank)
    WATCH["pending"] = [p["phase"] for p in plan(args, world)]
    watchdog.arm("main (set-up, frame gate, warm-up)", 3 * PT if PT else None)
    n_local = args.envs_per_gpu or WORKLOADS[args.workload][2]
    sim = make_sim(args.workload, n_local, local_rank, rank * n_local, args.seed)
    per_step, per_launch, kernel_name = algorithmic_bytes(args.workload, sim)
    is_xworld = WORKLOADS[args.workload][0] == "xworld"
    fused = args.fused if not is_xworld else 1
    assert fused == 1 or args.steps % fused == 0, "--steps must be a multiple of --fused"
    K, W, R = args.steps // fused, -(-args.warmup // fused), max(1, args.repeats)   # in step CALLS (warm-up rounded up)
    counts = [n_local] * world
    lib_comm_main, lib_note = None, None
    if guarded and args.exchange in ("lib", "auto") and args.backend == "nccl":
        lib_comm_main, lib_note = library_comm(args, world, rank, local_rank, dev)
    if lib_comm_main is not None:
        results = sharding.LibResultGather(sim, lib_comm_main, counts, rank)
    else:
        results = sharding.ResultGather(counts, rank, dev, force_collective=forced) if guarded else None
    with_screens = guarded and not args.no_screens_gather
    L = Loop(args, sim, results, world, forced, dev, fused)

    # ---- plan the run: the record ring must exist before the first step; its length needs the step count, which needs the spin
    # length, which needs a step time: a short untimed probe on a throw-away ring gives it ----
    L.rec = torch.zeros((2, n_local, 2), dtype=torch.float32, device=dev)
    sim.bind_results_ring(L.rec)
    frames = None
    if not args.no_parity and fused == 1 and not (is_xworld and sim.obs_is_float):      # (float32 frames: the oracle renders uint8)
        frames = frame_gate(args.workload, sim, L.one_step, L.fence, min(args.frame_envs, n_local), args.frame_steps, args.seed, rank)
    for _ in range(3):
        L.one_step()
    L.fence()
    t0 = time.perf_counter()
    for _ in range(10):
        L.one_step()
    L.fence()
    est = (time.perf_counter() - t0) / 10
    spin_calls = L.bcast_int(min(200000, math.ceil(max(0.0, args.spin_seconds) / max(est, 1e-7))))
    total_calls = 13 + W + spin_calls + 24 * K + 2 * R * K + (R * K + 2 * K if with_screens else 0) + 4 * K
    slots = int(max(2, min(total_calls, REC_BYTES_CAP // (n_local * 8))))
    L.rec = torch.zeros((slots, n_local, 2), dtype=torch.float32, device=dev)
    sim.bind_results_ring(L.rec)                         # (bind resets the library's slot counter: restart ours with it)
    probe_calls, L.calls = L.calls, 0
    # the write-ceiling anchor's buffer: allocated before the warm-up (nothing is allocated between the spin and the last region)
    ceiling_buf = torch.empty(int(n_local * sim.obs_bytes_per_env), dtype=torch.uint8, device=dev)
    for _ in range(W):
        L.one_step()
    # clocks warm, caches and allocator settled: spin by WALL TIME, every rank the same number of steps ...
    spun = 0
    t_end = time.perf_counter() + max(0.0, args.spin_seconds)
    while spun < 400000:
        for _ in range(50):
            L.one_step()
        spun += 50
        L.fence() if world > 1 else torch.cuda.synchronize()
        if not L.bcast_int(1 if time.perf_counter() < t_end else 0):
            break
    # ... then unreported K-step regions until three in a row agree within 1 % (at most 24)
    watchdog.arm("main", 3 * PT if PT else None)
    settle = []
    while len(settle) < 24:
        settle.append(L.timed_region())
        if L.bcast_int(1 if len(settle) >= 3 and max(settle[-3:]) <= 1.01 * min(settle[-3:]) else 0):
            break
    # ---- the timed regions: exactly K steps each between two barrier + synchronize fences, max over ranks ----
    del L.host_issue[:]
    regions = [L.timed_region() for _ in range(R)]
    dt_med = statistics.median(regions)
    host_us_per_step = statistics.median(L.host_issue) / args.steps * 1e6
    # ---- R more regions with hipEvents around every launch of the dominant kernel (on its launch stream, recorded inside libxwb) ----
    sim.profile_begin()
    ev_regions = [L.timed_region() for _ in range(R)]
    kern_us, kern_n = sim.profile_end("render" if is_xworld else "step")
    kernels_us = {}
    for kname in (("step", "render", "reset", "list") if is_xworld else ("step", "reset")):
        us, nl = sim.profile_end(kname)                  # (each on the stream it runs on; the map generator runs BESIDE the render)
        if nl:
            kernels_us[kname] = {"avg_us": us, "launches": nl}
    sim.profile_stop()
    path_default = sim.step_path()
    ceiling = write_ceiling(ceiling_buf)                 # the run's own bandwidth anchor, behind everything the headline is read from
    for _ in range(K):                                   # and the loop itself warm again before the secondary measurements
        L.one_step()
    watchdog.disarm()
    ranks_seen = {}
    if guarded:                                          # N ranks on N devices, in the record itself
        mine = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.get_device_name(dev), "index": torch.cuda.current_device(),
                "uuid": str(getattr(torch.cuda.get_device_properties(dev), "uuid", ""))}
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        ranks_seen = {"ranks": everyone, "ranks_seen": len({r["rank"] for r in everyone}),
                      "devices_seen": len({r["uuid"] or r["index"] for r in everyone})}

    def core_line():
        total_envs = n_local * world
        achieved = n_local * per_launch * fused / (kern_us * 1e-6) / 1e9 if kern_us > 0 else 0.0
        traffic, traffic_info = (measured_traffic(args.workload) if n_local == WORKLOADS[args.workload][2]
                                 else (None, {"traffic_source": None}))
        if not guarded:
            exchange = "none"
        else:
            who = ("libxwb.so on its communicator's stream, ordered by the step's %s" % ("epoch" if results.by_epoch else "event")
                   if lib_comm_main is not None else "torch.distributed")
            exchange = "all_gather(reward,done) per step (%s; %s), screens device-resident" % (
                "the step's stream waits for it one step late" if args.results_wait else "released: no reader on the step's stream", who)
        loop_name = (("step_n(%d): %d steps per launch, auto-reset" % (fused, fused)) if fused > 1 else
                     ("step_autoreset" if args.autoreset else "step + reset_done"))
        step_loop = total_envs * per_step * args.steps / dt_med / 1e9
        return {
            "metric": "env-steps/sec (batched random policy)", "value": total_envs * args.steps / dt_med, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_med / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (f64 trig)" if WORKLOADS[args.workload][0] == "simple_race" else
                      ("u8 state, f32 frames (pixel * 1/255)" if sim.obs_is_float else "u8")),
            "data": "synthetic",
            "config": {"workload": args.workload, "envs_per_gpu": n_local, "total_envs": total_envs, "obs": list(sim.obs.shape[1:]),
                       "seed": args.seed, "policy": "uniform random, drawn on device", "loop": loop_name, "exchange": exchange,
                       "parallelism": "env-sharded x%d" % world},
            "regions": {"repetitions": R, "statistic": "median", "steps_per_region": args.steps,
                        "ms_per_step_min": min(regions) / args.steps * 1e3, "ms_per_step_max": max(regions) / args.steps * 1e3,
                        "ms_per_step_all": [r / args.steps * 1e3 for r in regions], "trend": region_trend(regions),
                        "untimed_before": {"warmup_steps": args.warmup, "spin_steps": spun * fused, "spin_seconds_target": args.spin_seconds,
                                           "probe_steps": probe_calls * fused, "settle_regions": len(settle)}},
            "path": path_default,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": dominant_kernel_name(kernel_name, path_default["path"]),
                         "write_ceiling_GBps": ceiling["memset"]["GBps"], "write_ceiling": ceiling,
                         "frac_of_write_ceiling": achieved / ceiling["memset"]["GBps"] if ceiling["memset"]["GBps"] else None,
                         "kernels_us": kernels_us, "kernel_avg_us": kern_us, "kernel_launches": kern_n,
                         "algorithmic_bytes_per_launch": n_local * per_launch * fused, "algorithmic_bytes_per_env_step": per_step,
                         "step_loop_GBps": step_loop, "step_loop_frac": step_loop / HBM_PEAK_GBS / world, **traffic_info},
            "timed_with_events_ms_per_step": statistics.median(ev_regions) / args.steps * 1e3,
            "host_us_per_step": host_us_per_step,
            "rccl": dict(sharding.backend_info(), **ranks_seen,
                         **({"results_exchange": "libxwb.so" if lib_comm_main is not None else "torch.distributed",
                             "results_exchange_note": lib_note} if guarded else {})),
        }

    line = core_line() if rank == 0 else None
    WATCH["line"] = line
    watchdog.done("main")

    # ---- xwb_step_autoreset: a finished env's observation is the first frame of its next episode -- NOT the loop `value` is quoted on
    if fused == 1 and not args.autoreset:
        L.autoreset = True
        with phase("step_autoreset", PT):
            for _ in range(K):
                L.one_step()
            ar_regions = [L.timed_region() for _ in range(3)]
        ar = secondary.autoreset_line(sim, is_xworld, ar_regions, args.steps, n_local * world, n_local, per_step)
        L.autoreset = False
        L.one_step()                                     # back in the default loop before anything else is measured
        if rank == 0:
            line["step_autoreset"] = ar
    watchdog.done("step_autoreset")
    if is_xworld and world == 1 and fused == 1 and not args.autoreset and path_default["path"] in ("lazy", "lazy_fused"):
        line["classic_path"] = secondary.classic_line(args.workload, n_local, local_rank, args.seed, K, W, args.steps,
                                                      args.spin_seconds, per_step)
    watchdog.done("classic_path")
    # ---- N = 1, default workload: the other workloads under the same clock ----
    if any(p.startswith("secondary:") for p in WATCH["pending"]):
        line["secondary"] = {}
        for w in SECONDARY:
            try:
                with phase("secondary:" + w, None):
                    line["secondary"][w] = secondary.secondary_block(w, local_rank, args.seed, K, args.steps)
            except Exception as e:                       # noqa: BLE001 -- a block, not the line
                line["secondary"][w] = {"error": "%s: %s" % (type(e).__name__, e)}
            watchdog.done("secondary:" + w)

    # ---- N > 1: the same loop with the screens of every shard gathered into one tensor on rank 0 ----
    if with_screens:
        lib_comm = lib_comm_main if args.exchange == "lib" else None
        grids_ok = is_xworld and not sim.cfg.visible_radius
        modes = gather.modes_for(args.gather, grids_ok)
        blocks = {}
        for mode in modes:
            try:
                with phase("screens_gather:" + mode, PT):
                    blocks[mode] = gather.gather_regions(sim, mode, lib_comm, counts, rank, world, n_local, K, R, args.steps,
                                                         L.set_screens, L.one_step, L.timed_region, L.fence)
            except Exception as e:                       # noqa: BLE001 -- a second measurement must not take the line down with it
                blocks[mode] = {"error": "%s: %s" % (type(e).__name__, e)}
            L.set_screens(None)                          # (the batch keeps the buffer it is bound to alive)
            watchdog.done("screens_gather:" + mode)
            if rank == 0:                                # (what the watchdog would print if a later mode hangs)
                line["screens_gather"] = gather.merge_blocks(blocks, modes[:len(blocks)])
        if rank == 0 and len(modes) == 1 and args.gather != "screens" and not grids_ok:
            line["screens_gather"]["grids"] = {"skipped": "needs a full-observation xworld workload"}
    assert sim.check_errors() == 0
    # ---- N > 1: BASELINE C5 (xworld11, 8 x 32 768 envs, RCCL gather of screens) as a block of the same line ----
    if "c5" in WATCH["pending"]:
        try:
            with phase("c5", 2 * PT if PT else None):
                c5 = c5_block(args, world, rank, local_rank, dev, K, lib_comm_main)
        except Exception as e:                           # noqa: BLE001
            c5 = {"error": "%s: %s" % (type(e).__name__, e)}
        if rank == 0:
            line["c5"] = c5
        watchdog.done("c5")

    if rank == 0:
        line["phase_seconds"] = WATCH.get("phase_seconds", {})
        if world > 1:
            line["multi_gpu_note"] = ("value / screens_gather / c5 are this run's measurements; DESIGN.md carries no 8-GPU number of "
                                      "its own until a SCALE record exists")
        if not args.no_parity:                           # the checker leg: nothing above this line touched the oracle
            line["parity"] = parity_gate(args.workload, L.rec, L.calls, slots, fused, 0, args.seed, min(args.parity_envs, n_local), probe_calls)
            if frames is not None:
                line["parity"]["frames"] = frames
        if forced:
            line["forced_exchange"] = "world size 1: the N > 1 run's collectives were issued all the same (they move nothing between GPUs)"
        if not args.no_cpu_baseline and world == 1:      # rank 0, N = 1 only
            line["cpu_baseline"] = cpu_baseline(args.workload, args.seed)
        print(json.dumps(line))
        sys.stdout.flush()
    WATCH["printed"] = True
    sim.close()
    if guarded:
        with phase("teardown", PT):                      # (the line is out: a hang here only costs the exit)
            WATCH["line"] = None
            dist.barrier()
            dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched XWorld simulator on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload xworld7|xworld7_f32|xworld7_ego3|xworld8|xworld11|simple_game|simple_race]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over the whole batch with inputs resident in HBM:
SimulatorInterface::take_actions(act_rep=1) for every env under the built-in uniform random policy
(actions drawn on device), observation of every env materialised in HBM, then the reference example
loop's `if game_over: reset_game()` for the envs that finished (wavefront-ballot compaction, map
generation, re-render).  Default workload = BASELINE.json config C4 (the configuration the north-star
target is quoted on): XWorld2D 7x7, 84x84x3 uint8 planar BGR, 32 768 envs per GPU.

Multi-GPU: the env batch is sharded by global env id (weak scaling: 32 768 envs per GPU); every step
every rank's (reward, game_over) is exchanged through one RCCL all-gather (rank 0 consumes it).  `--gather-screens` also
gathers every shard's screens into one contiguous tensor on rank 0 (xGMI-link bound).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)

WORKLOADS = {
    # name: (game, opts, envs per GPU, algorithmic bytes per env-step, bytes per env per render launch)
    "xworld7": ("xworld", {"max_dim": 7, "num_blocks": 16, "color": True}, 32768),
    "xworld7_f32": ("xworld", {"max_dim": 7, "num_blocks": 16, "color": True, "obs_format": "float32"}, 32768),
    "xworld7_ego3": ("xworld", {"max_dim": 7, "num_blocks": 16, "color": True, "visible_radius": 3}, 32768),
    "xworld8": ("xworld", {"color": True}, 32768),
    "xworld11": ("xworld", {"max_dim": 11, "num_blocks": 30, "color": True}, 32768),
    "simple_game": ("simple_game", {"array_size": 64}, 65536),
    "simple_race": ("simple_race", {"track_width": 20.0, "track_length": 100.0, "track_radius": 30.0}, 65536),
}


def make_sim(workload, n_envs, device, gid0, seed=0xC0FFEE):
    from xworld_amd.batched import BatchedSimulator
    game, opts, _ = WORKLOADS[workload]
    opts = dict(opts)
    if game == "xworld":
        opts["xwd_conf_path"] = os.path.join(ROOT, "xworld_amd", "confs", "navigation2d.json")       # the five XWorld3DNav tasks
        opts["task_mode"] = "lang_acquisition"
    return BatchedSimulator(game, opts, num_envs=n_envs, device=device, env_gid0=gid0,
                            seed=seed, policy_seed=0x5EED)


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
        # egocentric: the frame is (r * (84 / r))^2 pixels; the render is compute-bound (16 view-pixel evaluations per
        # output pixel), its algorithmic bytes are the frame written + the grid read
        obs = c * sim.screen_dims[0] * sim.screen_dims[1]
        return 33 + 2 * d * d + obs, 2 * d * d + obs, "xw_render_ego_kernel"
    obs = c * 144 * d * d * (4 if sim.obs_is_float else 1)      # float32 variant: obs term x 4
    return 33 + 2 * d * d + obs, 2 * d * d + obs, "xw_render_all_kernel"


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in separate runs, gfx950 corrections applied by tools/summarize_prof.py); None if not profiled."""
    best = None
    for d in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        f = os.path.join(ROOT, "profiles", d, "traffic_%s.json" % workload)
        if os.path.exists(f):
            best = f
    if not best:
        return None, None
    with open(best) as fh:
        t = json.load(fh)
    return t["traffic_bytes_per_launch"], os.path.relpath(best, ROOT)


def cpu_baseline(workload, seconds_target=8.0):
    """The oracle (CPU restatement of the reference path, kind = "port") timed on this box's host cores on a bounded
    sample of the same workload (same loop: game_over? -> reset; get_state; random action; take_actions incl. screen):
    first one thread (calibration, also reported), then one independent env batch per thread on every core (ctypes
    releases the GIL; the oracle keeps no global state) -- `value` / `cores` are the all-core figures."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    game = WORKLOADS[workload][0]
    sim_opts = WORKLOADS[workload][1]
    pal = O.Palette(O.NAV_SUBTREES) if game == "xworld" else None

    def rollout(n, steps, gid0):
        if game == "simple_game":
            O.sg_rollout(n, 64, steps, 0x5EED, env_gid0=gid0)
        elif game == "simple_race":
            O.race_rollout(n, O.race_cfg(), 0xC0FFEE, steps, 0x5EED, env_gid0=gid0)
        else:
            d = sim_opts.get("max_dim", 8)
            cfg = O.xw_cfg(map_kind=0, max_dim=d, dim=d, num_goals=4, num_blocks=sim_opts.get("num_blocks", 16),
                           color=1, seed=0xC0FFEE, tasks=[0, 1, 2, 3, 4], visible_radius=sim_opts.get("visible_radius", 0))
            O.xw_rollout(n, cfg, pal, steps, 0x5EED, env_gid0=gid0, render=True)

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--seed", type=lambda v: int(v, 0), default=0xC0FFEE, help="env RNG seed (xwb-rng-v1 key word 0)")
    ap.add_argument("--workload", default="xworld7", choices=list(WORKLOADS))
    ap.add_argument("--envs-per-gpu", type=int, default=0)
    ap.add_argument("--gather-screens", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--autoreset", action="store_true", help="use the fused step+reset+single-render call")
    ap.add_argument("--fused", type=int, default=1, help="simple games only: steps per launch (xwb_step_n); --steps must be "
                    "a multiple; every step still writes its reward / code / observation")
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
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(0)
    n_gpus = world
    dev = torch.device("cuda", local_rank)
    n_local = args.envs_per_gpu or WORKLOADS[args.workload][2]
    sim = make_sim(args.workload, n_local, local_rank, rank * n_local, args.seed)
    per_step, per_launch, kernel_name = algorithmic_bytes(args.workload, sim)

    # per-step exchange (xworld_amd/sharding.py): (reward, game_over) of every shard to rank 0 through one
    # RCCL gather; with --gather-screens also every shard's screens into one contiguous tensor on rank 0
    from xworld_amd import sharding
    counts = [n_local] * world
    results = sharding.ResultGather(counts, rank, dev) if world > 1 else None
    screens_all = None
    if world > 1 and args.gather_screens and rank == 0:
        screens_all = torch.empty((world * n_local,) + tuple(sim.obs.shape[1:]), dtype=sim.obs.dtype, device=dev)
        sim.bind_obs(screens_all[:n_local])              # rank 0 renders straight into its slice

    def exchange_results():
        if world == 1:
            return
        # finish the gather of the previous step (it ran beside this step's kernels), start this step's: the step
        # kernel wrote (reward, code) straight into the buffer, no packing kernels
        results.finish()
        results.start()

    def exchange_screens():
        if world > 1 and args.gather_screens:
            sharding.gather_slabs(sim.obs, screens_all, counts, rank)

    fused = args.fused if WORKLOADS[args.workload][0] != "xworld" else 1
    assert args.steps % fused == 0 and args.warmup % fused == 0 or fused == 1, "--steps / --warmup must be multiples of --fused"

    def one_step():
        if fused > 1:                                    # `fused` steps in one launch (built-in policy, auto-reset)
            sim.step_n(fused)
            return
        if results is not None:
            sim.bind_results(results.next_buffer())
        if args.autoreset:
            sim.step_autoreset()
            exchange_results()
        else:
            sim.step()
            exchange_results()                           # this step's results, before reset_done clears the codes
            sim.reset_done()
        exchange_screens()                               # the frames the next policy step would see

    def fence():
        if results is not None:
            results.finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup // fused):
        one_step()
    # ---- the timed region: exactly K steps between two barrier + synchronize fences ----
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps // fused):
        one_step()
    fence()
    dt_clean = time.perf_counter() - t0
    dt_max = dt_clean
    if world > 1:
        tt = torch.tensor([dt_clean], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_max = float(tt.item())

    # ---- same K steps again with hipEvents around every launch of the dominant kernel (on its
    # launch stream, recorded inside libxwb) to get that kernel's average duration for the roofline ----
    sim.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps // fused):
        one_step()
    fence()
    dt = time.perf_counter() - t0
    kern = "render" if WORKLOADS[args.workload][0] == "xworld" else "step"
    kern_us, kern_n = sim.profile_end(kern)
    sim.profile_stop()
    errs = sim.check_errors()
    assert errs == 0

    if rank == 0:
        total_envs = n_local * world
        value = total_envs * args.steps / dt_max
        # algorithmic bytes of one launch = per-step bytes x the steps that launch runs
        achieved = n_local * per_launch * fused / (kern_us * 1e-6) / 1e9 if kern_us > 0 else 0.0
        traffic, traffic_src = measured_traffic(args.workload) if n_local == WORKLOADS[args.workload][2] else (None, None)
        line = {
            "metric": "env-steps/sec (batched random policy)",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
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
                       "exchange": ("all_gather(reward,done)" + ("+gather(screens)" if args.gather_screens else ""))
                       if world > 1 else "none", "parallelism": "env-sharded x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_name,
                         "kernel_avg_us": kern_us, "kernel_launches": kern_n,
                         "algorithmic_bytes_per_launch": n_local * per_launch * fused,
                         "algorithmic_bytes_per_env_step": per_step,
                         "step_loop_GBps": total_envs * per_step * args.steps / dt_max / 1e9},
            "timed_with_events_ms_per_step": dt / args.steps * 1e3,
        }
        if not args.no_cpu_baseline and world == 1:          # rank 0, N = 1 only
            line["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(line))
    sim.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

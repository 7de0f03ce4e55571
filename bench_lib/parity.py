"""The checker legs of bench.py: oracle rollouts (test infrastructure: used here, outside every timed region, only), the parity
gate over the device-side record, the frame gate and the cpu_baseline timing."""
import os
import sys
import threading
import time

from .workloads import POLICY_SEED, ROOT, WORKLOADS


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
    return {"checked_env_steps": (calls - first) * slab, "mismatches": mism, "envs": slab,
            "scope": "a slab of envs, every recorded step: reward bits + game_over code (not the frames, not the whole batch)",
            "step_calls": [calls_before + first, calls_before + calls],
            "against": "oracle/liboracle.so rollout from reset, reward bits + game_over code"}


def frame_gate(workload, sim, one_step, fence, envs, steps, seed, rank):
    """The frame every policy step sees, for a slab of rank 0's envs over the first `steps` steps from reset, against the
    oracle's own renderer (position-weighted checksums, oracle/oracle.h orc_obs_checksum).  Untimed, before anything else runs."""
    import numpy as np
    import torch
    got = []
    for _ in range(steps):
        fence()
        got.append(sim.obs[:envs].contiguous().view(torch.uint8).reshape(envs, -1).cpu().numpy() if rank == 0 else None)
        one_step()
    fence()
    if rank != 0:
        return None
    ref = oracle_rollout(workload, envs, steps, 0, seed, render=True)
    O = _oracle()
    bad = sum(int(np.count_nonzero(O.obs_checksum_np(got[t]) != ref.obs_ck[t])) for t in range(steps))
    return {"checked_frames": envs * steps, "mismatches": bad, "envs": envs, "steps": steps,
            "against": "oracle/liboracle.so renderer (64 px canvas + cv::resize restatement), checksum of every byte of the frame"}


def cpu_baseline(workload, seed, seconds_target=8.0):
    """The oracle (CPU restatement of the reference path, kind = "port") timed on this box's host cores on a bounded
    sample of the same workload (same loop: game_over? -> reset; get_state; random action; take_actions incl. screen):
    first one thread (calibration, also reported), then one independent env batch per thread on every core (ctypes
    releases the GIL; the oracle keeps no global state) -- `value` / `cores` are the all-core figures."""
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

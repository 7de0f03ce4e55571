"""Secondary measurements of the default bench line, each a few K-step regions behind the main one: the fused call
xwb_step_autoreset, the classic kernel sequence of the default loop, and -- N = 1 -- the other workloads under the same clock
(`secondary`: the egocentric mode and the two simple games; reference loops: python/examples/test_xworld.py:41-60,
test_simple_game.py:15-30), so that the one line the driver records carries more than the headline configuration."""
import statistics
import time

from .parity import oracle_rollout
from .workloads import HBM_PEAK_GBS, WORKLOADS, algorithmic_bytes, dominant_kernel_name, make_sim


def autoreset_line(sim, is_xworld, regions, steps, n_total, n_local, per_step):
    med = statistics.median(regions)
    return {"loop": "step_autoreset (terminal frames of finished envs not materialised)" if is_xworld else
                    "step_autoreset (one launch per step: the step kernel resets the envs it finishes)", "regions": len(regions),
            "ms_per_step": med / steps * 1e3, "value": n_total * steps / med, "unit": "env-steps/s",
            "step_loop_frac": n_local * per_step * steps / med / 1e9 / HBM_PEAK_GBS, "path": sim.step_path()}


def classic_line(workload, n_local, local_rank, seed, K, W, steps, spin_seconds, per_step):
    """The OTHER path of the default loop ("a trainer can end up on a path the bench never timed"): a second batch of the same
    workload held on the classic kernel sequence (debug switch no_pregen = what a batch runs after three foreign resets, or with
    a curriculum / minstd / exclusive groups), timed with the same loop, 3 regions."""
    import torch
    sim2 = make_sim(workload, n_local, local_rank, 0, seed, debug=["no_pregen"])

    def region():
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
    t_end = time.perf_counter() + min(spin_seconds, 0.2)
    while time.perf_counter() < t_end:
        region()
    med = statistics.median([region() for _ in range(3)])
    out = {"loop": "step + reset_done on the classic path (terminal snapshots, map generator beside the render, list render)",
           "regions": 3, "ms_per_step": med / steps * 1e3, "value": n_local * steps / med, "unit": "env-steps/s",
           "step_loop_frac": n_local * per_step * steps / med / 1e9 / HBM_PEAK_GBS, "path": sim2.step_path()}
    assert sim2.check_errors() == 0
    sim2.close()
    return out


def secondary_block(workload, local_rank, seed, K, steps, spin_seconds=0.15, parity_envs=256, parity_steps=16):
    """One more workload in the same process: the default loop (step + reset_done, built-in policy), a short parity gate on the
    first steps from reset, 3 timed regions, the dominant kernel's event time, the host's issue time; the simple games also as
    one launch per step (step_autoreset)."""
    import numpy as np
    import torch
    game, _, n = WORKLOADS[workload]
    sim = make_sim(workload, n, local_rank, 0, seed)
    per_step, per_launch, kernel = algorithmic_bytes(workload, sim)
    ring = torch.zeros((parity_steps, n, 2), dtype=torch.float32, device=sim.obs.device)
    sim.bind_results_ring(ring)

    def one_step():
        sim.step()
        sim.reset_done()
    for _ in range(parity_steps):
        one_step()
    torch.cuda.synchronize()
    got = ring[:, :parity_envs, :].cpu().numpy()
    sim.bind_results(None)
    for _ in range(30):
        one_step()
    host = []

    def region(fn=one_step):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        host.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    t_end = time.perf_counter() + spin_seconds
    while time.perf_counter() < t_end:
        region()
    settle = []
    while len(settle) < 8:
        settle.append(region())
        if len(settle) >= 2 and max(settle[-2:]) <= 1.015 * min(settle[-2:]):
            break
    del host[:]
    regions = [region() for _ in range(3)]
    med = statistics.median(regions)
    host_us = statistics.median(host) / steps * 1e6
    sim.profile_begin()
    region()
    kern = "render" if game == "xworld" else "step"
    kern_us, kern_n = sim.profile_end(kern)
    sim.profile_stop()
    path = sim.step_path()
    achieved = n * per_launch / (kern_us * 1e-6) / 1e9 if kern_us > 0 else 0.0
    out = {"workload": workload, "envs": n, "loop": "step + reset_done", "regions": 3, "steps_per_region": steps,
           "value": n * steps / med, "unit": "env-steps/s", "ms_per_step": med / steps * 1e3,
           "ms_per_step_min_max": [min(regions) / steps * 1e3, max(regions) / steps * 1e3],
           "host_us_per_step": host_us, "path": path,
           "roofline": {"bound": "hbm", "kernel": dominant_kernel_name(kernel, path["path"]), "kernel_avg_us": kern_us,
                        "kernel_launches": kern_n, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "step_loop_frac": n * per_step * steps / med / 1e9 / HBM_PEAK_GBS}}
    if game != "xworld":
        # the same two launches per step issued from C (xwb_run: no per-call binding cost), and one launch per step
        def run_region():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sim.run(K)
            host_run = time.perf_counter() - t0
            torch.cuda.synchronize()
            return time.perf_counter() - t0, host_run
        rr = [run_region() for _ in range(4)][1:]
        rmed = statistics.median([r[0] for r in rr])
        out["xwb_run"] = {"loop": "xwb_run(K): K x (step; reset_done) issued from C", "value": n * steps / rmed, "ms_per_step": rmed / steps * 1e3,
                          "host_us_per_step": statistics.median([r[1] for r in rr]) / steps * 1e6}
        ar = statistics.median([region(sim.step_autoreset) for _ in range(4)][1:])
        out["one_launch"] = {"loop": "step_autoreset (one launch per step)", "value": n * steps / ar, "ms_per_step": ar / steps * 1e3}
    assert sim.check_errors() == 0
    sim.close()
    # the checker leg, behind the timed regions: the first steps' (reward, code) of a slab of envs against the oracle
    ref = oracle_rollout(workload, parity_envs, parity_steps, 0, seed, render=False)
    mism = 0
    for t in range(parity_steps):
        mism += int(np.count_nonzero(got[t][:, 0].view(np.uint32) != ref.rewards[t].view(np.uint32)))
        mism += int(np.count_nonzero(got[t][:, 1].astype(np.uint8) != ref.codes[t]))
    out["parity"] = {"checked_env_steps": parity_envs * parity_steps, "mismatches": mism,
                     "against": "oracle/liboracle.so rollout from reset, reward bits + game_over code"}
    return out

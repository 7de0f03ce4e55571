"""BASELINE.json config C5 -- XWorld2D 11x11, 32 768 envs per GPU (262 144 over 8), "sharded 8 x MI355X with RCCL gather of
screens" -- as a sub-object of the N > 1 line: the same loop as the main measurement on the xworld11 workload, once with the
screens left device-resident (value) and once per gather mode with every shard's screens on rank 0, beside the xGMI link
ceiling.  Fewer regions than the main line (3): it is a second measurement."""
import statistics
import time

from . import gather
from .workloads import WORKLOADS, make_sim


def c5_block(args, world, rank, local_rank, dev, K, lib_comm_main=None):
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
    results = (sharding.LibResultGather(sim, lib_comm_main, counts, rank) if lib_comm_main is not None
               else sharding.ResultGather(counts, rank, dev))
    state = {"screens": None, "calls": 0}

    def one_step():
        if state["screens"] is not None:
            state["screens"].bind_next()
        state["calls"] += 1
        sim.step()
        sim.reset_done()
        (results.finish(convert=False) if args.results_wait else results.release())
        results.start() if lib_comm_main is not None else results.start(packed=packed[(state["calls"] - 1) % 8])
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
    modes = gather.modes_for(args.gather, True)
    blocks = {}
    for mode in modes:
        try:
            g = gather.make_gather(sim, mode, lib_comm, counts, rank)
            state["screens"] = g
            for _ in range(4):
                one_step()
            regs = [region() for _ in range(3)]
            fence()
            blocks[mode] = gather.gather_block(sim, mode, lib_comm, g.depth, world, n_local, K, regs)
        except Exception as e:                           # noqa: BLE001 -- a block, not the line
            blocks[mode] = {"error": "%s: %s" % (type(e).__name__, e)}
        state["screens"] = None
    errs = sim.check_errors()
    d_med = statistics.median(dev_regions)
    out = {"workload": "xworld11",
           "config": "BASELINE C5: 11x11, 132x132x3 u8, %d envs per GPU, %d in all" % (n_local, n_local * world),
           "value": n_local * world * K / d_med, "unit": "env-steps/s", "ms_per_step": d_med / K * 1e3,
           "exchange": "all_gather(reward,done) per step (%s), screens device-resident" %
                       ("libxwb.so, beside the step loop" if lib_comm_main is not None else "torch.distributed"),
           "regions": 3, "steps_per_region": K, "path": sim.step_path(), "action_errors": errs,
           "screens_gather": gather.merge_blocks(blocks, modes)}
    sim.close()
    return out

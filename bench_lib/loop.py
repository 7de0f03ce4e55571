"""The step loop bench.py times: one_step() = the reference example loop's body for the whole batch, fences, timed regions."""
import statistics
import time


class Loop:
    """the step loop of the main measurement and everything that times it"""

    def __init__(self, args, sim, results, world, forced, dev, fused):
        self.args, self.sim, self.results, self.world, self.forced, self.dev, self.fused = args, sim, results, world, forced, dev, fused
        self.screens, self.autoreset, self.calls, self.rec, self.host_issue = None, args.autoreset, 0, None, []
        self.K = args.steps // fused

    def exchange_results(self):
        # finish the gather of the previous step (it ran beside this step's kernels), start this step's: the step kernel wrote
        # (reward, code) straight into the record's slot, no packing kernels
        if self.results is None:
            return
        (self.results.finish(convert=False) if self.args.results_wait else self.results.release())
        if hasattr(self.results, "ring_slots"):            # LibResultGather: the library knows which row of the ring the step wrote
            self.results.start()
        else:
            self.results.start(packed=self.rec[(self.calls - 1) % self.rec.shape[0]])

    def one_step(self):
        if self.screens is not None:
            self.screens.bind_next()
        self.calls += 1
        if self.fused > 1:                               # `fused` steps in one launch (built-in policy, auto-reset)
            self.sim.step_n(self.fused)
            return
        if self.autoreset:
            self.sim.step_autoreset()
        else:
            self.sim.step()
            self.sim.reset_done()
        # this step's results: the rows the step wrote into the record, which reset_done leaves alone.  Behind reset_done, whose
        # list render publishes the epoch of a fused step + render launch: the exchange waits for it on its own stream
        self.exchange_results()
        if self.screens is not None:
            self.screens.start()                         # the frames the next policy step would see

    def fence(self):
        import torch
        import torch.distributed as dist
        if self.results is not None:
            self.results.drain()
        if self.screens is not None:
            self.screens.drain()
        torch.cuda.synchronize()
        if self.world > 1 or self.forced:
            dist.barrier()
        torch.cuda.synchronize()

    def set_screens(self, g):
        self.screens = g

    def bcast_int(self, v):
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return int(v)
        t = torch.tensor([int(v)], dtype=torch.int64, device=self.dev)
        dist.broadcast(t, 0)
        return int(t.item())

    def timed_region(self):
        import torch
        import torch.distributed as dist
        self.fence()
        t0 = time.perf_counter()
        for _ in range(self.K):
            self.one_step()
        self.host_issue.append(time.perf_counter() - t0)
        self.fence()
        dt = time.perf_counter() - t0
        if self.world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt


def region_trend(regions):
    """(median of the last third - median of the first third) / median of all: a settled run is within +-1 %."""
    k = max(1, len(regions) // 3)
    med = statistics.median(regions)
    return (statistics.median(regions[-k:]) - statistics.median(regions[:k])) / med if med > 0 else 0.0

#!/usr/bin/env python3
"""Condenses a tools/profile_gpu.sh output directory into a short text summary + traffic.json (kept under profiles/)."""
import csv
import glob
import json
import os
import sys

FILL_BYTES = 693633024          # tools/hbm_write_ceiling.py buffer


def _fingerprint():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from xworld_amd import build
    return build.source_fingerprint()


def counter_by_kernel(out, tag, counter):
    agg = {}
    for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                a = agg.setdefault(r.get("Kernel_Name", ""), [0, 0.0])
                a[0] += 1
                a[1] += float(r.get("Counter_Value", 0))
    return agg


def main(out, bench_args):
    workload = "xworld7"
    if "--workload" in bench_args:
        workload = bench_args[bench_args.index("--workload") + 1]
    print("== rocprofv3 --kernel-trace --stats (python bench.py --steps 100 --warmup 10 %s) ==" % " ".join(bench_args))
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as fh:
            rows = list(csv.DictReader(fh))
        print("%-70s %8s %14s %12s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
        for r in rows[:12]:
            print("%-70s %8s %14s %12s %8s" % (r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"),
                                               r.get("AverageNs"), r.get("Percentage")))
    per_kernel = {}
    agg_count = {}
    for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        agg = counter_by_kernel(out, tag, counter)
        print("== rocprofv3 --pmc %s (raw counter, KiB per dispatch) ==" % counter)
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
            print("%-70s dispatches %5d  %.1f" % (k[:70], n, v / n))
            per_kernel.setdefault(k, {})[counter] = v / n * 1024.0
            agg_count[k] = n
    # calibration: bytes the counters report for a fill / copy of exactly FILL_BYTES
    cal = {}
    for tag, counter in (("cal_write", "WRITE_SIZE"), ("cal_fetch", "FETCH_SIZE")):
        agg = counter_by_kernel(out, tag, counter)
        print("== calibration: %s over tools/hbm_write_ceiling.py (buffers of %d B) ==" % (counter, FILL_BYTES))
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:4]:
            b = v / n * 1024.0
            print("%-70s dispatches %5d  %.0f B per dispatch = %.3f x buffer" % (k[:70], n, b, b / FILL_BYTES))
            cal.setdefault(counter, {})[k[:60]] = b / FILL_BYTES
    # (xw_step_render_kernel: the whole-batch render with the step's blocks in the same launch, the default loop since round 6)
    dom = [k for k in per_kernel if "render_all" in k or "step_render" in k or "sg_kernel" in k or "race_kernel" in k or "render_ego" in k]
    span = [k for k in per_kernel if "xw_ego_gather_kernel" in k]
    if dom or span:
        # write side: the fill kernels of the calibration report ~1.0 x -> WRITE_SIZE taken at face value;
        # read side: FETCH_SIZE x 2 (gfx950 correction for 16 B/lane streaming reads, MI355X_MICROARCH.md "HBM")
        if span:
            # the egocentric span path: the whole-batch render is four launches per step (bench.py times them together)
            group = [name for name in per_kernel if any(t in name for t in ("xw_ego_cells_kernel", "xw_ego_eval_kernel", "xw_ego_gather_list_kernel", "xw_ego_gather_kernel"))]
            k = " + ".join(sorted(name.split("(")[0].replace("void xwb::", "") for name in group))
            w = sum(per_kernel[name].get("WRITE_SIZE", 0.0) for name in group)
            f = sum(per_kernel[name].get("FETCH_SIZE", 0.0) for name in group)
        else:
            k = max(dom, key=lambda name: per_kernel[name].get("WRITE_SIZE", 0.0) * agg_count.get(name, 1))   # the step loop's kernel
            w = per_kernel[k].get("WRITE_SIZE", 0.0)
            f = per_kernel[k].get("FETCH_SIZE", 0.0)
        traffic = {"workload": workload, "kernel": k[:200], "write_bytes_per_launch": w, "fetch_bytes_per_launch_raw": f,
                   "fetch_bytes_per_launch_corrected": 2 * f, "traffic_bytes_per_launch": w + 2 * f,
                   "calibration": cal, "source": os.path.basename(out),
                   # which code this describes: bench.py compares source_sha16 with the sources it runs and flags a stale quote
                   "commit": os.environ.get("GIT_HEAD", "unknown"), "source_sha16": _fingerprint()}
        with open(os.path.join(out, "traffic.json"), "w") as fh:
            json.dump(traffic, fh, indent=1)
        print("== traffic ==")
        print(json.dumps(traffic))
    log = os.path.join(out, "bench_under_rocprof.log")
    if os.path.exists(log):
        print("== bench line under rocprof ==")
        lines = [l for l in open(log).read().splitlines() if l.startswith('{"metric"')]     # (rocprofv3 logs after the line)
        print(lines[-1][:2500] if lines else "(no bench line)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])

#!/usr/bin/env python3
"""Condenses a tools/profile_gpu.sh output directory into a short text summary (kept under profiles/)."""
import csv
import glob
import os
import sys


def main(out):
    stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
    print("== rocprofv3 --kernel-trace --stats (python bench.py --steps 100 --warmup 10) ==")
    for f in stats:
        with open(f) as fh:
            rows = list(csv.DictReader(fh))
        print("%-70s %8s %14s %12s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
        for r in rows[:12]:
            print("%-70s %8s %14s %12s %8s" % (r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"),
                                               r.get("AverageNs"), r.get("Percentage")))
    for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        files = glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True)
        print("== rocprofv3 --pmc %s ==" % counter)
        agg = {}
        for f in files:
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if r.get("Counter_Name") != counter:
                        continue
                    k = r.get("Kernel_Name", "")[:70]
                    a = agg.setdefault(k, [0, 0.0])
                    a[0] += 1
                    a[1] += float(r.get("Counter_Value", 0))
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
            print("%-70s dispatches %5d  %s per dispatch (raw counter units, KB): %.1f" % (k, n, counter, v / n))
    log = os.path.join(out, "bench_under_rocprof.log")
    if os.path.exists(log):
        print("== bench line under rocprof ==")
        print(open(log).read().strip().splitlines()[-1][:2000])


if __name__ == "__main__":
    main(sys.argv[1])

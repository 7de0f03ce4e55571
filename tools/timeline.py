#!/usr/bin/env python3
"""Per-iteration kernel timeline from a rocprofv3 --kernel-trace CSV: start offsets, durations, gaps (us)."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
steps = [i for i, r in enumerate(rows) if "xw_step_kernel" in r["Kernel_Name"] or "xw_step_render_kernel" in r["Kernel_Name"]]
k = steps[len(steps) // 2]
t0 = int(rows[k]["Start_Timestamp"])
for r in rows[k:steps[len(steps) // 2 + 2]]:
    print("%9.1f %9.1f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                    r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))

#!/usr/bin/env python3
"""Markdown rows of DESIGN.md section 7 from the bench lines of a round: python tools/bench_table.py profiles/r4"""
import glob
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "profiles/r4"
order = ["xworld7", "xworld7_driver_args", "xworld7_autoreset", "xworld7_f32", "xworld8", "xworld11", "xworld7_ego3", "xworld8_ego5",
         "xworld7_ego7", "simple_game", "simple_race"]
print("| workload | env-steps/s | ms/step (9 regions: min-max) | path | dominant kernel avg us (events) | frac of 8 TB/s: kernel / whole loop | of the run's own write ceiling | kernels of the step, avg us | HBM traffic per launch (PMC) vs algorithmic | `step_autoreset` | CPU oracle 1 thread / all cores |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for w in order:
    f = os.path.join(d, "bench_%s.json" % w)
    if not os.path.exists(f):
        continue
    l = json.load(open(f))
    r = l["roofline"]
    reg = l["regions"]
    ks = ", ".join("%s %.1f" % (k, v["avg_us"]) for k, v in r.get("kernels_us", {}).items())
    tr = "%.1f vs %.1f MB (%.3fx)" % (r["traffic"] / 1e6, r["algorithmic_bytes_per_launch"] / 1e6, r["traffic"] / r["algorithmic_bytes_per_launch"]) if r.get("traffic") else "-"
    ar = l.get("step_autoreset")
    ars = "%.1f M, %.4f ms" % (ar["value"] / 1e6, ar["ms_per_step"]) if ar else "-"
    cb = l.get("cpu_baseline")
    cbs = "%.1f k / %.1f k (%d)" % (cb["single_thread_value"] / 1e3, cb["value"] / 1e3, cb["cores"]) if cb else "-"
    val = l["value"]
    vs = "%.2f G" % (val / 1e9) if val > 2e9 else "%.1f M" % (val / 1e6)
    print("| %s | **%s** | %.4f (%.4f-%.4f) | %s / %s | %.1f | %.3f / %.3f | %.2f (%.0f GB/s memset) | %s | %s | %s | %s |" % (
        w, vs, l["ms_per_step"], reg["ms_per_step_min"], reg["ms_per_step_max"], l["path"]["path"], l["path"]["queue_sync"],
        r["kernel_avg_us"], r["frac"], r["step_loop_frac"], r["frac_of_write_ceiling"] or 0, r["write_ceiling_GBps"], ks, tr, ars, cbs))

#!/usr/bin/env python3
"""Markdown tables of DESIGN.md section 7 / docs/measurements.md from the bench lines of a round (rows <= 150 columns):
    python tools/bench_table.py profiles/r5"""
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "profiles/r5"
order = ["xworld7", "xworld7_driver_args", "xworld7_autoreset", "xworld7_f32", "xworld8", "xworld11", "xworld7_ego3", "xworld8_ego5",
         "xworld7_ego7", "xworld11_ego9", "simple_game", "simple_race"]
lines = {}
for w in order:
    f = os.path.join(d, "bench_%s.json" % w)
    if os.path.exists(f):
        try:
            txt = [x for x in open(f).read().splitlines() if x.startswith('{"metric"')]
            lines[w] = json.loads(txt[-1])
        except Exception:
            pass


def val(v):
    return "%.2f G" % (v / 1e9) if v > 2e9 else "%.1f M" % (v / 1e6)


print("| workload | env-steps/s | ms/step (min-max of 9) | trend | path | host us/step | `step_autoreset` | CPU oracle 1 / all cores |")
print("|---|---|---|---|---|---|---|---|")
for w, l in lines.items():
    reg = l["regions"]
    ar = l.get("step_autoreset")
    cb = l.get("cpu_baseline")
    print("| %s | **%s** | %.4f (%.4f-%.4f) | %+.2f %% | %s / %s | %.1f | %s | %s |" % (
        w, val(l["value"]), l["ms_per_step"], reg["ms_per_step_min"], reg["ms_per_step_max"], 100 * reg.get("trend", 0.0),
        l["path"]["path"], l["path"]["queue_sync"], l.get("host_us_per_step", 0.0),
        ("%s, %.4f ms" % (val(ar["value"]), ar["ms_per_step"])) if ar else "-",
        ("%.1f k / %.1f k (%d)" % (cb["single_thread_value"] / 1e3, cb["value"] / 1e3, cb["cores"])) if cb else "-"))
print()
print("| workload | dominant kernel, avg us (events) | frac of 8 TB/s: kernel / loop | of the run's write ceiling | HBM traffic (PMC) vs algorithmic | kernels of the step, avg us |")
print("|---|---|---|---|---|---|")
for w, l in lines.items():
    r = l["roofline"]
    ks = ", ".join("%s %.1f" % (k, v["avg_us"]) for k, v in r.get("kernels_us", {}).items())
    tr = "%.1f vs %.1f MB (%.3fx)%s" % (r["traffic"] / 1e6, r["algorithmic_bytes_per_launch"] / 1e6, r["traffic"] / r["algorithmic_bytes_per_launch"],
                                        " stale" if r.get("traffic_stale") else "") if r.get("traffic") else "-"
    print("| %s | %.1f | %.3f / %.3f | %.2f (%.0f GB/s) | %s | %s |" % (
        w, r["kernel_avg_us"], r["frac"], r["step_loop_frac"], r["frac_of_write_ceiling"] or 0, r["write_ceiling_GBps"], tr, ks))

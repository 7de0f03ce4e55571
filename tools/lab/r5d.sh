#!/bin/bash
# On the GPU box: (1) goal-cell workgroups of the evaluation launch, A/B; (2) simple games: is the default loop host-bound?
set -u
OUT=$PWD/gpurun_out/${1:-r5d}
mkdir -p $OUT
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); r = d["roofline"]
    print("%s: %.1f M env-steps/s, %.4f ms/step, host %.1f us/step, kernel %.1f us, kernels %s, trend %.4f, autoreset %.1f M" % (
        sys.argv[2], d["value"] / 1e6, d["ms_per_step"], d.get("host_us_per_step", 0), r["kernel_avg_us"],
        {k: round(v["avg_us"], 1) for k, v in r["kernels_us"].items()}, d["regions"]["trend"], d.get("step_autoreset", {}).get("value", 0) / 1e6))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
}
for WL in xworld7_ego3 xworld8_ego5 xworld7_ego7; do
  for N in 0 2048 8192 16384; do
    XWB_DEBUG=ego_miss_blocks=$N timeout 600 python bench.py --workload $WL --steps 100 --warmup 20 --repeats 5 --no-cpu-baseline 2>$OUT/b_${WL}_$N.err | grep '^{"metric"' | tail -1 > $OUT/b_${WL}_$N.json
    show $OUT/b_${WL}_$N.json "$WL ego_miss_blocks=$N"
  done
done | tee $OUT/miss_blocks.txt
for WL in simple_game simple_race; do
  timeout 600 python bench.py --workload $WL --no-cpu-baseline 2>$OUT/b_$WL.err | grep '^{"metric"' | tail -1 > $OUT/b_$WL.json
  show $OUT/b_$WL.json "$WL"
done | tee $OUT/simple.txt
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_nccl_branch.py tests/test_gpu_bench_line.py -q -x --timeout 600 2>&1 | tail -5 | tee $OUT/tests.txt

#!/bin/bash
# On the GPU box: the fused egocentric render (XWB_DEBUG=ego_fused=N) against the three-kernel span path, same box, same run.
# Usage: tools/lab/ego_fused_ab.sh <outdir> [workloads...]
set -u
# (the fused kernel is a lab build)
[ -n "${NO_BUILD:-}" ] || XWB_EXTRA_FLAGS=-DXWB_EGO_FUSED_LAB python -m xworld_amd.build > /dev/null 2>&1
OUT=$PWD/gpurun_out/${1:-ego_fused}; shift || true
WLS=${@:-xworld7_ego3}
mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== frame parity, fused path (tests/test_gpu_ego.py, test_gpu_doc_image.py under XWB_DEBUG=ego_fused=1) =="
XWB_DEBUG=ego_fused=1 timeout 900 python -m pytest tests/test_gpu_ego.py tests/test_gpu_doc_image.py -q -x --timeout 600 2>&1 | tail -15
} > $OUT/parity.txt 2>&1
tail -5 $OUT/parity.txt
for WL in $WLS; do
  for V in 0 1 3 17 19 33 35; do
    XWB_DEBUG=ego_fused=$V timeout 600 python bench.py --workload $WL --steps 100 --warmup 20 --repeats 5 --no-cpu-baseline $([ $((V & 2)) = 2 ] && echo --no-parity) 2>$OUT/bench_${WL}_$V.err | grep '^{"metric"' | tail -1 > $OUT/bench_${WL}_$V.json
    python - $OUT/bench_${WL}_$V.json $WL $V <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    r = d["roofline"]
    print("%s ego_fused=%s: %.1f M env-steps/s, %.4f ms/step, render %.1f us, kernels %s, autoreset %.1f M, parity %s" % (
        sys.argv[2], sys.argv[3], d["value"] / 1e6, d["ms_per_step"], r["kernel_avg_us"],
        {k: round(v["avg_us"], 1) for k, v in r["kernels_us"].items()}, d.get("step_autoreset", {}).get("value", 0) / 1e6,
        {k: v for k, v in d.get("parity", {}).items() if k in ("mismatches",)} | {"frames": d.get("parity", {}).get("frames", {}).get("mismatches")}))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "no line:", e)
PY
  done
done | tee $OUT/ab.txt
# kernel trace of the fused path (variant 1) and of the three-kernel path
cd /tmp
for V in 0 1; do
  XWB_DEBUG=ego_fused=$V rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_$V -- python $OLDPWD/bench.py --workload xworld7_ego3 --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline --no-parity > $OUT/trace_$V.log 2>&1
  f=$(ls $OUT/trace_$V/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== kernel stats ego_fused=$V =="; [ -n "$f" ] && head -12 "$f" | cut -c1-200
done | tee $OUT/trace.txt

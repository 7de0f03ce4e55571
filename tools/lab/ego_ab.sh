#!/bin/bash
# On the GPU box: A/B of the egocentric span render -- builds (XWB_EXTRA_FLAGS) x XWB_DEBUG settings, same box, same call.
# Usage: tools/lab/ego_ab.sh <outdir> "<flags1>|<flags2>|..." "<debug1>|<debug2>|..." [workloads...]
set -u
OUT=$PWD/gpurun_out/${1:-ego_ab}; IFS='|' read -ra BUILDS <<< "${2:-}"; IFS='|' read -ra DEBUGS <<< "${3:-}"; shift 3 || true
WLS=${*:-xworld7_ego3}
mkdir -p $OUT; export TMPDIR=/tmp
[ ${#BUILDS[@]} = 0 ] && BUILDS=("")
[ ${#DEBUGS[@]} = 0 ] && DEBUGS=("")
bi=0
for B in "${BUILDS[@]}"; do
  XWB_EXTRA_FLAGS="$B" python -m xworld_amd.build > $OUT/build_$bi.log 2>&1 || { echo "build '$B' failed"; tail -5 $OUT/build_$bi.log; bi=$((bi+1)); continue; }
  di=0
  for D in "${DEBUGS[@]}"; do
    for WL in $WLS; do
      XWB_DEBUG="$D" timeout 600 python bench.py --workload $WL --steps 100 --warmup 20 --repeats 5 --no-cpu-baseline 2>$OUT/bench_${WL}_${bi}_$di.err | grep '^{"metric"' | tail -1 > $OUT/bench_${WL}_${bi}_$di.json
      python - "$OUT/bench_${WL}_${bi}_$di.json" "$WL" "$B" "$D" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); r = d["roofline"]
    print("%s build[%s] debug[%s]: %.1f M env-steps/s, %.4f ms/step, render %.1f us (frac %.3f), kernels %s, autoreset %.1f M, frames %s" % (
        sys.argv[2], sys.argv[3], sys.argv[4], d["value"] / 1e6, d["ms_per_step"], r["kernel_avg_us"], r["frac"],
        {k: round(v["avg_us"], 1) for k, v in r["kernels_us"].items()}, d.get("step_autoreset", {}).get("value", 0) / 1e6,
        (d.get("parity") or {}).get("frames", {}).get("mismatches")))
except Exception as e:
    print(sys.argv[2], sys.argv[3], sys.argv[4], "no line:", e)
PY
    done
    di=$((di+1))
  done
  if [ -n "${TRACE:-}" ]; then
    R=$PWD; cd /tmp
    rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_$bi -- python $R/bench.py --workload ${WLS%% *} --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline --no-parity > $OUT/trace_$bi.log 2>&1
    cd $R; echo "== timeline build[$B]"; python tools/timeline.py $OUT/trace_$bi | head -16 | cut -c1-120
  fi
  bi=$((bi+1))
done
python -m xworld_amd.build > /dev/null 2>&1

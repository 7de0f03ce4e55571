#!/bin/bash
# On the GPU box (via gpurun): the round's evidence for profiles/ -- per workload a plain bench line, the rocprofv3 kernel-trace
# summary of the same command and the calibrated HBM traffic of its dominant kernel (tools/profile_gpu.sh).
# Usage: GIT_HEAD=$(git rev-parse HEAD) tools/profile_round.sh <round tag> <workload>...   -> gpurun_out/profiles_<tag>/
set -u
TAG=${1:-r3}; shift
OUT=$PWD/gpurun_out/profiles_$TAG
mkdir -p $OUT
for W in "$@"; do
  # counters first: the bench line quotes profiles/<tag>/traffic_<workload>.json, which then is this HEAD's (traffic_stale false)
  bash tools/profile_gpu.sh $W --workload $W > /dev/null 2>&1
  cp gpurun_out/prof_$W/summary.txt $OUT/${W}_rocprof_summary.txt 2>/dev/null
  cp gpurun_out/prof_$W/traffic.json $OUT/traffic_$W.json 2>/dev/null
  mkdir -p profiles/$TAG && cp gpurun_out/prof_$W/traffic.json profiles/$TAG/traffic_$W.json 2>/dev/null
  rm -rf gpurun_out/prof_$W                                # (raw traces: ~25 MB per workload; gpurun brings back at most 64 MiB)
  python bench.py --workload $W > $OUT/bench_$W.json 2> $OUT/bench_$W.err || true
  tail -c 300 $OUT/bench_$W.json | head -c 300; echo
done
python bench.py --workload xworld7 --autoreset --no-cpu-baseline > $OUT/bench_xworld7_autoreset.json 2>/dev/null || true
# SQ / TA / TCP counters of the egocentric span path per radius (tools/pmc_ego_span.sh)
for RM in "3 7" "5 8" "7 7"; do
  set -- $RM
  bash tools/pmc_ego_span.sh $1 $2 > /dev/null 2>&1
  { echo "== commit ${GIT_HEAD:-unknown} (tools/pmc_ego_span.sh $1 $2) =="; cat gpurun_out/pmc_ego_span_r$1/summary.txt; } > $OUT/ego_r$1_span_counters.txt
done
python tools/ego_sweep.py > $OUT/ego_sweep.txt 2>&1 || true
python bench.py --steps 20 --warmup 5 > $OUT/bench_xworld7_driver_args.json 2>/dev/null || true
ls -la $OUT

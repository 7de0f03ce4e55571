#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + HBM PMC passes for the bench workload.
# Usage: GIT_HEAD=$(git rev-parse HEAD) tools/profile_gpu.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
# (the GPU box has no .git: the commit the profile describes comes in through GIT_HEAD and is stamped on the summary)
set -u
TAG=${1:-r1}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
# 1. per-kernel time (same command as the bench line)
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -- python $REPO/bench.py --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline --no-secondary "$@" > $OUT/bench_under_rocprof.log 2>&1
# 2. HBM traffic counters, separate passes (FETCH_SIZE and WRITE_SIZE cannot share a pass)
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -- python $REPO/bench.py --steps 4 --warmup 2 --repeats 1 --spin-seconds 0 --no-parity --no-cpu-baseline --no-secondary "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -- python $REPO/bench.py --steps 4 --warmup 2 --repeats 1 --spin-seconds 0 --no-parity --no-cpu-baseline --no-secondary "$@" > $OUT/pmc_write.log 2>&1
# 3. counter calibration on a known byte count in a similar access pattern: a 693 633 024-byte device fill
#    (write stream) and copy (16 B/lane read stream) -- MI355X_MICROARCH.md: FETCH_SIZE under-reports wide
#    streaming reads by 2x on gfx950, WRITE_SIZE is uncalibrated
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/cal_write -- python $REPO/tools/hbm_write_ceiling.py > $OUT/cal_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/cal_fetch -- python $REPO/tools/hbm_write_ceiling.py > $OUT/cal_fetch.log 2>&1
cd $REPO
echo "== commit ${GIT_HEAD:-unknown} ==" > $OUT/summary.txt
python tools/summarize_prof.py $OUT "$@" >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt

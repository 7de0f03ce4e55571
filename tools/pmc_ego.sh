#!/bin/bash
# SQ instruction-mix counters of the egocentric render (runs on the GPU box): tools/pmc_ego.sh <tag> [bench args]
set -u
TAG=${1:-ego}; shift || true
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -f csv -d $OUT/a -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -f csv -d $OUT/b -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/b.log 2>&1
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k, row["Counter_Name"])] += 1
    for k, d in acc.items():
        if "ego" not in k and "xw_" not in k: continue
        print(k)
        for c, v in sorted(d.items()):
            n = cnt[(k, c)]
            print("   %-22s %14.0f per launch (%d launches)" % (c, v / n, n))
PY

#!/bin/bash
# tools/pmc_kernel.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- per-kernel PMC averages for the bench workload
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
i=0
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C -f csv -d $OUT/pass$i -- python $REPO/bench.py --steps 6 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/pass$i.log 2>&1
done
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        a = agg[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in agg.items():
    print(k)
    for c, (n, v) in sorted(d.items()):
        print("    %-28s avg/dispatch %14.1f  (n=%d)" % (c, v / n, n))
PY

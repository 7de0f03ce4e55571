#!/bin/bash
# On the GPU box: SQ / TA counters of the egocentric span path's kernels (separate --pmc passes; the library falls back to events
# under counter collection).  tools/_lab-free copy of what DESIGN.md section 5 quotes.  Output: gpurun_out/pmc_ego_span/summary.txt
# Usage: tools/pmc_ego_span.sh [r = 3] [max_dim = 7]   -> gpurun_out/pmc_ego_span_r<r>/summary.txt
R=${1:-3}; MD=${2:-7}
cd /tmp; export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_ego_span_r$R
rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<PY
import sys
sys.path.insert(0, '/root/repo')
import torch
from xworld_amd.batched import BatchedSimulator
o = {"xwd_conf_path": "/root/repo/xworld_amd/confs/navigation2d.json", "task_mode": "lang_acquisition", "max_dim": $MD, "dim": $MD, "num_blocks": 16,
     "visible_radius": $R, "color": True}
sim = BatchedSimulator("xworld", o, num_envs=32768, seed=12648430)
sim.reset()
for i in range(30):
    sim.step(); sim.reset_done()
torch.cuda.synchronize()
PY
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU TA_BUSY_avr" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set -f csv -d $OUT/p$i -o p -- python $OUT/run.py > $OUT/p$i.log 2>&1
done
python - $R $MD <<'PY' > $OUT/summary.txt
import csv, glob, collections, sys
R, MD = sys.argv[1], sys.argv[2]
print("== rocprofv3 --pmc, egocentric span path, 32 768 envs, r = %s on a %sx%s world, colour: average counter value per dispatch ==" % (R, MD, MD))
for f in sorted(glob.glob('/root/repo/gpurun_out/pmc_ego_span_r%s/p*/*counter_collection.csv' % R)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'xw_ego_' in k and 'build' not in k:
            agg[k.split('(')[0].replace('void xwb::', '')][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print("%-52s %s" % (k, {c: round(sum(x) / len(x)) for c, x in sorted(v.items())}))
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
cat $OUT/summary.txt

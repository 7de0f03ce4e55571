#!/bin/bash
# On the GPU box: SQ / TA counters of the egocentric span path's kernels (separate --pmc passes; the library falls back to events
# under counter collection).  tools/_lab-free copy of what DESIGN.md section 5 quotes.  Output: gpurun_out/pmc_ego_span/summary.txt
cd /tmp; export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_ego_span
rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<'PY'
import sys
sys.path.insert(0, '/root/repo')
import torch
from xworld_amd.batched import BatchedSimulator
o = {"xwd_conf_path": "/root/repo/xworld_amd/confs/navigation2d.json", "task_mode": "lang_acquisition", "max_dim": 7, "dim": 7, "num_blocks": 16,
     "visible_radius": 3, "color": True}
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
python - <<'PY' > $OUT/summary.txt
import csv, glob, collections
print("== rocprofv3 --pmc, egocentric span path, 32 768 envs, r = 3, 84x84x3: average counter value per dispatch ==")
for f in sorted(glob.glob('/root/repo/gpurun_out/pmc_ego_span/p*/*counter_collection.csv')):
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

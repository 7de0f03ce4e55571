#!/bin/bash
# On the GPU box (via gpurun): everything profiles/<tag>/ holds, from ONE snapshot of the sources -- the GPU suite, the
# per-workload bench lines + kernel traces + HBM counters (tools/profile_round.sh), the soaks, the PCIe-inclusive rates, the
# forced one-rank exchange and the leak check.  Usage: GIT_HEAD=$(git rev-parse --short HEAD) tools/final_round.sh <tag>
set -u
TAG=${1:-r6}
OUT=$PWD/gpurun_out/profiles_$TAG
mkdir -p $OUT
HEAD=${GIT_HEAD:-unknown}
{ echo "== python -m pytest tests -q -m gpu at commit $HEAD, one MI355X =="; timeout 2400 python -m pytest tests -q -m gpu --slow --timeout 600 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path"; } > $OUT/gputest.txt
tail -2 $OUT/gputest.txt
GIT_HEAD=$HEAD bash tools/profile_round.sh $TAG xworld7 xworld7_f32 xworld8 xworld11 xworld7_ego3 xworld8_ego5 xworld7_ego7 xworld11_ego9 simple_game simple_race > $OUT/profile_round.log 2>&1
{
  echo "== soaks at commit $HEAD (tools/soak.py, tools/ego_soak.py, tools/soak_pregen.py, tools/pcie_rate.py; one MI355X) =="
  for a in "nav7 0 4096 2000" "nav7 3 4096 1500" "nav8_dim5 5 2048 1500" "walls7 0 4096 1500 2d" "nav8 0 4096 3000" "nav8 0 2048 6000 curriculum=0.1" \
           "nav8 0 4096 2000 weights=1,2,3,4,5" "nav8 0 4096 2500 2d" "nav7 7 2048 1200" "nav8 3 1024 4000 curriculum=0.1"; do
    timeout 900 python tools/soak.py $a 2>&1 | grep -E "^soak|mismatch|Error|error" | cut -c1-300
  done
  timeout 900 python tools/ego_soak.py 2>&1 | tail -14
  timeout 900 python tools/soak_pregen.py 2>&1 | tail -3
  for a in "6000 40 7" "3000 40 11" "3000 40 8"; do timeout 600 python tools/soak_fused.py $a 2>&1 | grep "^soak_fused"; done
  timeout 600 python tools/pcie_rate.py 2>&1 | tail -3
  timeout 600 python tools/leak_check.py 2>&1 | grep "^no leak"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
} > $OUT/soak.txt
# >= 3 seeds per headline number
for SEED in 1 2 3; do
  timeout 600 python bench.py --seed $SEED --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/bench_xworld7_seed$SEED.json
done
# the default loop as two launches per step (the step kernel in front of the render: rounds 1-5), same box, same run
XWB_DEBUG=no_fused timeout 600 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/bench_xworld7_no_fused.json
timeout 60 python bench.py --gpus 8 --dry-run > $OUT/bench_dry_run_n8.json 2>/dev/null
for X in torch lib; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2993$([ $X = lib ] && echo 2 || echo 1) \
    bench.py --gpus 1 --backend nccl --force-exchange --exchange $X --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/bench_xworld7_forced_exchange_$X.json
done
tail -12 $OUT/soak.txt
ls $OUT | wc -l

#!/bin/bash
# On the GPU box: (1) forced one-rank exchange, the step's stream waiting for the results gather or not; (2) the two-rank bench on one
# GPU (watchdog phases, ranks record); (3) stage stamps of xw_step_kernel / xw_render_list_kernel (lab build -DXWB_STEP_PROF).
set -u
OUT=$PWD/gpurun_out/${1:-stepprof}
mkdir -p $OUT
for W in "" "--results-wait"; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29931 bench.py --gpus 1 --backend nccl --force-exchange --no-cpu-baseline $W 2>$OUT/forced$W.err | grep '^{"metric"' | tail -1 > $OUT/forced$W.json
  python -c "
import json,sys
d=json.load(open('$OUT/forced$W.json')); print('forced exchange $W:', round(d['value']/1e6,1), 'M', d['ms_per_step'], d['config']['exchange'], d.get('host_us_per_step'), d['rccl'], d.get('phase_seconds'))"
done 2>&1 | tee $OUT/forced.txt
python bench.py --no-cpu-baseline 2>$OUT/plain.err | grep '^{"metric"' | tail -1 > $OUT/plain.json
python -c "
import json
d=json.load(open('$OUT/plain.json')); print('plain:', round(d['value']/1e6,1), 'M', d['ms_per_step'], 'host_us', d.get('host_us_per_step'), 'trend', d['regions']['trend'])" | tee -a $OUT/forced.txt
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_nccl_branch.py -q -x --timeout 600 2>&1 | tail -5 | tee $OUT/sharded_tests.txt
XWB_EXTRA_FLAGS=-DXWB_STEP_PROF python -m xworld_amd.build > $OUT/build.log 2>&1
python tools/step_prof.py xworld7 2>&1 | tail -20 | tee $OUT/step_prof.txt

#!/bin/bash
# On the GPU box: the results exchange beside the step loop (xwb_gather_results_beside) -- its test, then the forced one-rank
# bench: no exchange / torch (released, waited) / library (released, waited), same box, same call.
set -u
OUT=$PWD/gpurun_out/${1:-r5g}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_lib_gather.py tests/test_gpu_nccl_branch.py -q --timeout 600 2>&1 | tail -6 | tee $OUT/tests.txt
run() {  # name, args...
  local name=$1; shift
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29941 bench.py --gpus 1 --backend nccl --force-exchange --no-cpu-baseline --no-screens-gather "$@" 2>$OUT/$name.err | grep '^{"metric"' | tail -1 > $OUT/$name.json
  python -c "
import json
d=json.load(open('$OUT/$name.json')); print('$name:', round(d['value']/1e6,1), 'M env-steps/s,', round(d['ms_per_step']*1e3,1), 'us/step, host', round(d.get('host_us_per_step',0),1), 'us/step;', d['config']['exchange'])"
}
{
python bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/plain.json
python -c "
import json
d=json.load(open('$OUT/plain.json')); print('no exchange:', round(d['value']/1e6,1), 'M env-steps/s,', round(d['ms_per_step']*1e3,1), 'us/step, host', round(d.get('host_us_per_step',0),1), 'us/step')"
run torch_released
run torch_waited --results-wait
run lib_released --exchange lib
run lib_waited --exchange lib --results-wait
run torch_released_2
run lib_released_2 --exchange lib
} 2>&1 | tee $OUT/forced.txt

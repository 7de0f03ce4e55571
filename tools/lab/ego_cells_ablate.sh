#!/bin/bash
# On the GPU box: the whole-batch cells kernel cut short at each of its stages (-DEGO_CELLS_STOP=i: wrong frames), its duration from a
# kernel trace -- what each stage costs without stamps in the way.  Usage: tools/lab/ego_cells_ablate.sh <r> <max_dim>
set -u
R=${1:-3}; D=${2:-7}; OUT=$PWD/gpurun_out/cells_ablate_r$R; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
for STOP in 0 1 7 8 2 9 4 5 none; do
  FL=""; [ $STOP != none ] && FL="-DEGO_CELLS_STOP=$STOP"
  XWB_EXTRA_FLAGS="$FL" python -m xworld_amd.build > /dev/null 2>&1 || { echo "build $STOP failed"; continue; }
  cd /tmp; rm -rf $OUT/t_$STOP
  timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $OUT/t_$STOP -- python $REPO/tools/lab/ego_steps_only.py $R $D > $OUT/t_$STOP.log 2>&1
  cd $REPO
  f=$(find $OUT/t_$STOP -name "*kernel_stats.csv" | head -1)
  echo "stop after stage $STOP: $(grep 'xw_ego_cells_kernel<.*false>' $f | head -1 | awk -F, '{printf "cells kernel avg %.1f us (%s calls)", $4/1000, $2}')"
done
python -m xworld_amd.build > /dev/null 2>&1

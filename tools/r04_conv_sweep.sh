#!/bin/bash
# Round 4, GPU box (library built with WCT_BUILD_TUNING=1): tile configurations on the narrow conv layers.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04_conv_sweep.txt
: > $OUT
for B in 8 16; do
for CFG in 0 2 3 4; do
  echo "== batch $B WCT_CONV_CFG=$CFG" >> $OUT
  WCT_CONV_CFG=$CFG timeout 300 python tools/bench_conv.py $B 128 2>&1 | grep -v amdgpu.ids >> $OUT
done
done
cat $OUT

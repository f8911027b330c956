#!/bin/bash
# (tuning build) stream groups of the eigensolver / V-pass threshold at batch 32 and 8, same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04_ngrp.txt
: > $OUT
for V in "WCT_EIG_NGRP=1" "WCT_EIG_NGRP=2" "WCT_JACOBI_VSTRIP_MIN=12" "WCT_EIG_NGRP=1"; do
for B in 32 8; do
  env $V timeout 300 python bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
b=l['breakdown_ms_per_step']
print('$V batch %3d: %.1f frames/s, %.2f ms/step; jacobi %.2f, conv3x3 %.2f' % (l['config']['global_batch'], l['value'], l['ms_per_step'], b['jacobi'], b['conv3x3']))" >> $OUT 2>&1
done
done
cat $OUT

#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out
python tools/wct_tol_probe.py > $OUT/r02d_tol.log 2>&1
for pw in 1 0; do for tol in 1e-2 5e-3 3e-3 2e-3 1e-3; do
  WCT_JACOBI_PW=$pw WCT_JACOBI_CONV_TOL=$tol python tools/wct_tol_probe.py >> $OUT/r02d_tol.log 2>&1
done; done
cat $OUT/r02d_tol.log | grep -v amdgpu.ids
for tol in 1e-2 5e-3 3e-3; do
  WCT_JACOBI_CONV_TOL=$tol WCT_EIG_NGRP=4 python bench.py --batch 32 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tol=$tol b32 fps %.1f jacobi %.2f'%(d['value'], d['breakdown_ms_per_step']['jacobi']))"
done

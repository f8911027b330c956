#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out
python tools/wct_tol_probe.py > /dev/null 2>&1
: > $OUT/r02e_tol.log
for cor in 1 0; do for tol in 1e-1 5e-2 3e-2 1e-2 3e-3; do
  echo -n "correct=$cor " >> $OUT/r02e_tol.log
  WCT_EIG_CORRECT=$cor WCT_JACOBI_CONV_TOL=$tol python tools/wct_tol_probe.py 2>/dev/null >> $OUT/r02e_tol.log
done; done
cat $OUT/r02e_tol.log
for tol in 1e-1 5e-2 3e-2 1e-2; do
  WCT_JACOBI_CONV_TOL=$tol WCT_EIG_NGRP=4 python bench.py --batch 32 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tol=$tol b32 fps %.1f'%d['value'], {k: round(v,2) for k,v in d['breakdown_ms_per_step'].items()})"
done
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/r02e_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/r02e_pytest.log
grep -E "passed|failed" $OUT/r02e_pytest.log | tail -3
grep -E "^FAILED" $OUT/r02e_pytest.log

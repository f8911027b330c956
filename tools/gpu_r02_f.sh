#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out
python tools/wct_tol_probe.py > /dev/null 2>&1
: > $OUT/r02h_tol.log
for tf in 2e-2 1e-2; do
  echo -n "tol_fn=$tf " >> $OUT/r02h_tol.log
  WCT_JACOBI_TOL_FN=$tf python tools/wct_tol_probe.py 2>/dev/null >> $OUT/r02h_tol.log
done
cat $OUT/r02h_tol.log
for tf in 2e-2 1e-2; do
  WCT_JACOBI_TOL_FN=$tf WCT_EIG_NGRP=4 python bench.py --batch 32 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tol_fn=$tf b32 fps %.1f'%d['value'], {k: round(v,2) for k,v in d['breakdown_ms_per_step'].items()})"
done
timeout 600 python -m pytest tests -m gpu -q -s --durations=15 > $OUT/r02h_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/r02h_pytest.log
grep -E "passed|failed" $OUT/r02h_pytest.log | tail -3
grep -E "^FAILED" $OUT/r02h_pytest.log

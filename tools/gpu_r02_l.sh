#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "eigh or wct_matches or synthetic or rank or straddle or loud" 2>&1 | tail -5
for pw in 1 2; do for b in 32 8 1; do
  WCT_JACOBI_PW=$pw python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pw=$pw b=$b fps %.1f ms %.2f jacobi %.2f'%(d['value'],d['ms_per_step'],d['breakdown_ms_per_step']['jacobi']))"
done; done
python tools/wct_tol_probe.py 2>/dev/null | tail -1

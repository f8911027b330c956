#!/bin/bash
# Round 4, GPU box: A-B of the pair-problem kernels.  WCT_JACOBI_R4 = 0 (round 3: LDS image, 1024 threads) | 1 (registers,
# 256 threads, 1 x W strips) | 2 (registers, 2 x 2 patches).
# usage (gpurun): bash tools/r04_ab.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04_ab.txt
: > $OUT
for R4 in 1 2 0; do
  export WCT_JACOBI_R4=$R4
  echo "=== WCT_JACOBI_R4=$R4" >> $OUT
  ( timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "eigh or wct_matches or hard_512 or straddle or tf_mode" 2>&1 | tail -4 ) >> $OUT
  for B in 32 8 1; do
    timeout 300 python bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2> gpurun_out/r04_ab_err_${R4}_${B}.txt | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch %3d: %.1f frames/s, %.2f ms/step; jacobi %.2f ms, conv3x3 %.2f, apply %.2f, cov %.2f; sweeps %s' % (l['config']['global_batch'], l['value'], l['ms_per_step'], l['breakdown_ms_per_step']['jacobi'], l['breakdown_ms_per_step']['conv3x3'], l['breakdown_ms_per_step']['wct_apply'], l['breakdown_ms_per_step']['wct_cov'], {k:v['mean'] for k,v in l['eigensolver']['sweeps'].items()}))" >> $OUT 2>&1
  done
done
unset WCT_JACOBI_R4
bash tools/r04_ts.sh >> $OUT 2>&1
cat $OUT

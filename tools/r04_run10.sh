#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04_run10.txt
: > $OUT
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py -x -q -k "wct or swap or cutoff or straddle" 2>&1 | tail -4 ) >> $OUT
( timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -k "config3 or batch32 or fused_pipeline or swap5 or shared_style" 2>&1 | tail -3 ) >> $OUT
for B in 32 8 1; do
  timeout 300 python bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch %3d: %.1f frames/s, %.2f ms/step; jacobi %.2f ms, conv3x3 %.2f, apply %.2f, cov %.2f' % (l['config']['global_batch'], l['value'], l['ms_per_step'], l['breakdown_ms_per_step']['jacobi'], l['breakdown_ms_per_step']['conv3x3'], l['breakdown_ms_per_step']['wct_apply'], l['breakdown_ms_per_step']['wct_cov']))" >> $OUT 2>&1
done
cat $OUT

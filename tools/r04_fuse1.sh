#!/bin/bash
# Round 4, GPU box: conv1_1 inside conv1_2's patch loader -- bit parity with the two launches, then the step at batch 32 / 8 / 1
# (WCT_FUSE_CONV1=0 beside it).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04_fuse1.txt
: > $OUT
( timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ops.py -x -q -k "conv1_1_inside or encode or conv3x3 or pool or epilogue_statistics" 2>&1 | tail -6 ) >> $OUT
for F in 1 0; do
for B in 32 8 1; do
  WCT_FUSE_CONV1=$F timeout 300 python bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2> gpurun_out/r04_fuse1_err_${F}_${B}.txt | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
b=l['breakdown_ms_per_step']
print('fuse $F batch %3d: %.1f frames/s, %.2f ms/step; conv3x3 %.2f (frac %.3f), conv_first %.2f, jacobi %.2f, apply %.2f, cov %.2f' % (l['config']['global_batch'], l['value'], l['ms_per_step'], b['conv3x3'], l['roofline']['frac'], b.get('conv_first', -1), b['jacobi'], b['wct_apply'], b['wct_cov']))" >> $OUT 2>&1
done
done
cat $OUT

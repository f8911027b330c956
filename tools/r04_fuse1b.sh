#!/bin/bash
# parity test of conv1_1-in-the-loader, phase timing, step at batch 32 / 8 / 1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04_fuse1b.txt
: > $OUT
( timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -k "conv1_1_inside" 2>&1 | tail -4 ) >> $OUT
bash tools/r04_fuse1_ts.sh > /dev/null 2>&1
head -4 gpurun_out/r04_fuse1_ts_1.txt >> $OUT
for B in 32 8 1; do
  timeout 300 python bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2> gpurun_out/r04_fuse1_err_${B}.txt | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
b=l['breakdown_ms_per_step']
print('batch %3d: %.1f frames/s, %.2f ms/step; conv3x3 %.2f (frac %.3f), conv_first %.2f, jacobi %.2f, apply %.2f, cov %.2f' % (l['config']['global_batch'], l['value'], l['ms_per_step'], b['conv3x3'], l['roofline']['frac'], b.get('conv_first', -1), b['jacobi'], b['wct_apply'], b['wct_cov']))" >> $OUT 2>&1
done
cat $OUT

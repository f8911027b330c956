#!/bin/bash
# Round 4, GPU box: eigensolver tests + bench at batch 32 / 8 / 1 of the build as it is (no tuning switches).
# usage (gpurun): bash tools/r04_occ.sh <tag>
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-occ}
OUT=gpurun_out/r04_${TAG}.txt
: > $OUT
( timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "eigh or wct_matches or hard_512 or straddle or tf_mode or cutoff" 2>&1 | tail -4 ) >> $OUT
for B in 32 8 1; do
  timeout 300 python bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2> gpurun_out/r04_${TAG}_err_${B}.txt | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch %3d: %.1f frames/s, %.2f ms/step; jacobi %.2f ms, conv3x3 %.2f, apply %.2f, cov %.2f; sweeps %s' % (l['config']['global_batch'], l['value'], l['ms_per_step'], l['breakdown_ms_per_step']['jacobi'], l['breakdown_ms_per_step']['conv3x3'], l['breakdown_ms_per_step']['wct_apply'], l['breakdown_ms_per_step']['wct_cov'], {k:v['mean'] for k,v in l['eigensolver']['sweeps'].items()}))" >> $OUT 2>&1
done
cat $OUT

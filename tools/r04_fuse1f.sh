#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04_fuse1f.txt
: > $OUT
( timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -k "conv1_1_inside" 2>&1 | tail -3 ) >> $OUT
bash tools/gpu_prof_stats.sh r04_fuse1f > /dev/null 2>&1
grep "conv" gpurun_out/r04_fuse1f_kernel_stats.csv | cut -c1-150 >> $OUT
bash tools/r04_fuse1_ts.sh 2>&1 | grep -A1 "second patch" | head -2 >> $OUT
cat $OUT

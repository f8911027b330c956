#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04_run2.txt
: > $OUT
for R4 in 1 0; do TAG="R4=$R4" WCT_JACOBI_R4=$R4 timeout 300 python tools/r04_probe.py >> $OUT 2>&1; done
bash tools/r04_ts.sh >> $OUT 2>&1
bash tools/r03_trace.sh r04_a "1 32" >> $OUT 2>&1
cat $OUT

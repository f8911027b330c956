#!/bin/bash
# usage (gpurun): bash tools/gpu_fuzz_wide.sh <scale> [pytest -k expression]  -- the hypothesis sweeps of tests/test_gpu_fuzz.py with <scale> times the
# examples, drawn at random (not the fixed sequence of the suite); every failing example is reported as drawn.  Output: gpurun_out/fuzz_wide.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
WCT_FUZZ_SCALE=$1 timeout ${3:-1500} python -m pytest tests/test_gpu_fuzz.py -q -m gpu -s ${2:+-k "$2"} 2>&1 | grep -E "^wct case|wide band: vs|near cut-off|^FUZZ-FAIL|^E   |Falsifying|passed|failed|^FAILED|WCT sweep|    [a-z_]+=" | cut -c1-400 > gpurun_out/fuzz_wide.txt
grep -E "^FUZZ-FAIL|^E   |Falsifying|passed|failed|^FAILED|WCT sweep|    [a-z_]+=" gpurun_out/fuzz_wide.txt | head -80

#!/bin/bash
# vstrip kernel alone (pair problems and tile updates exit at once: WCT_JACOBI_DBG=3), durations from a kernel trace
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/vs_alone
WCT_JACOBI_DBG=${1:-3} WCT_JACOBI_MAX_SWEEPS=4 WCT_JACOBI_MID=-1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/vs_alone -o t -- python $GRAFT_REPO_ROOT/tools/r03_eig_time.py 64 > /tmp/vs_alone.log 2>&1
f=$(find /tmp/vs_alone -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/trace_summary.py "$f" | head -6

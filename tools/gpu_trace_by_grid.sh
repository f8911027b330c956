#!/bin/bash
# Per-launch durations of one bench step, grouped by (kernel, grid): rocprofv3 --kernel-trace as CSV + tools/trace_by_grid.py.
# usage (gpurun): bash tools/gpu_trace_by_grid.sh <out-name> [batch] [substring filter ...]
NAME=$1; BATCH=${2:-32}; shift 2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out && cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tbg -o tbg -- python $R/bench.py --batch $BATCH --no-cpu-baseline --no-latency --no-prof --steps 2 --warmup 1 > /tmp/tbg.log 2>&1
F=$(find /tmp/tbg -name '*kernel_trace.csv' | head -1)
[ -z "$F" ] && { tail -20 /tmp/tbg.log; exit 1; }
( echo "# rocprofv3 --kernel-trace -- python bench.py --batch $BATCH --no-prof --steps 2 --warmup 1 (3 steps): launches by (kernel, grid in blocks)"; python $R/tools/trace_by_grid.py $F "$@" ) > $R/gpurun_out/$NAME.txt
cat $R/gpurun_out/$NAME.txt | cut -c1-140

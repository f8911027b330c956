#!/bin/bash
# Kernel trace of a bench run at one batch size, summarised by tools/trace_summary.py (per-kernel time and the gaps between consecutive
# eigensolver kernels on a queue).  usage (gpurun): bash tools/gpu_trace_summary.sh <out-name> [batch]
NAME=$1; BATCH=${2:-1}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out && cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tsum -o tsum -- python $R/bench.py --batch $BATCH --no-cpu-baseline --no-latency --no-prof --steps 2 --warmup 1 > /tmp/tsum.log 2>&1
F=$(find /tmp/tsum -name '*kernel_trace.csv' | head -1)
[ -z "$F" ] && { tail -20 /tmp/tsum.log; exit 1; }
( echo "# rocprofv3 --kernel-trace -- python bench.py --batch $BATCH --no-prof --steps 2 --warmup 1 (3 steps)"; python $R/tools/trace_summary.py $F ) > $R/gpurun_out/$NAME.txt
cut -c1-170 $R/gpurun_out/$NAME.txt

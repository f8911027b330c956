#!/bin/bash
# rocprofv3 kernel stats of the default bench workload -> gpurun_out/<tag>_kernel_stats.csv   usage: gpu_prof_stats.sh <tag> [bench args]
export TMPDIR=/tmp
TAG=$1; shift
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-prof "$@" > /tmp/prof_$TAG.log 2>&1
f=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv && head -30 "$f" | cut -c1-200
tail -2 /tmp/prof_$TAG.log | cut -c1-300

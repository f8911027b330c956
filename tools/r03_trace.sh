#!/bin/bash
# kernel traces (start/end timestamps) of the bench workload at several batch sizes -> gpurun_out/<tag>_trace_b<B>.csv.gz
# usage (gpurun): bash tools/r03_trace.sh <tag> "1 8 32"
export TMPDIR=/tmp
TAG=$1; BATCHES=${2:-"1 8 32"}
cd /tmp
for b in $BATCHES; do
  rm -rf /tmp/tr_$b
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$b -o tr -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-prof > /tmp/tr_$b.log 2>&1
  f=$(find /tmp/tr_$b -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/trace_summary.py "$f" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace_b$b.txt && gzip -c "$f" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace_b$b.csv.gz
  tail -1 /tmp/tr_$b.log | cut -c1-200
done

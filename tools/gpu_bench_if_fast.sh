#!/bin/bash
# The boxes gpurun hands out fall into two populations (the unchanged conv3x3 <16,128,2,2> averages ~0.414 ms per launch on one
# and ~0.428 on the other).  This probes the box with a short bench and, only on the faster population, records the default bench line and
# the rocprofv3 kernel stats under <tag>_*.   usage (gpurun): bash tools/gpu_bench_if_fast.sh <tag> [threshold_ms]
TAG=$1; THR=${2:-0.419}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
A=$(python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline']['avg_launch_ms'])")
echo "probe: conv3x3 average launch $A ms (threshold $THR)" | tee $O/${TAG}_probe.txt
python -c "import sys; sys.exit(0 if float('$A') < float('$THR') else 1)" || { echo "slower population: nothing recorded"; exit 0; }
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cut -c1-260 $O/${TAG}_bench.json
W=/tmp/prof_$TAG; mkdir -p $W/db; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $W/raw_stats -o stats -- python $R/bench.py --no-cpu-baseline --no-latency --steps 2 --warmup 1 > $W/stats.log 2>&1
F=$(find $W/raw_stats -name '*results.db' | head -1); [ -n "$F" ] && cp $F $W/db/stats_results.db
python - <<PY
import sqlite3
rows=list(sqlite3.connect('$W/db/stats_results.db').cursor().execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
with open('$O/${TAG}_kernel_stats.csv','w') as f:
    f.write('# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline (batch 32 x 512x512, 5-level); durations in us\n')
    f.write('name,calls,total_us,avg_us,percent\n')
    for r in rows: f.write('"%s",%d,%.3f,%.3f,%.3f\n' % r)
PY
head -8 $O/${TAG}_kernel_stats.csv | cut -c1-140
cd $R; python bench.py --no-cpu-baseline --no-latency --steps 5 --warmup 2 > $O/${TAG}_bench_noprofiler.json 2>/dev/null; grep -o '"value": [0-9.]*' $O/${TAG}_bench_noprofiler.json | head -2

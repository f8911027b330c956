#!/bin/bash
# Round profile on the GPU box: kernel stats, HBM counters (FETCH_SIZE / WRITE_SIZE in separate passes), MFMA-pipe
# utilisation, and the counter calibration on known byte counts.  Summaries land in gpurun_out/ as <tag>_*.
# usage: bash tools/gpu_profile.sh <tag>
export TMPDIR=/tmp
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
W=/tmp/prof_$TAG
mkdir -p $W/db && cd /tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-latency"
run() {   # name, rocprof args..., -- command
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace "$@" > $W/$name.log 2>&1
  local f=$(find $W/raw_$name -name '*results.db' | head -1)
  [ -n "$f" ] && cp "$f" $W/db/${name}_results.db || { echo "no db for $name"; tail -5 $W/$name.log; }
}
run stats --stats -d $W/raw_stats -o stats -- $BENCH --steps 2 --warmup 1
run fetch --pmc FETCH_SIZE -d $W/raw_fetch -o fetch -- $BENCH --steps 1 --warmup 1 --no-prof
run write --pmc WRITE_SIZE -d $W/raw_write -o write -- $BENCH --steps 1 --warmup 1 --no-prof
run mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $W/raw_mfma -o mfma -- $BENCH --steps 1 --warmup 1 --no-prof
run calib_fetch --pmc FETCH_SIZE -d $W/raw_calib_fetch -o calib_fetch -- $R/tools/probe/fetch_calib
run calib_write --pmc WRITE_SIZE -d $W/raw_calib_write -o calib_write -- $R/tools/probe/fetch_calib
ls -la $W/db
python $R/tools/summarize_prof.py $W/db $TAG 32 $O
python $R/tools/pmc_mfma_summary.py $W/db/mfma_results.db $O/${TAG}_pmc_mfma_util.csv "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -- python bench.py --steps 1 --warmup 1 --no-prof (batch 32 x 512x512)"
python $R/tools/fetch_calib_summary.py $W/db/calib_fetch_results.db $W/db/calib_write_results.db $O/${TAG}_pmc_calibration.txt
$BENCH --steps 5 --warmup 2 > $O/${TAG}_bench_noprofiler.json 2>/dev/null
grep -o '"value": [0-9.]*' $O/${TAG}_bench_noprofiler.json

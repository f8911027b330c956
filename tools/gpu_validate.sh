#!/bin/bash
# Round validation on the GPU box.  usage (gpurun): bash tools/gpu_validate.sh <tag> [full|tests|bench|profile]
#   tests    the full GPU test suite (tail of the log + rc) and smoke()
#   bench    the default bench line, batch sweep 1..32, the other BASELINE configurations, the 2-rank dry run on one GPU
#   profile  rocprofv3 kernel stats + the PMC passes (HBM, MFMA utilisation, LDS / issue), each in its own run
#   full     all of the above (default)
# Everything lands in gpurun_out/<tag>_*; copy what is to be judged into profiles/.
TAG=${1:-r05_final}; WHAT=${2:-full}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ $WHAT = full ] || [ $WHAT = tests ]; then
  ( python -m pytest tests/ -q -m gpu --durations=15 2>&1 | tail -60; echo "pytest rc=${PIPESTATUS[0]}" ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1
  tail -3 gpurun_out/${TAG}_pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
fi
if [ $WHAT = full ] || [ $WHAT = bench ]; then
  python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
  cut -c1-300 gpurun_out/${TAG}_bench.json
  ( for B in 1 2 4 8 16 32; do python bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
b=l['breakdown_ms_per_step']
print('batch %3d: %7.1f frames/s  %6.2f ms/step | conv3x3 %5.2f conv12 %4.2f conv_first %4.2f conv_last %4.2f cov %4.2f jacobi %5.2f apply %4.2f' % (l['config']['global_batch'], l['value'], l['ms_per_step'], b['conv3x3'], b.get('conv12', 0), b['conv_first'], b['conv_last'], b['wct_cov'], b['jacobi'], b['wct_apply']))"; done ) > gpurun_out/${TAG}_batch_sweep.txt 2>&1
  cat gpurun_out/${TAG}_batch_sweep.txt
  python tools/bench_configs.py 10 > gpurun_out/${TAG}_configs_latency.txt 2>&1; cat gpurun_out/${TAG}_configs_latency.txt
  WCT_BENCH_BACKEND=gloo WCT_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_dryrun_2ranks_1gpu_gloo.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_dryrun_2ranks_1gpu_gloo.json
fi
if [ $WHAT = full ] || [ $WHAT = profile ]; then
  bash tools/gpu_profile.sh $TAG 2>&1 | tail -12
  bash tools/gpu_pmc_lds.sh $TAG 2>&1 | tail -8
fi

#!/bin/bash
# Per-layer conv table and batch sweep of the final build -> gpurun_out/<tag>_conv_layers_b8.txt, <tag>_batch_sweep.txt
TAG=${1:-r03_final}
cd $GRAFT_REPO_ROOT
( echo "# python tools/bench_conv.py 8  (per-layer HIP-event timing of the conv3x3 kernel, fp32-output variant, batch 8 emulated by an 8x taller image)"; python tools/bench_conv.py 8 2>/dev/null ) > gpurun_out/${TAG}_conv_layers_b8.txt
( echo "# python bench.py --batch B --steps 5 --warmup 2 --no-cpu-baseline --no-latency   (device-resident pairs per step; per-class event timing on)"
  for b in 1 2 4 8 16 32; do
    python bench.py --batch $b --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('batch %2d: %.1f frames/s, %.2f ms/step, eigensolver %.2f ms, conv3x3 %.2f ms' % (d['config']['pairs_per_gpu_per_step'], d['value'], d['ms_per_step'], d['breakdown_ms_per_step']['jacobi'], d['breakdown_ms_per_step']['conv3x3']))
"
  done
  echo "# strong scaling as BASELINE configs[3] states it (64 frames): one GPU = 2 steps of 32; eight GPUs = one step of 8 each"
) > gpurun_out/${TAG}_batch_sweep.txt
cat gpurun_out/${TAG}_batch_sweep.txt; tail -3 gpurun_out/${TAG}_conv_layers_b8.txt

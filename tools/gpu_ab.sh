#!/bin/bash
# Same-box A-B on the GPU box: one environment switch, several values, the bench at several batch sizes per value
# (per-class HIP-event breakdown), optionally a pytest subset per value first.  The boxes gpurun hands out differ by +-2.5 %
# on the conv class, so two builds / two switch values are only ever compared inside ONE call.
#
# usage (gpurun):  bash tools/gpu_ab.sh <out-name> <ENV_VAR> "<value> <value> ..." ["<batch> <batch> ..."] ["<pytest -k expr>"]
#   e.g.           bash tools/gpu_ab.sh r05_conv_lds WCT_CONV_LDSB "0 1" "32 8" "conv3x3 or every_conv_layer"
# Switches other than the three test hooks (csrc/common.h) need a tuning build: WCT_BUILD_TUNING=1 python -m wct_tf_amd.build --force
# (run it HERE, before the bench: the .so that travelled is the product build).  <ENV_VAR> = NONE runs the plain build once.
NAME=$1; VAR=$2; VALUES=${3:-x}; BATCHES=${4:-"32 8 1"}; KEXPR=$5
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/${NAME}.txt
: > $OUT
for V in $VALUES; do
  if [ "$VAR" != NONE ]; then export $VAR=$V; echo "=== $VAR=$V" >> $OUT; fi
  if [ -n "$KEXPR" ]; then ( timeout 900 python -m pytest tests -q -m gpu -k "$KEXPR" 2>&1 | tail -4 ) >> $OUT; fi
  for B in $BATCHES; do
    timeout 300 python bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2> gpurun_out/${NAME}_err_${V}_${B}.txt | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
b=l['breakdown_ms_per_step']
print('batch %3d: %7.1f frames/s %6.2f ms/step (no_prof %.2f) | conv3x3 %5.2f (%.3f of peak) conv12 %4.2f conv_first %4.2f conv_last %4.2f cov %4.2f jacobi %5.2f apply %4.2f | sweeps %s' % (
  l['config']['global_batch'], l['value'], l['ms_per_step'], l['no_prof']['ms_per_step'], b['conv3x3'], l['roofline']['frac'], b.get('conv12', 0), b['conv_first'], b['conv_last'],
  b['wct_cov'], b['jacobi'], b['wct_apply'], {k: round(v['mean'], 2) for k, v in l['eigensolver']['sweeps'].items()}))" >> $OUT 2>&1
  done
done
[ "$VAR" != NONE ] && unset $VAR
cat $OUT

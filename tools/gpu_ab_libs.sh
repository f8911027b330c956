#!/bin/bash
# Same-box A-B of several BUILDS of the library (wct_tf_amd/variants/<name>.so, made by tools/build_variant.sh): each variant is
# copied over libwct_hip.so in turn and the bench runs at the given batch sizes (per-class HIP-event breakdown).
# usage (gpurun): bash tools/gpu_ab_libs.sh <out-name> "<variant> <variant> ..." ["<batch> ..."] [ENV=VALUE ...applied to every run]
NAME=$1; VARIANTS=$2; BATCHES=${3:-"32 8 1"}; shift 3
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/${NAME}.txt
: > $OUT
cp wct_tf_amd/libwct_hip.so /tmp/libwct_hip.so.keep
for V in $VARIANTS; do
  LIBV=${V%%:*}; ENVV=""; [ "$V" != "$LIBV" ] && ENVV=${V#*:}
  cp wct_tf_amd/variants/$LIBV.so wct_tf_amd/libwct_hip.so
  echo "=== $V" >> $OUT
  for B in $BATCHES; do
    env $ENVV "$@" timeout 300 python bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2> gpurun_out/${NAME}_err_${LIBV}_${B}.txt | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
b=l['breakdown_ms_per_step']
print('batch %3d: %7.1f frames/s %6.2f ms/step (no_prof %.2f) | conv3x3 %5.2f (%.3f of peak) conv12 %4.2f conv_first %4.2f conv_last %4.2f cov %4.2f jacobi %5.2f apply %4.2f | sweeps %s | frames %s' % (
  l['config']['global_batch'], l['value'], l['ms_per_step'], l['no_prof']['ms_per_step'], b['conv3x3'], l['roofline']['frac'], b.get('conv12', 0), b['conv_first'], b['conv_last'],
  b['wct_cov'], b['jacobi'], b['wct_apply'], {k: round(v['mean'], 2) for k, v in l['eigensolver']['sweeps'].items()}, (l.get('frames_sha256') or '')[:12]))" >> $OUT 2>&1
  done
done
cp /tmp/libwct_hip.so.keep wct_tf_amd/libwct_hip.so
cat $OUT

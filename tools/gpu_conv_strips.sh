#!/bin/bash
# STRIPS variants of the generic conv loader (tuning build): bit-identity on tall images + per-layer table, WCT_CONV_STRIPS=0 | 1.
# usage (gpurun): bash tools/gpu_conv_strips.sh <out-name> <variant .so>
NAME=$1; LIBV=$2
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
OUT=gpurun_out/${NAME}.txt; : > $OUT
cp wct_tf_amd/libwct_hip.so /tmp/libwct_hip.so.keep
cp wct_tf_amd/variants/$LIBV.so wct_tf_amd/libwct_hip.so
for S in 0 1; do
  export WCT_CONV_STRIPS=$S
  echo "=== WCT_CONV_STRIPS=$S" >> $OUT
  CONV_CHECK_TALL=1 timeout 300 python tools/probe/conv_cfg_check.py 64 2>&1 | tail -9 >> $OUT
  timeout 300 python tools/bench_conv.py 8 128 2>&1 | tail -9 >> $OUT
done
unset WCT_CONV_STRIPS
cp /tmp/libwct_hip.so.keep wct_tf_amd/libwct_hip.so
cat $OUT

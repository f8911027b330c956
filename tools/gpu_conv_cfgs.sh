#!/bin/bash
# Conv tile configurations of a TUNING build on the GPU box: bit-identity digests (tools/probe/conv_cfg_check.py) and the per-layer
# table (tools/bench_conv.py, batch 8 emulated by an 8x taller image) for each "CFG:XCD" pair.
# usage (gpurun): bash tools/gpu_conv_cfgs.sh <out-name> <variant .so name> "0:0 6:0 ..." [batch]
NAME=$1; LIBV=$2; CFGS=$3; BATCH=${4:-8}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/${NAME}.txt
: > $OUT
cp wct_tf_amd/libwct_hip.so /tmp/libwct_hip.so.keep
cp wct_tf_amd/variants/$LIBV.so wct_tf_amd/libwct_hip.so
for C in $CFGS; do
  export WCT_CONV_CFG=${C%%:*} WCT_CONV_XCD=${C#*:}
  echo "=== WCT_CONV_CFG=$WCT_CONV_CFG WCT_CONV_XCD=$WCT_CONV_XCD" >> $OUT
  timeout 200 python tools/probe/conv_cfg_check.py 2>&1 | tail -12 >> $OUT
  timeout 300 python tools/bench_conv.py $BATCH 2>&1 | tail -17 >> $OUT
done
unset WCT_CONV_CFG WCT_CONV_XCD
cp /tmp/libwct_hip.so.keep wct_tf_amd/libwct_hip.so
cat $OUT

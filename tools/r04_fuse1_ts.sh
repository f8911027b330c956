#!/bin/bash
# Round 4, GPU box: phase timing (-DCONV_TS library built beforehand as wct_tf_amd/libwct_ts.so) of the conv launches of one
# 32-pair step, conv1_1 inside the loader on / off.  Build it here first (hipcc cross-compiles; the .so travels with gpurun):
#   cd wct_tf_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCONV_TS -c conv.hip -o /tmp/conv_ts.o &&
#   hipcc --offload-arch=gfx950 -shared -fPIC -o ../libwct_ts.so api.o /tmp/conv_ts.o wct.o coral.o train.o
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp wct_tf_amd/libwct_hip.so /tmp/lib_keep.so
cp wct_tf_amd/libwct_ts.so wct_tf_amd/libwct_hip.so
for F in 1 0; do
  WCT_FUSE_CONV1=$F timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-latency --no-prof > /tmp/ts_$F.log 2>&1
  grep "^TS" /tmp/ts_$F.log | grep -B1 "Cin 64 Cout 64 H 512 up 0" > gpurun_out/r04_fuse1_ts_$F.txt
done
cp /tmp/lib_keep.so wct_tf_amd/libwct_hip.so
head -30 gpurun_out/r04_fuse1_ts_1.txt; echo; head -12 gpurun_out/r04_fuse1_ts_0.txt

#!/bin/bash
# phase timing of the pair-problem blocks with the -DJACOBI_TS library built beforehand (wct_tf_amd/libwct_jts.so):
#   cd wct_tf_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DJACOBI_TS -c wct.hip -o /tmp/wct_ts.o &&
#   hipcc --offload-arch=gfx950 -shared -fPIC -o ../libwct_jts.so api.o conv.o /tmp/wct_ts.o coral.o train.o
cd $GRAFT_REPO_ROOT
cp wct_tf_amd/libwct_hip.so /tmp/libwct_hip.so.keep
cp wct_tf_amd/libwct_jts.so wct_tf_amd/libwct_hip.so
for n in 2 16 64; do
  WCT_JACOBI_MAX_SWEEPS=4 timeout 200 python tools/r03_eig_time.py $n 2>&1 | grep -E "jacobi_ts" | tail -2
done
cp /tmp/libwct_hip.so.keep wct_tf_amd/libwct_hip.so

#!/bin/bash
# phase timing of the pair-problem blocks (-DJACOBI_TS build of wct.hip into a scratch copy of the library), both kernels
cd $GRAFT_REPO_ROOT
cp wct_tf_amd/libwct_hip.so /tmp/libwct_hip.so.keep
cd wct_tf_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -fno-slp-vectorize -DJACOBI_TS -c wct.hip -o /tmp/wct_ts.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libwct_hip.so api.o conv.o /tmp/wct_ts.o coral.o train.o
cd ../..
for R4 in 1 2 0; do
  for n in 2 16 64; do
    WCT_JACOBI_R4=$R4 WCT_JACOBI_MAX_SWEEPS=4 WCT_JACOBI_MID=-1 timeout 200 python tools/r03_eig_time.py $n 2>&1 | grep -E "jacobi_ts" | tail -2 | sed "s/^/R4=$R4 /"
  done
done
cp /tmp/libwct_hip.so.keep wct_tf_amd/libwct_hip.so

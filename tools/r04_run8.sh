#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04_run8.txt
: > $OUT
cp wct_tf_amd/libwct_hip.so /tmp/libwct_hip.so.keep
cd wct_tf_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -fno-slp-vectorize -DJACOBI_TS -c wct.hip -o /tmp/wct_ts.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libwct_hip.so api.o conv.o /tmp/wct_ts.o coral.o train.o
cd ../..
for n in 2 64; do
  WCT_TS_INTRA=1 WCT_JACOBI_MAX_SWEEPS=8 WCT_JACOBI_MID=-1 timeout 200 python tools/r03_eig_time.py $n 2>&1 | grep -E "jacobi_ts" | tail -3 >> $OUT
done
cp /tmp/libwct_hip.so.keep wct_tf_amd/libwct_hip.so
cat $OUT

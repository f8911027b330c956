#!/bin/bash
# Phase timing of the pair-problem blocks of jacobi_fused4_kernel: a -DJACOBI_TS build of wct.hip (lane 0 of every block stamps
# s_memtime at its phase boundaries, every launch is synchronised and summarised by the launcher) linked into a scratch copy of
# the library; wct_eigh at a fixed sweep count on 2 / 16 / 64 matrices of 512 channels (= batch 1 / 8 / 32 of the pipeline).
# usage (gpurun): bash tools/gpu_jacobi_ts.sh <out-name>
NAME=${1:-r05_jacobi_ts}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp wct_tf_amd/libwct_hip.so /tmp/libwct_hip.so.keep
( cd wct_tf_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -fno-slp-vectorize -DJACOBI_TS -c wct.hip -o /tmp/wct_ts.o 2>/dev/null &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libwct_hip.so api.o conv.o /tmp/wct_ts.o coral.o train.o )
for n in 2 16 64; do
  WCT_JACOBI_MAX_SWEEPS=4 timeout 200 python tools/probe/r03_eig_time.py $n 2>&1 | grep -E "jacobi_ts|C=512" | tail -4
done > gpurun_out/${NAME}.txt 2>&1
cp /tmp/libwct_hip.so.keep wct_tf_amd/libwct_hip.so
cat gpurun_out/${NAME}.txt

#!/bin/bash
# Phase timing of the conv kernel on the GPU box: builds a -DCONV_TS variant of conv.hip (s_memtime stamps per block), swaps it
# in for one bench step and prints the per-launch means.  usage (gpurun): bash tools/conv_phase_timing.sh
cd $GRAFT_REPO_ROOT/wct_tf_amd/csrc || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -DCONV_TS -c conv.hip -o /tmp/conv_ts.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libwct_ts.so api.o /tmp/conv_ts.o wct.o coral.o train.o || exit 1
cd $GRAFT_REPO_ROOT
cp wct_tf_amd/libwct_hip.so /tmp/lib_keep.so
cp /tmp/libwct_ts.so wct_tf_amd/libwct_hip.so
timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-latency --no-prof > /tmp/ts.log 2>&1
cp /tmp/lib_keep.so wct_tf_amd/libwct_hip.so
grep "^TS" /tmp/ts.log > gpurun_out/conv_phase_timing.txt
wc -l gpurun_out/conv_phase_timing.txt

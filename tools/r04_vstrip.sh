#!/bin/bash
# the V pass alone (tuning build, WCT_JACOBI_DBG=3: pair problems and tile updates exit at once) and in situ: launch durations
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp wct_tf_amd/libwct_hip.so /tmp/libwct_hip.so.keep
cd wct_tf_amd/csrc
for f in api conv wct coral train; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $( [ $f = wct ] && echo -fno-slp-vectorize ) -DWCT_TUNING -c $f.hip -o /tmp/t_$f.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libwct_hip.so /tmp/t_api.o /tmp/t_conv.o /tmp/t_wct.o /tmp/t_coral.o /tmp/t_train.o
cd /tmp
for DBG in 3 0; do
rm -rf /tmp/vs_alone
WCT_JACOBI_DBG=$DBG WCT_JACOBI_MAX_SWEEPS=4 WCT_JACOBI_MID=-1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/vs_alone -o t -- python $GRAFT_REPO_ROOT/tools/r03_eig_time.py 64 > /tmp/vs_alone.log 2>&1
f=$(find /tmp/vs_alone -name '*kernel_trace.csv' | head -1)
echo "WCT_JACOBI_DBG=$DBG"; python $GRAFT_REPO_ROOT/tools/trace_summary.py "$f" | head -8
done
cp /tmp/libwct_hip.so.keep $GRAFT_REPO_ROOT/wct_tf_amd/libwct_hip.so

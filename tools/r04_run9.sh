#!/bin/bash
# V pass placement: side stream (default) | in the solver's stream (WCT_JACOBI_VSTRIP_INLINE=1) | inside the update tasks (WCT_JACOBI_VSTRIP=0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04_run9.txt
: > $OUT
cp wct_tf_amd/libwct_hip.so /tmp/libwct_hip.so.keep
cd wct_tf_amd/csrc
for f in api conv wct coral train; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $( [ $f = wct ] && echo -fno-slp-vectorize ) -DWCT_TUNING -c $f.hip -o /tmp/t_$f.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libwct_hip.so /tmp/t_api.o /tmp/t_conv.o /tmp/t_wct.o /tmp/t_coral.o /tmp/t_train.o
cd ../..
run() {
for B in 32 8; do
  env "$@" timeout 300 python bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$* batch %3d: %.1f frames/s, %.2f ms/step; jacobi %.2f ms, conv3x3 %.2f, apply %.2f, cov %.2f' % (l['config']['global_batch'], l['value'], l['ms_per_step'], l['breakdown_ms_per_step']['jacobi'], l['breakdown_ms_per_step']['conv3x3'], l['breakdown_ms_per_step']['wct_apply'], l['breakdown_ms_per_step']['wct_cov']))" >> $OUT 2>&1
done
}
run X=0




cp /tmp/libwct_hip.so.keep wct_tf_amd/libwct_hip.so
cat $OUT
( timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "eigh" 2>&1 | tail -2 )
bash tools/r03_trace.sh r04_c "32" 2>&1 | tail -1
head -14 gpurun_out/r04_c_trace_b32.txt

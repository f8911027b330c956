#!/bin/bash
# strips over 8 waves (WCT_JACOBI_R4=1, default) vs 4 waves (=2) vs the round-3 kernel (=0): tuning build (-DWCT_TUNING)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04_run6.txt
: > $OUT
cp wct_tf_amd/libwct_hip.so /tmp/libwct_hip.so.keep
cd wct_tf_amd/csrc
for f in api conv wct coral train; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $( [ $f = wct ] && echo -fno-slp-vectorize ) -DWCT_TUNING -c $f.hip -o /tmp/t_$f.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libwct_hip.so /tmp/t_api.o /tmp/t_conv.o /tmp/t_wct.o /tmp/t_coral.o /tmp/t_train.o
cd ../..
( WCT_JACOBI_R4=1 timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "eigh or wct_matches or hard_512 or straddle or tf_mode or loud or cutoff" 2>&1 | tail -3 ) >> $OUT
for R4 in 1 2 0; do
for B in 32 8 1; do
  WCT_JACOBI_R4=$R4 timeout 300 python bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('R4=$R4 batch %3d: %.1f frames/s, %.2f ms/step; jacobi %.2f ms, conv3x3 %.2f, apply %.2f, cov %.2f; sweeps %s' % (l['config']['global_batch'], l['value'], l['ms_per_step'], l['breakdown_ms_per_step']['jacobi'], l['breakdown_ms_per_step']['conv3x3'], l['breakdown_ms_per_step']['wct_apply'], l['breakdown_ms_per_step']['wct_cov'], {k:v['mean'] for k,v in l['eigensolver']['sweeps'].items()}))" >> $OUT 2>&1
done
done
cd wct_tf_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -fno-slp-vectorize -DWCT_TUNING -DJACOBI_TS -c wct.hip -o /tmp/wct_ts.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libwct_hip.so /tmp/t_api.o /tmp/t_conv.o /tmp/wct_ts.o /tmp/t_coral.o /tmp/t_train.o
cd ../..
for R4 in 1 2; do
for n in 2 64; do
  WCT_JACOBI_R4=$R4 WCT_JACOBI_MAX_SWEEPS=4 WCT_JACOBI_MID=-1 timeout 200 python tools/r03_eig_time.py $n 2>&1 | grep -E "jacobi_ts" | tail -2 | sed "s/^/R4=$R4 /" >> $OUT
done
done
cp /tmp/libwct_hip.so.keep wct_tf_amd/libwct_hip.so
cat $OUT

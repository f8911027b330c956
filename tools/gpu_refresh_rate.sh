cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp wct_tf_amd/libwct_hip.so /tmp/keep.so
cp wct_tf_amd/variants/fin_t.so wct_tf_amd/libwct_hip.so
WCT_REFRESH_STATS=1 timeout 600 python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-latency > /dev/null 2> gpurun_out/r06_refresh_rate_synth.err
grep "refresh fired" gpurun_out/r06_refresh_rate_synth.err | tail -1 > gpurun_out/r06_refresh_rate.txt
cp /tmp/keep.so wct_tf_amd/libwct_hip.so
cat gpurun_out/r06_refresh_rate.txt
bash tools/gpu_ab_libs.sh r06_lgkm_ab "pro fin pro fin" "32" > /dev/null 2>&1; cat gpurun_out/r06_lgkm_ab.txt

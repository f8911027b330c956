#!/bin/bash
# usage (gpurun): bash tools/gpu_fuzz_case.sh "<variant>[:ENV=VALUE] ..." [sweep]  -- tools/probe/r06_fuzz_case.py (and, with `sweep`, the WCT fuzz sweep)
# with each library build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp wct_tf_amd/libwct_hip.so /tmp/libwct_hip.so.keep
OUT=gpurun_out/r06_fuzz_case.txt; : > $OUT
for V in $1; do
  LIBV=${V%%:*}; ENVV=""; [ "$V" != "$LIBV" ] && ENVV=${V#*:}
  cp wct_tf_amd/variants/$LIBV.so wct_tf_amd/libwct_hip.so
  echo "=== $V" >> $OUT
  env $ENVV timeout 600 python ${PROBE:-tools/probe/r06_fuzz_case.py} 2>&1 | grep -v Warn >> $OUT
  [ "$2" = sweep ] && env $ENVV timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -s -k "wct_random" 2>&1 | grep -E "^wct case|wide band: vs|^E   |passed|failed" | cut -c1-250 >> $OUT
done
cp /tmp/libwct_hip.so.keep wct_tf_amd/libwct_hip.so
cat $OUT

#!/bin/bash
# PMC passes of ONE conv layer through the reduced-FLOP kernel (algo 2, tile shapes WCT_WINO_CFG of a TUNING build) or the direct one
# (cfg "d"): MFMA-pipe utilisation, wait / issue fractions, LDS counters, HBM fetch bytes -- each set in its own rocprofv3 run.
# usage (gpurun): bash tools/gpu_wino_pmc.sh <out-name> <variant .so> "<cfg> <cfg> ..." cin cout H up pool batch
NAME=$1; LIBV=$2; CFGS=$3; shift 3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
cp wct_tf_amd/libwct_hip.so /tmp/libwct_hip.so.keep
cp wct_tf_amd/variants/$LIBV.so wct_tf_amd/libwct_hip.so
OUT=$R/gpurun_out/${NAME}.txt
echo "# python tools/bench_wino_one.py $* <algo> under rocprofv3 --kernel-trace --pmc <set>; per configuration WCT_WINO_CFG (d = the direct kernel; tuning build $LIBV)" > $OUT
cd /tmp
for C in $CFGS; do
  ALGO=2; [ "$C" = d ] && ALGO=1
  [ "$C" != d ] && export WCT_WINO_CFG=$C
  for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    rm -rf /tmp/cpmc; timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/cpmc -o c -- python $R/tools/bench_wino_one.py "$@" $ALGO > /tmp/cpmc.log 2>&1
    f=$(find /tmp/cpmc -name '*results.db' | head -1)
    python3 - "$f" "$C" >> $OUT <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
d = {}
for k, c, n, v in cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
    if 'conv3x3_' in k: d.setdefault(k, {})[c] = (n, v)
for k, v in d.items():
    n = max(x[0] for x in v.values())
    print('CFG=%s %s launches=%d  ' % (sys.argv[2], k[k.index('<'):k.index('>') + 1], n) + '  '.join('%s=%.4g' % (c, x[1] / x[0]) for c, x in sorted(v.items())))
PY
  done
  unset WCT_WINO_CFG
done
cd $R; cp /tmp/libwct_hip.so.keep wct_tf_amd/libwct_hip.so
cat $OUT

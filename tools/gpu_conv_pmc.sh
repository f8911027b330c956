#!/bin/bash
# PMC passes of ONE conv layer under several tile configurations of a TUNING build (tools/bench_conv_one.py): MFMA-pipe utilisation and
# LDS / issue counters (each set in its own rocprofv3 run, --kernel-trace only) and the HBM fetch bytes.
# usage (gpurun): bash tools/gpu_conv_pmc.sh <out-name> <variant .so> "<cfg> <cfg> ..." cin cout H up batch
NAME=$1; LIBV=$2; CFGS=$3; shift 3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
cp wct_tf_amd/libwct_hip.so /tmp/libwct_hip.so.keep
cp wct_tf_amd/variants/$LIBV.so wct_tf_amd/libwct_hip.so
OUT=$R/gpurun_out/${NAME}.txt
echo "# python tools/bench_conv_one.py $* under rocprofv3 --kernel-trace --pmc <set>; per configuration WCT_CONV_CFG (tuning build $LIBV)" > $OUT
cd /tmp
for C in $CFGS; do
  export WCT_CONV_CFG=$C
  for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE"; do
    rm -rf /tmp/cpmc; timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/cpmc -o c -- python $R/tools/bench_conv_one.py "$@" > /tmp/cpmc.log 2>&1
    f=$(find /tmp/cpmc -name '*results.db' | head -1)
    python3 - "$f" "$C" >> $OUT <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
d = {}
for k, c, n, v in cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
    if 'conv3x3_mfma' in k: d.setdefault(k, {})[c] = (n, v)
for k, v in d.items():
    n = max(x[0] for x in v.values())
    print('CFG=%s %s launches=%d  ' % (sys.argv[2], k[k.index('<'):k.index('>') + 1], n) + '  '.join('%s=%.4g' % (c, x[1] / x[0]) for c, x in sorted(v.items())))
PY
  done
done
unset WCT_CONV_CFG
cd $R; cp /tmp/libwct_hip.so.keep wct_tf_amd/libwct_hip.so
cat $OUT

#!/bin/bash
# LDS / issue-stall counters per kernel (one PMC pass): SQ_LDS_IDX_ACTIVE (LDS-array cycles), SQ_LDS_BANK_CONFLICT, SQ_INSTS_LDS,
# SQ_INSTS_VALU, SQ_WAIT_INST_ANY, SQ_WAIT_INST_LDS, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES -> gpurun_out/<tag>_pmc_lds.csv
export TMPDIR=/tmp
TAG=${1:-r03_final}
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_lds -o lds -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-latency --no-prof > /tmp/pmc_lds.log 2>&1
f=$(find /tmp/pmc_lds -name '*results.db' | head -1)
python3 - "$f" "$R/gpurun_out/${TAG}_pmc_lds.csv" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
d = {}
for k, c, n, v in cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
    d.setdefault(k, {})[c] = (n, v)
rows = []
for k, v in d.items():
    act = v.get('GRBM_GUI_ACTIVE', (0, 0))[1]
    if act <= 0: continue
    g = lambda c: v.get(c, (0, 0))[1]
    wc = max(g('SQ_WAVE_CYCLES'), 1)
    cu_cycles = act / 8 * 256          # GRBM_GUI_ACTIVE summed over 8 XCDs -> per-XCD cycles x 256 CUs
    rows.append((act, k, v['GRBM_GUI_ACTIVE'][0], g('SQ_LDS_IDX_ACTIVE') / cu_cycles, g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1),
                 g('SQ_INSTS_LDS'), g('SQ_INSTS_VALU'), g('SQ_WAIT_INST_ANY') / wc, g('SQ_WAIT_INST_LDS') / wc))
rows.sort(reverse=True)
with open(sys.argv[2], 'w') as f:
    f.write('# rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- python bench.py --steps 1 --warmup 1 --no-prof (batch 32 x 512x512)\n')
    f.write('# lds_array_busy = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE/8 x 256 CUs); conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; stall fractions of SQ_WAVE_CYCLES\n')
    f.write('kernel,launches,gpu_active_cycles,lds_array_busy,conflict_frac,insts_lds,insts_valu,issue_stall_frac,lds_issue_stall_frac\n')
    for r in rows[:14]:
        f.write('"%s",%d,%.4g,%.3f,%.3f,%.4g,%.4g,%.3f,%.3f\n' % (r[1], r[2], r[0], r[3], r[4], r[5], r[6], r[7], r[8]))
        print('%-70s lds_busy %.3f conflict %.3f stall %.3f lds_stall %.3f' % (r[1][:70], r[3], r[4], r[7], r[8]))
PY
tail -3 /tmp/pmc_lds.log | cut -c1-200

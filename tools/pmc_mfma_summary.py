"""Summarise a rocprofv3 PMC results db (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_WAVE_CYCLES, SQ_WAIT_ANY,
SQ_ACTIVE_INST_ANY) per kernel.  usage: pmc_mfma_summary.py <results.db> <out.csv> <description>"""
import sqlite3, sys
db, out_path, desc = sys.argv[1], sys.argv[2], sys.argv[3]
cur = sqlite3.connect(db).cursor()
d = {}
for k, c, n, v in cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
    d.setdefault(k, {})[c] = (n, v)
rows = []
for k, v in d.items():
    act = v.get('GRBM_GUI_ACTIVE', (0, 0))[1]
    if act <= 0 or 'SQ_VALU_MFMA_BUSY_CYCLES' not in v:
        continue
    wc = max(v.get('SQ_WAVE_CYCLES', (0, 1))[1], 1)
    rows.append((act, k, v['GRBM_GUI_ACTIVE'][0], v['SQ_VALU_MFMA_BUSY_CYCLES'][1] / (act / 8 * 1024),
                 v.get('SQ_WAIT_ANY', (0, 0))[1] / wc, v.get('SQ_ACTIVE_INST_ANY', (0, 0))[1] / wc))
rows.sort(reverse=True)
with open(out_path, 'w') as f:
    f.write('# %s\n' % desc)
    f.write('# GRBM_GUI_ACTIVE is summed over the 8 XCDs: mfma_util = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (sum(GRBM_GUI_ACTIVE)/8 * 1024 SIMDs); wait / issue = fractions of SQ_WAVE_CYCLES\n')
    f.write('kernel,launches,gpu_active_cycles,mfma_util,wave_wait_frac,wave_issue_frac\n')
    for act, k, n, u, wa, ai in rows[:16]:
        f.write('"%s",%d,%.4g,%.3f,%.3f,%.3f\n' % (k, n, act, u, wa, ai))
        print('%-86s n=%5d active=%.3g mfma_util=%.3f wait=%.2f issue=%.2f' % (k[:86], n, act, u, wa, ai))

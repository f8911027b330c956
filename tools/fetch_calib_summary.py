"""FETCH_SIZE / WRITE_SIZE per launch of tools/probe/fetch_calib divided by the known bytes.
usage: fetch_calib_summary.py <fetch_results.db> <write_results.db> <out.txt>"""
import sqlite3, sys
fetch_db, write_db, out = sys.argv[1:4]
BYTES = float(1 << 30)
lines = ['# tools/probe/fetch_calib on MI355X: every kernel moves exactly 1 GiB per launch through a 1 GiB buffer (4x the Infinity Cache);',
         '# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes); counters are in KB; ratio = counter bytes / true bytes',
         'kernel,counter,launches,KB_per_launch,ratio_to_true_bytes']
for db, ctr in ((fetch_db, 'FETCH_SIZE'), (write_db, 'WRITE_SIZE')):
    cur = sqlite3.connect(db).cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    src = 'counters_collection' if 'counters_collection' in tables else None
    if src is None:
        lines.append('# %s: no counters_collection view in %s (tables: %s)' % (ctr, db, tables[:8]))
        continue
    for name, n, avg in cur.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name='%s' group by kernel_name" % ctr):
        lines.append('%s,%s,%d,%.1f,%.3f' % (name.split('(')[0], ctr, n, avg, avg * 1024 / BYTES))
open(out, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))

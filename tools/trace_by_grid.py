"""Per-launch view of a rocprofv3 --kernel-trace CSV: launches grouped by (kernel, grid, workgroup), with count, average and total
duration -- what the --stats summary hides when one kernel serves five problem sizes.
usage: python tools/trace_by_grid.py <kernel_trace.csv> [substring filter ...]"""
import csv, re, sys
from collections import defaultdict

path, filters = sys.argv[1], sys.argv[2:]
rows = list(csv.DictReader(open(path)))
if not rows:
    sys.exit('empty trace')
key = {k.lower(): k for k in rows[0]}
def col(r, name):
    return r[key[name]]
groups = defaultdict(list)
for r in rows:
    name = re.sub(r'\(.*$', '', col(r, 'kernel_name')).replace('void ', '')
    if filters and not any(f in name for f in filters):
        continue
    grid = tuple(int(col(r, 'grid_size_' + a)) // max(1, int(col(r, 'workgroup_size_' + a))) for a in 'xyz')
    groups[(name, grid, int(col(r, 'workgroup_size_x')))].append((int(col(r, 'end_timestamp')) - int(col(r, 'start_timestamp'))) / 1e3)
tot = sum(sum(v) for v in groups.values())
print('%-64s %-18s %5s %6s %10s %10s %6s' % ('kernel', 'blocks', 'wg', 'calls', 'avg us', 'total us', '%'))
for (name, grid, wg), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    print('%-64s %-18s %5d %6d %10.1f %10.1f %6.2f' % (name[:64], 'x'.join(map(str, grid)), wg, len(v), sum(v) / len(v), sum(v), 100 * sum(v) / tot))

"""Summarise a rocprofv3 kernel_trace.csv: per kernel name count / mean duration, and for the eigensolver kernels the
mean gap between consecutive kernels on the same queue (launch boundary cost of the serial chain)."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = re.sub(r'\(.*', '', n)
    return n[:70]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = collections.defaultdict(list)
for r in rows:
    dur[short(r['Kernel_Name'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in dur.values())
print('kernel                                                                 count   mean_us   total_ms   share')
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print('%-70s %6d %9.2f %9.3f %6.1f%%' % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6, 100.0 * sum(v) / tot))
t0 = int(rows[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in rows)
print('span %.3f ms, sum of kernel durations %.3f ms, kernels %d' % ((t1 - t0) / 1e6, tot / 1e6, len(rows)))
# per-queue gaps for jacobi kernels
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get('Queue_Id', '0')].append(r)
gaps = collections.defaultdict(list)
for q, rs in byq.items():
    for a, b in zip(rs, rs[1:]):
        na, nb = short(a['Kernel_Name']), short(b['Kernel_Name'])
        if 'jacobi' in na and 'jacobi' in nb:
            gaps[(na[:40], nb[:40])].append(int(b['Start_Timestamp']) - int(a['End_Timestamp']))
print('gaps between consecutive jacobi kernels on one queue (end -> next start):')
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:12]:
    v2 = sorted(v)
    print('  %-40s -> %-40s n=%5d mean %.2f us median %.2f us' % (k[0], k[1], len(v), sum(v) / len(v) / 1e3, v2[len(v2) // 2] / 1e3))

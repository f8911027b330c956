#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/r03_trace.sh r04_d "8 1" 2>&1 | tail -2
python - <<'PY'
import csv, gzip, collections
for b in (8, 1):
    rows = list(csv.DictReader(gzip.open('gpurun_out/r04_d_trace_b%d.csv.gz' % b,'rt')))
    d = collections.defaultdict(list)
    for r in rows:
        if 'jacobi' in r['Kernel_Name']:
            g = int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])
            d[(r['Kernel_Name'][:34], g)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    print('batch', b)
    for g,v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:14]:
        v2=sorted(v)
        print('  %-34s blocks %6d: n=%4d mean %6.1f us median %6.1f total %.2f ms' % (g[0], g[1], len(v), sum(v)/len(v), v2[len(v2)//2], sum(v)/1e3))
PY
head -12 gpurun_out/r04_d_trace_b8.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04_run7.txt
: > $OUT
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py -x -q -k "eigh or wct_matches or hard_512 or straddle or tf_mode or loud or cutoff or batch32 or config3 or fused_pipeline" 2>&1 | tail -3 ) >> $OUT
for B in 32 8 1; do
  timeout 300 python bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch %3d: %.1f frames/s, %.2f ms/step; jacobi %.2f ms, conv3x3 %.2f, apply %.2f, cov %.2f; sweeps %s' % (l['config']['global_batch'], l['value'], l['ms_per_step'], l['breakdown_ms_per_step']['jacobi'], l['breakdown_ms_per_step']['conv3x3'], l['breakdown_ms_per_step']['wct_apply'], l['breakdown_ms_per_step']['wct_cov'], {k:v['mean'] for k,v in l['eigensolver']['sweeps'].items()}))" >> $OUT 2>&1
done
bash tools/r03_trace.sh r04_b "32" >> $OUT 2>&1
python - >> $OUT <<'PY'
import csv, gzip, collections
rows = list(csv.DictReader(gzip.open('gpurun_out/r04_b_trace_b32.csv.gz','rt')))
d = collections.defaultdict(list)
for r in rows:
    if 'jacobi_fused' in r['Kernel_Name']:
        g = int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])
        d[(r['Kernel_Name'][:40], g)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for g,v in sorted(d.items()):
    print('%s blocks %6d: n=%4d mean %.1f us min %.1f max %.1f' % (g[0], g[1], len(v), sum(v)/len(v), min(v), max(v)))
PY
cat $OUT

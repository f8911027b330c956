#!/bin/bash
# A/B of an environment switch over the bench batch sweep: usage (gpurun): bash tools/r03_ab.sh "VAR=a VAR=b" "1 8 32"
cd $GRAFT_REPO_ROOT
for v in $1; do
  for b in $2; do
    env $v timeout 200 python bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); bd = d['breakdown_ms_per_step']
        print('$v batch %2d: %.1f frames/s, %.2f ms/step, eig %.2f conv3x3 %.2f cov %.2f apply %.2f' % (d['config']['pairs_per_gpu_per_step'], d['value'], d['ms_per_step'], bd['jacobi'], bd['conv3x3'], bd['wct_cov'], bd['wct_apply']))
"
  done
done

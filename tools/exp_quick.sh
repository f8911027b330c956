#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp_quick.txt
: > $O
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-latency"
timeout 300 $B 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('fps %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), {k: round(v, 2) for k, v in d['breakdown_ms_per_step'].items()}, 'conv frac %.3f' % d['roofline']['frac'])
" >> $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3 >> $O
cat $O

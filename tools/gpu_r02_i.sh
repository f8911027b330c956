#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out
timeout 900 python -m pytest tests -m gpu -q -s --durations=8 > $OUT/r02i_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/r02i_pytest.log
grep -E "passed|failed" $OUT/r02i_pytest.log | tail -3
grep -E "^FAILED" $OUT/r02i_pytest.log
for b in 32 16 8 4 1; do
  python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null > $OUT/r02i_bench_b$b.json
  python -c "
import json
d=json.load(open('$OUT/r02i_bench_b$b.json')); print('b=$b fps %.1f ms %.2f'%(d['value'],d['ms_per_step']), {k: round(v,2) for k,v in d['breakdown_ms_per_step'].items()})"
done

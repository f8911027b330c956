#!/bin/bash
# GPU session C: full parity suite, eigensolver group-count sweep over batch sizes
export TMPDIR=/tmp
OUT=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/r02c_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/r02c_pytest.log
grep -E "passed|failed" $OUT/r02c_pytest.log | tail -3
for b in 64 16 4 2 1; do
  for ng in 1 2 4; do
    [ $b -eq 64 ] && continue
    WCT_EIG_NGRP=$ng timeout 300 python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline > $OUT/r02c_bench_ng${ng}_b${b}.json 2> $OUT/r02c_bench.err
    python - <<PY
import json
try:
    d=json.load(open('$OUT/r02c_bench_ng${ng}_b${b}.json'))
    print('ngrp=$ng batch=$b fps %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), {k: round(v,2) for k,v in d['breakdown_ms_per_step'].items()})
except Exception as e:
    print('ngrp=$ng batch=$b FAILED', e)
PY
  done
done

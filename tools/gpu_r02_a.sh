#!/bin/bash
# GPU session A of round 2: full parity suite, then eigensolver A/B (round-1 pair kernel vs pivot wave) at batch 32 / 8 / 1
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q -s > $OUT/r02a_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/r02a_pytest.log
tail -5 $OUT/r02a_pytest.log
for pw in 0 1; do
  for b in 32 8 1; do
    WCT_JACOBI_PW=$pw timeout 300 python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline > $OUT/r02a_bench_pw${pw}_b${b}.json 2> $OUT/r02a_bench_pw${pw}_b${b}.err
    python - <<PY
import json
try:
    d=json.load(open('$OUT/r02a_bench_pw${pw}_b${b}.json'))
    print('pw=$pw batch=$b fps %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), {k: round(v,2) for k,v in d['breakdown_ms_per_step'].items()})
except Exception as e:
    print('pw=$pw batch=$b FAILED', e)
PY
  done
done
cd /tmp
for b in 32 8; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b$b -o prof -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps 2 --warmup 1 --no-cpu-baseline --no-prof > /tmp/prof_b$b.log 2>&1
  f=$(find /tmp/prof_b$b -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/$OUT/r02a_kernel_stats_b$b.csv
done
ls -la $GRAFT_REPO_ROOT/$OUT | tail -20

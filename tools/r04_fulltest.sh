#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( python -m pytest tests/ -x -q -m gpu 2>&1 | tail -40; echo "pytest rc=${PIPESTATUS[0]}" ) > gpurun_out/r04_mid_pytest_gpu.log 2>&1
tail -5 gpurun_out/r04_mid_pytest_gpu.log
python bench.py --no-cpu-baseline > gpurun_out/r04_mid_bench.json 2> gpurun_out/r04_mid_bench.err
cut -c1-600 gpurun_out/r04_mid_bench.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3

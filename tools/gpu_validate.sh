#!/bin/bash
# Round validation on the GPU box: full GPU test suite, the default bench line, smoke(), then the profile passes.
# usage (gpurun): bash tools/gpu_validate.sh <tag>
TAG=${1:-r03_final}
cd $GRAFT_REPO_ROOT
( python -m pytest tests/ -x -q -m gpu 2>&1 | tail -60; echo "pytest rc=${PIPESTATUS[0]}" ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1
tail -3 gpurun_out/${TAG}_pytest_gpu.log
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cut -c1-400 gpurun_out/${TAG}_bench.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tools/gpu_profile.sh $TAG 2>&1 | tail -15

#!/bin/bash
# last check of the round on the GPU box: the tests that exercise the eigensolver through the pipeline (bit identities, end to
# end against the oracle, fuzz), then the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 270 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fuzz.py tests/test_gpu_ops.py -x -q -k "batch32_at_512 or fused_equals_stepwise or config3_five_levels_512_end_to_end_on or shared_style or fused_pipeline_equals_chained or wct_random_shapes_and_scales or reports_failed or golden or hard_512 or cutoff" 2>&1 | tail -5 ) > gpurun_out/r04_last_pytest.log
cat gpurun_out/r04_last_pytest.log
timeout 200 python bench.py > gpurun_out/r04_last_bench.json 2> gpurun_out/r04_last_bench.err
cut -c1-260 gpurun_out/r04_last_bench.json

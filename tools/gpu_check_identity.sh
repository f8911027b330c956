#!/bin/bash
# After a change that must not move a bit: the eigensolver / transform op tests, the pipeline's bit identities, and the frames' digest at
# four batch sizes (compare with the digests of the library before: profiles/r05_intra_registers.txt).  usage (gpurun): bash tools/gpu_check_identity.sh <out-name>
NAME=${1:-identity}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 400 python -m pytest tests/test_gpu_ops.py -k "eigh or eigensolver or wct" -x -q 2>&1 | tail -2
  timeout 600 python -m pytest tests/test_gpu_pipeline.py -k "equals_single_pairs or fused_equals_stepwise or bit_for_bit or shared_style_batch_equals_per_pair" -x -q 2>&1 | tail -2 ) > gpurun_out/${NAME}_tests.txt
cat gpurun_out/${NAME}_tests.txt
for B in 32 16 8 1; do
  python bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=l['breakdown_ms_per_step']
print('batch %2d: %6.1f frames/s %6.2f ms (no_prof %.2f) conv3x3 %5.2f jacobi %5.2f apply %4.2f | frames %s' % (l['config']['global_batch'], l['value'], l['ms_per_step'], l['no_prof']['ms_per_step'], b['conv3x3'], b['jacobi'], b['wct_apply'], l['frames_sha256'][:12]))"
done > gpurun_out/${NAME}_digests.txt
cat gpurun_out/${NAME}_digests.txt

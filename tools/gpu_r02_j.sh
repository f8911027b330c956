#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out
t0=$(date +%s)
python bench.py > $OUT/r02j_bench_default.json 2> $OUT/r02j_bench_default.err; echo "default bench rc=$? in $(( $(date +%s) - t0 )) s"
python -c "
import json; d=json.load(open('$OUT/r02j_bench_default.json')); print(json.dumps({k:v for k,v in d.items() if k not in ('config',)}, indent=0)[:3000])"
python bench.py --global-batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('strong G=64 N=1: fps %.1f ms %.2f'%(d['value'],d['ms_per_step']), d['scaling'], d['config']['parallelism'][:80])"
echo "--- two ranks sharing the GPU, gloo staging (dry run of the N>1 control flow)"
WCT_BENCH_BACKEND=gloo WCT_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --batch 8 > $OUT/r02j_dryrun_gloo_weak.json 2> $OUT/r02j_dryrun_gloo_weak.err; echo rc=$?; tail -c 600 $OUT/r02j_dryrun_gloo_weak.json
WCT_BENCH_BACKEND=gloo WCT_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --global-batch 16 > $OUT/r02j_dryrun_gloo_strong.json 2> $OUT/r02j_dryrun_gloo_strong.err; echo rc=$?; tail -c 600 $OUT/r02j_dryrun_gloo_strong.json
echo "--- two ranks sharing the GPU over RCCL (may be refused: duplicate GPU)"
WCT_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 3 --warmup 1 --global-batch 16 > $OUT/r02j_dryrun_rccl_strong.json 2> $OUT/r02j_dryrun_rccl_strong.err; echo rc=$?; tail -c 400 $OUT/r02j_dryrun_rccl_strong.json; tail -5 $OUT/r02j_dryrun_rccl_strong.err | cut -c1-300

#!/usr/bin/env python
"""Throughput of the stylize hot path on MI355X.

A step = one pass of the full 5-level relu5_1->relu1_1 pipeline (alpha 0.8, WCT at every
level, style features recomputed -- wct.py:97-103 behaviour) over one batch of synthetic
512x512 content/style pairs that is already resident in HBM.  One process per GPU; each
rank stylizes its own shard of independent pairs (no data-path collective) and, for N > 1,
the finished uint8 frames are gathered to rank 0 over RCCL at the end of every step.

Prints ONE JSON line on rank 0 (see the driver contract): metric/value = whole-job stylized
frames/sec, plus `roofline` for the dominant kernel class (conv3x3 on fp16 MFMA; achieved
TFLOP/s from HIP events around every launch of that class on the library's stream) and
`cpu_baseline` (the NumPy oracle of the same path timed on this host's cores, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

LEVELS = ['relu5_1', 'relu4_1', 'relu3_1', 'relu2_1', 'relu1_1']
MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense fp16/bf16
HBM_PEAK_GBS = 8000.0
PMC_BATCH = 32                            # batch the committed PMC passes were collected at


def conv_flops_per_frame(size):
    """Algorithmic conv FLOPs of one 5-level frame: style encoder to relu5_1,
    five content encoders, five decoders; 2*H*W*9*Cin*Cout per layer."""
    from wct_tf_amd.weights import ENCODER_CONVS, decoder_plan, RELU_LEVEL

    def enc(level):
        h = size
        total = 0.0
        for name, cin, cout in ENCODER_CONVS:
            lvl = int(name[4])
            hh = size >> (lvl - 1)
            total += 2.0 * hh * hh * 9 * cin * cout
            if name == 'conv%d_1' % level:
                break
        return total

    def dec(relu):
        h = size >> (RELU_LEVEL[relu] - 1)
        total = 0.0
        for kind, cin, cout, _ in decoder_plan(relu):
            if kind == 'U':
                h *= 2
            else:
                total += 2.0 * h * h * 9 * cin * cout
        return total

    return enc(5) + sum(enc(RELU_LEVEL[r]) for r in LEVELS) + sum(dec(r) for r in LEVELS)


def pmc_traffic(batch, size):
    """HBM bytes per conv3x3 launch from the committed rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in
    separate runs, gfx950 correction applied by tools/summarize_prof.py).  PMC counters cannot be read
    from inside this process, so the figure is the profile of this exact workload; null otherwise."""
    path = os.path.join(ROOT, 'profiles', 'r01_final_pmc_conv3x3.json')
    if batch != PMC_BATCH or size != 512 or not os.path.exists(path):
        return None
    return json.load(open(path))['hbm_bytes_per_launch_corrected']


def cpu_baseline(size, weights):
    """The CPU oracle (NumPy restatement of the reference path, OpenBLAS threads) timed on a
    bounded sample: ONE frame of the same workload."""
    import oracle
    from wct_tf_amd.weights import synthetic_image
    c = synthetic_image(1000, size, size)
    s = synthetic_image(2000, size, size)
    t0 = time.time()
    oracle.stylize(c, s, weights, LEVELS, alpha=0.8, wct_mode='tf')
    dt = time.time() - t0
    return {'value': 1.0 / dt, 'unit': 'frames/s', 'cores': os.cpu_count(), 'kind': 'port',
            'sample': '1 frame %dx%d, 5-level, alpha 0.8, NumPy oracle (im2col+OpenBLAS convs, LAPACK SVD), %.1f s' % (size, size, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=32, help='independent content/style pairs per GPU per step')
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--alpha', type=float, default=0.8)
    ap.add_argument('--shared-style', action='store_true',
                    help='NOT the headline metric: every pair of a step uses ONE style image (fixed-style video, '
                         'WCT_FLAG_STYLE_SHARED): the style side runs once per step instead of once per frame')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prof', action='store_true', help='no per-class HIP-event timing inside the timed region')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the stylize path has no CPU fallback')
    # dry-run switches for a box with fewer GPUs than ranks (the control flow of the N > 1 path without RCCL):
    # WCT_BENCH_BACKEND=gloo stages the exchange through the host, WCT_BENCH_SHARE_GPU=1 wraps ranks onto the GPUs
    backend = os.environ.get('WCT_BENCH_BACKEND', 'nccl')
    if os.environ.get('WCT_BENCH_SHARE_GPU'):
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend, rank=rank, world_size=world)      # 'nccl' = RCCL over xGMI

    from wct_tf_amd.context import Context
    from wct_tf_amd.weights import synthetic_weights, synthetic_image
    from wct_tf_amd.dist import shard_range, gather_frames

    weights = synthetic_weights(seed=42)
    ctx = Context(local_rank)
    ctx.set_weights(weights)

    B, S = args.batch, args.size
    total_pairs = B * world
    lo, hi = shard_range(total_pairs, world, rank)                         # contiguous shard of the global batch
    content = np.stack([synthetic_image(1000 + i, S, S) for i in range(lo, hi)])
    style = np.stack([synthetic_image(2000 + i, S, S) for i in range(lo, hi)])
    if args.shared_style:
        style = style[0]
    dev = torch.device('cuda', local_rank)
    d_content = torch.from_numpy(content).to(dev)                          # inputs resident in HBM
    d_style = torch.from_numpy(style).to(dev)
    d_out = torch.empty_like(d_content)
    torch.cuda.synchronize()

    import ctypes as C

    def step():
        ctx.stylize_batch_dev(C.c_void_p(d_content.data_ptr()), S, S, C.c_void_p(d_style.data_ptr()), S, S,
                              B, LEVELS, args.alpha, C.c_void_p(d_out.data_ptr()), shared_style=args.shared_style)
        if world > 1:
            ctx.sync()                                                     # library stream -> torch stream hand-off
            frames = gather_frames(d_out if backend == 'nccl' else d_out.cpu(), world, rank)
            torch.cuda.synchronize()                                       # RCCL must be done with d_out before the
            return frames                                                  # next step overwrites it on the library stream
        return None

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    if not args.no_prof:
        ctx.prof_reset()
        ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ctx.prof_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    prof = None if args.no_prof else ctx.prof_read()
    if rank == 0:
        frames = total_pairs * args.steps
        fps = frames / dt
        line = {
            'metric': 'stylized frames/sec @512x512, 5-level relu5->1 pipeline, alpha=0.8',
            'value': fps, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': 'configs[2]: full 5-level relu5_1->relu1_1, %dx%d content+style, alpha %.1f, '
                                   'wct_tf semantics, %s' % (S, S, args.alpha, 'ONE style per step (fixed-style video mode, not the headline '
                                   'metric)' if args.shared_style else 'style features recomputed per frame'),
                       'pairs_per_gpu_per_step': B, 'global_batch': total_pairs,
                       'parallelism': 'pairs sharded over %d GPU(s), RCCL gather of uint8 frames' % world,
                       'weights': 'synthetic He-normal seed 42 (no pre-trained weights offline)'},
        }
        if prof is not None:
            conv = prof['conv3x3']
            ach = conv['flops'] / (conv['ms'] * 1e-3) / 1e12 if conv['ms'] > 0 else 0.0
            line['roofline'] = {
                'bound': 'mfma', 'achieved': ach, 'peak': MFMA_F16_DENSE_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': ach / MFMA_F16_DENSE_PEAK_TFLOPS, 'traffic': pmc_traffic(B, S),
                'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC, profiles/r01_final_pmc_hbm.csv)',
                'algorithmic_bytes_per_launch': conv['bytes'] / max(1, conv['launches']),
                'kernel': 'conv3x3_mfma_kernel (all launches of the class)',
                'launches': conv['launches'], 'avg_launch_ms': conv['ms'] / max(1, conv['launches']),
                'algorithmic_flops_per_frame': conv_flops_per_frame(S),
                'algorithmic_gbytes_per_s': conv['bytes'] / (conv['ms'] * 1e-3) / 1e9 if conv['ms'] > 0 else 0.0,
            }
            line['breakdown_ms_per_step'] = {k: v['ms'] / args.steps for k, v in prof.items()}
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(S, weights)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == '__main__':
    main()

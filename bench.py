#!/usr/bin/env python
"""Throughput of the stylize hot path on MI355X.

A step = one pass of the full 5-level relu5_1->relu1_1 pipeline (alpha 0.8, WCT at every
level, style features recomputed -- wct.py:97-103 behaviour) over one batch of synthetic
512x512 content/style pairs that is already resident in HBM.  One process per GPU; each
rank stylizes its own shard of independent pairs (no data-path collective) and, for N > 1,
the finished uint8 frames are gathered to rank 0 over RCCL at the end of every step.

Prints ONE JSON line on rank 0 (see the driver contract): metric/value = whole-job stylized
frames/sec (device-resident batch throughput), plus `roofline` for the dominant kernel class
(conv3x3 on fp16 MFMA; achieved TFLOP/s from HIP events around every launch of that class on the
library's stream), `eigensolver` (the second-largest class), `latency_fps` (batch 1, host uint8
in -> host uint8 out: SURVEY 8d asks for both figures) and `cpu_baseline` (BASELINE.md section 3:
NumPy transform + torch-CPU conv stand-in on this host's cores, rank 0, N=1).

Scaling modes: weak (default; --batch pairs per GPU per step) and strong (--global-batch G pairs
per step in total: BASELINE configs[3] is --global-batch 64 on 8 GPUs = 8 pairs per GPU).  With --gpus N > 1 the
headline is the weak figure and the same line carries a `strong` sub-record (64 pairs per step over the N GPUs,
timed right after), so one driver run measures configs[3] as written.

`eigensolver` in the line: ms per step, the sweeps the covariances took (mean / max per channel count), the
fp32-MFMA rate of its tile updates against the 157.3 TFLOP/s peak, and `hard_spectrum`: the transform on graded
512-channel covariances (eigenvalues over 6 decades; N = 4096 and N = 256 < C) -- `--spectrum graded` prints only
that leg.
"""
import argparse
import hashlib
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


F32_MFMA_PEAK_TFLOPS = 157.3               # MI355X_MICROARCH.md: v_mfma_f32_* at the fp32 vector rate


def jacobi_flops_per_sweep(c):
    """fp32-MFMA FLOPs of one sweep of the block Jacobi on one C x C matrix: per outer step the two-sided update of the
    off-diagonal tiles (two M2^3 products each, mirror tiles are copies) and V <- V Q (C rows x the pairs' M2 x M2
    rotations); C/B outer steps per sweep.  Block pairs of 64 indices from C = 256 up, 32 below (csrc/wct.hip)."""
    m2 = 64 if (c >= 256 and c % 64 == 0) else 32
    nblk = c // (m2 // 2)
    npair = nblk // 2
    n_off = npair * (npair - 1) // 2
    return nblk * (n_off * 2 * 2.0 * m2 ** 3 + 2.0 * c * c * m2)


def graded_features(seed, n, c, decades):
    """post-ReLU-like features whose covariance is graded: per-channel scales log-spaced over `decades`/2 (eigenvalues
    over ~`decades`), mild channel mixing"""
    rng = np.random.default_rng(seed)
    mix = np.eye(c) + 0.3 * rng.standard_normal((c, c)) / np.sqrt(c)
    d = 10.0 ** (-np.arange(c) * decades / (c - 1) / 2)
    return np.float32(np.maximum(rng.standard_normal((n, c)) @ mix + 0.3, 0) * d * 3.0)


def hard_spectrum_leg(ctx, c=512, decades=6.0):
    """Sweeps and eigensolver time of wct_transform (one pair, wct_tf semantics) on graded covariances: N = 8 C
    (full rank, eigenvalues over `decades`) and N = C / 2 (rank-deficient: the null space is rounding noise)."""
    from wct_tf_amd import _lib
    out = []
    for n in (8 * c, c // 2):
        fc, fs = graded_features(11, n, c, decades), graded_features(12, n, c, decades)
        ctx.transform(fc, fs, 0.8, _lib.WCT_TF)
        ctx.prof_reset(); ctx.prof_enable(True)
        reps = 5
        for _ in range(reps):
            _, sweeps = ctx.transform(fc, fs, 0.8, _lib.WCT_TF, return_sweeps=True)
        ctx.prof_enable(False)
        ms = ctx.prof_read()['jacobi']['ms'] / reps
        out.append({'C': c, 'N': n, 'eigenvalue_decades': decades, 'sweeps_content_style': [int(x) for x in sweeps[:2]],
                    'eigensolver_ms': ms})
    ctx.eig_stats()
    return out


def pmc_traffic(batch, size):
    """HBM bytes per conv3x3 launch from the COMMITTED rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate
    runs, gfx950 correction applied by tools/summarize_prof.py).  PMC counters cannot be read from inside this
    process, so the figure is the profile of this exact workload, not a measurement of this run; null otherwise."""
    for tag in ('r06_final', 'r05_final', 'r04_final', 'r03_final', 'r02_final', 'r01_final'):
        path = os.path.join(ROOT, 'profiles', '%s_pmc_conv3x3.json' % tag)
        if batch == PMC_BATCH and size == 512 and os.path.exists(path):
            return json.load(open(path))['hbm_bytes_per_launch_corrected'], 'profiles/%s_pmc_hbm.csv' % tag
    return None, None


def cpu_baseline(size, weights, alpha):
    """the reference's path on this host's cores -- its transform in NumPy (oracle.wct_tf, the
    restatement pinned to the reference's own wct_np outputs at these sizes) around a torch-CPU STAND-IN for the
    CPU-TensorFlow conv stack (oracle/torch_path.py: reflect pad + conv2d + max_pool2d(ceil) + nearest upsample, same
    weights, same frames).  One warm-up frame, then the median of 5 frames.  A reported baseline, not the target."""
    import torch
    from oracle.torch_path import TorchPath
    from wct_tf_amd.weights import synthetic_image
    path = TorchPath(weights)
    c = synthetic_image(1000, size, size)
    s = synthetic_image(2000, size, size)
    # where the reference tree is present (the build container; WCT_REFERENCE overrides the
    # path) the transform timed is the reference's OWN wct_np, lifted out of its ops.py and executed -- wct_np(eps=0) +
    # (1 - alpha) mc is the graph's wct_tf (tests/golden/wct_tf_reference.npz pins that) -- `kind: reference`; on a box
    # without the tree (the GPU box) it is the restatement oracle.wct_tf -- `kind: port`.
    # (executing code lifted from an untrusted tree is OPT-IN -- WCT_REFERENCE must name the tree; without it, or if
    #  the lift fails, the restatement runs and the line says why.  eps: wct_np(eps=0) has no 1e-8 on the covariance diagonals,
    #  the graph's wct_tf has; worth 0.5e-8 / lambda_min in a gain, nothing in a timing.)
    transform, kind, why = None, 'port', 'WCT_REFERENCE not set: the restatement (oracle.wct_tf)'
    ref_root = os.environ.get('WCT_REFERENCE')
    if ref_root:
        ref_ops = os.path.join(ref_root, 'ops.py')
        try:
            from oracle.make_golden import lift_function
            ref_wct_np = lift_function(ref_ops, 'wct_np')

            def transform(fc, fs, a):
                mc = fc.reshape(-1, fc.shape[-1]).mean(0, dtype=np.float32)
                return np.float32(ref_wct_np(fc[None], fs[None], a, 0.0) + np.float32(1 - a) * mc)
            kind, why = 'reference', 'wct_np lifted from %s' % ref_ops
        except Exception as e:
            transform, kind, why = None, 'port', 'lifting wct_np from %s failed (%s: %s): the restatement (oracle.wct_tf)' % (ref_ops, type(e).__name__, e)
    path.stylize(c, s, LEVELS, alpha, 'tf', transform=transform)
    times, t_transform = [], []
    for _ in range(5):
        timers = {}
        t0 = time.time()
        path.stylize(c, s, LEVELS, alpha, 'tf', timers=timers, transform=transform)
        times.append(time.time() - t0)
        t_transform.append(timers['transform_s'])
    med = sorted(times)[len(times) // 2]
    med_t = sorted(t_transform)[len(t_transform) // 2]
    try:
        from threadpoolctl import threadpool_info
        blas = max([p.get('num_threads', 0) for p in threadpool_info() if p.get('user_api') == 'blas'] or [0])
    except Exception:
        blas = 0
    what = ("the reference's own wct_np(eps=0) + (1-alpha) mc lifted from its ops.py (= its wct_tf graph), LAPACK SVD" if kind == 'reference'
            else "oracle.wct_tf, the RESTATEMENT of the reference's wct_tf/wct_np (no reference tree on this box), LAPACK SVD")
    return {'value': 1.0 / med, 'unit': 'frames/s', 'cores': max(torch.get_num_threads(), blas), 'kind': kind,
            'host_cores': os.cpu_count(), 'torch_threads': torch.get_num_threads(), 'blas_threads': blas,
            # part (i), the reference's transform in NumPy (five whiten-colour transforms per frame, LAPACK
            # SVD), reported separately from part (ii), the torch-CPU stand-in for the CPU-TF conv stack
            'transform_s': med_t, 'conv_standin_s': med - med_t, 'transform_source': why,
            'sample': '1 warm-up + median of 5 frames %dx%d, 5-level, alpha %.1f: NumPy transform (%s) + torch-CPU stand-in for the '
                      'CPU-TF conv stack; %.2f s per frame (min %.2f, max %.2f)' % (size, size, alpha, what, med, min(times), max(times))}


def real_image_leg(ctx, alpha, n=10):
    """The headline's spectra are those of synthetic noise images; this leg puts a PHOTOGRAPH next to them: the reference's
    sample photo (samples/gilbert.jpg resampled to 512 x 512: tests/golden/gilbert_512.npz, written by oracle/make_golden.py)
    as content and its mirror image as style through the five levels, batch 1, resident inputs -- sweeps per channel
    count, eigensolver ms per frame, frames/s.  (The weights stay the synthetic He-normal stand-in: no VGG weights offline.)"""
    path = os.path.join(ROOT, 'tests', 'golden', 'gilbert_512.npz')
    if not os.path.exists(path):
        return None
    img = np.load(path)['image']
    c, s = np.ascontiguousarray(img), np.ascontiguousarray(img[:, ::-1])
    # warm-up with the per-class events ON and as many frames as the timed loop: the context creates its HIP events on demand and
    # recycles them when the records are drained (prof_reset below), so the timed frames find every event they need in the pool
    # (created inside the timed loop they cost 0.7 ms per frame in a fresh process and 4 ms after the 32-pair steps)
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(n):
        ctx.stylize(c, s, LEVELS, alpha=alpha)
    ctx.eig_stats()
    ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(n):
        out = ctx.stylize(c, s, LEVELS, alpha=alpha)
    dt = (time.perf_counter() - t0) / n
    ctx.prof_enable(False)
    prof, eig = ctx.prof_read(), ctx.eig_stats()
    return {'image': 'samples/gilbert.jpg resampled to 512x512 (content) and its mirror image (style), 5 levels, alpha %.1f, batch 1' % alpha,
            'frames_per_s': 1.0 / dt, 'ms_per_frame': 1e3 * dt, 'eigensolver_ms_per_frame': prof['jacobi']['ms'] / n,
            'sweeps': {str(cc): {'mean': v['sweeps'] / max(1, v['matrices']), 'max': v['max_sweeps']} for cc, v in sorted(eig.items())},
            'output_std': float(np.asarray(out, np.float64).std()),
            'note': 'host uint8 in -> host uint8 out like latency_ms; synthetic He-normal weights'}


def latency_leg(ctx, size, alpha, n=20):
    """SURVEY 8d latency mode: ONE predict() -- host uint8 content + style in, host uint8 frame out (wct_stylize,
    PCIe transfers included), batch 1, style recomputed."""
    from wct_tf_amd.weights import synthetic_image
    c, s = synthetic_image(1000, size, size), synthetic_image(2000, size, size)
    for _ in range(3):
        ctx.stylize(c, s, LEVELS, alpha=alpha)
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.stylize(c, s, LEVELS, alpha=alpha)
    dt = (time.perf_counter() - t0) / n
    return {'latency_fps': 1.0 / dt, 'latency_ms': 1e3 * dt,
            'latency_note': 'batch 1, wct_stylize: host uint8 in -> host uint8 out, mean of %d calls' % n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=32, help='weak scaling (default): independent content/style pairs per GPU per step')
    ap.add_argument('--global-batch', type=int, default=0,
                    help='strong scaling: this many pairs per step IN TOTAL, sharded over the GPUs (BASELINE configs[3]: 64 over '
                         '8 GPUs = 8 per GPU); 0 = weak scaling with --batch pairs per GPU (and, for --gpus > 1, a `strong` '
                         'sub-record with 64 pairs per step in the same line)')
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--alpha', type=float, default=0.8)
    ap.add_argument('--shared-style', action='store_true',
                    help='NOT the headline metric: every pair of a step uses ONE style image (fixed-style video, '
                         'WCT_FLAG_STYLE_SHARED): the style side runs once per step instead of once per frame')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-latency', action='store_true', help='skip the batch-1 host-in/host-out latency leg')
    ap.add_argument('--no-prof', action='store_true', help='no per-class HIP-event timing inside the timed region')
    ap.add_argument('--spectrum', choices=['frames', 'graded'], default='frames',
                    help="'graded': only the eigensolver's hard-spectrum leg (graded 512-channel covariances, N = 4096 and "
                         "N = 256 < C): sweeps and ms")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from wct_tf_amd.dist import resolve_world, launch_ranks
    # dry-run switches for a box with fewer GPUs than ranks (the control flow of the N > 1 path without RCCL):
    # WCT_BENCH_BACKEND=gloo stages the exchange through the host, WCT_BENCH_SHARE_GPU=1 wraps ranks onto the GPUs
    backend = os.environ.get('WCT_BENCH_BACKEND', 'nccl')
    share_gpu = bool(os.environ.get('WCT_BENCH_SHARE_GPU'))
    # --gpus N always means N ranks: a launcher's WORLD_SIZE must agree, and without a launcher this process starts them
    # itself (never a silent single-rank run under an n_gpus = N label)
    role = resolve_world(args.gpus, share_gpu=share_gpu)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the stylize path has no CPU fallback')
    if role[0] == 'launch':
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], script=os.path.abspath(__file__)))
    _, rank, world, local_rank = role
    if share_gpu:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend, rank=rank, world_size=world)      # 'nccl' = RCCL over xGMI

    from wct_tf_amd.context import Context
    from wct_tf_amd.weights import synthetic_weights, synthetic_image
    from wct_tf_amd.dist import shard_range, gather_frames

    ctx = Context(local_rank)
    if args.spectrum == 'graded':
        if rank == 0:
            print(json.dumps({'metric': 'eigensolver on graded covariances (wct_transform, one pair)', 'unit': 'ms',
                              'hard_spectrum': hard_spectrum_leg(ctx), 'sweep_budget': 16}), flush=True)
        ctx.close()
        return
    weights = synthetic_weights(seed=42)
    ctx.set_weights(weights)

    S = args.size
    dev = torch.device('cuda', local_rank)
    import ctypes as C
    LIB_MAX = 32                                                            # pairs per library call
    frame_bytes = S * S * 3
    # WCT_BENCH_FORCE_OVERLAP=1: run the event protocol of the overlapped gather on a single rank too (tests)
    overlap = (world > 1 and backend == 'nccl') or bool(os.environ.get('WCT_BENCH_FORCE_OVERLAP'))

    def measure(total_pairs, steps, warmup, prof_on, gather_once=False):
        """`steps` timed steps of `total_pairs` pairs per step sharded over the ranks: (seconds [max over ranks], pairs
        of this rank, per-class profile or None, eigensolver statistics or None)"""
        lo, hi = shard_range(total_pairs, world, rank)                     # contiguous shard of the global batch
        n_local = hi - lo
        content = np.stack([synthetic_image(1000 + i, S, S) for i in range(lo, hi)]) if n_local else np.zeros((0, S, S, 3), np.uint8)
        style = np.stack([synthetic_image(2000 + i, S, S) for i in range(lo, hi)]) if n_local else np.zeros((0, S, S, 3), np.uint8)
        if args.shared_style:
            style = style[:1]
        d_content = torch.from_numpy(content).to(dev)                      # inputs resident in HBM
        d_style = torch.from_numpy(style).to(dev)
        # two output buffers: the gather of step k reads one while step k+1 writes the other
        d_out = [torch.empty_like(d_content), torch.empty_like(d_content)]
        torch.cuda.synchronize()
        chunks = [(a, min(n_local, a + LIB_MAX)) for a in range(0, n_local, LIB_MAX)]

        def compute(out):
            for a, b in chunks:
                sp = d_style.data_ptr() if args.shared_style else d_style.data_ptr() + a * frame_bytes
                ctx.stylize_batch_dev(C.c_void_p(d_content.data_ptr() + a * frame_bytes), S, S, C.c_void_p(sp), S, S, b - a, LEVELS,
                                      args.alpha, C.c_void_p(out.data_ptr() + a * frame_bytes), shared_style=args.shared_style)

        if overlap:
            # the library's stream as a torch stream: events order the RCCL gather behind the frames and the next write of
            # a buffer behind the gather that still reads it -- no host synchronisation between the steps
            lib_stream = torch.cuda.ExternalStream(ctx.stream_handle(), device=dev)
            comm = torch.cuda.Stream(dev)
            done_compute = [torch.cuda.Event(), torch.cuda.Event()]
            done_gather = [torch.cuda.Event(), torch.cuda.Event()]
        state = {'k': 0, 'frames': None}

        def step():
            k = state['k'] & 1
            state['k'] += 1
            if not overlap:
                compute(d_out[k])
                if world > 1:                                              # dry run: stage through the host (gloo)
                    ctx.sync()
                    state['frames'] = gather_frames(d_out[k].cpu(), world, rank, n_items=total_pairs)
                return
            lib_stream.wait_event(done_gather[k])                          # buffer k was last read by the gather two steps ago
            compute(d_out[k])
            done_compute[k].record(lib_stream)
            comm.wait_event(done_compute[k])
            with torch.cuda.stream(comm):                                  # overlaps the next step's kernels
                state['frames'] = gather_frames(d_out[k], world, rank, n_items=total_pairs)
                done_gather[k].record(comm)

        def barrier():
            ctx.sync()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()

        for _ in range(warmup):
            step()
        barrier()
        ctx.eig_stats()                                                    # clear
        if prof_on:
            ctx.prof_reset()
            ctx.prof_enable(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        ctx.prof_enable(False)
        info = {'rank_ms_per_step': [1e3 * dt / max(1, steps)]}
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)                                      # every rank's own clock (min / max go into the line)
            info['rank_ms_per_step'] = [1e3 * float(x.item()) / max(1, steps) for x in every]
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            if rank == 0:
                assert state['frames'] is not None and state['frames'].shape[0] == total_pairs
        # checksum of the frames of the LAST step as rank 0 holds them (N > 1: the gathered tensor).  The inputs are a function
        # of the global pair index alone and a frame does not depend on the batch it is computed in (tested bit for bit), so
        # the digest of `total_pairs` pairs is the same for every N: a wrong shard offset or gather order changes it.
        if rank == 0 and steps + warmup > 0:
            last = state['frames'] if world > 1 else d_out[(state['k'] - 1) & 1]
            info['frames_sha256'] = hashlib.sha256(last.cpu().numpy().tobytes()).hexdigest()
            info['frames'] = int(last.shape[0])
        if world > 1 and gather_once:
            # the exchange step alone, once, NOT overlapped: a barrier, then the gather of the last step's frames, timed on
            # the host around a device synchronisation (max over the ranks)
            barrier()
            k = (state['k'] - 1) & 1
            t0 = time.perf_counter()
            gather_frames(d_out[k] if backend == 'nccl' else d_out[k].cpu(), world, rank, n_items=total_pairs)
            torch.cuda.synchronize()
            tg = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
            dist.all_reduce(tg, op=dist.ReduceOp.MAX)
            info['gather_ms_unoverlapped'] = 1e3 * float(tg.item())
            info['gather_bytes_per_rank'] = int(d_out[k].numel())
        prof = ctx.prof_read() if prof_on else None
        return dt, n_local, prof, ctx.eig_stats(), info

    strong = args.global_batch > 0
    total_pairs = args.global_batch if strong else args.batch * world
    dt, n_local, prof, eig, info = measure(total_pairs, args.steps, args.warmup, not args.no_prof, gather_once=world > 1)
    # the same steps without the per-class events (they cost 1-2 %): reported beside the headline, never instead of it
    dt_np = None
    if not args.no_prof:
        dt_np = measure(total_pairs, args.steps, 1, False)[0]
    # N > 1, weak headline: BASELINE configs[3] as written (64 pairs per step in total) in the same line
    strong_rec = None
    if world > 1 and not strong:
        g = 64
        dts, _, _, _, sinfo = measure(g, args.steps, args.warmup, False, gather_once=True)
        strong_rec = {'global_batch': g, 'pairs_per_gpu_per_step': g / world, 'value': g * args.steps / dts, 'unit': 'frames/s',
                      'ms_per_step': 1e3 * dts / args.steps, 'scaling': 'strong',
                      'rank_ms_per_step_min_max': [min(sinfo['rank_ms_per_step']), max(sinfo['rank_ms_per_step'])],
                      'gather_ms_unoverlapped': sinfo.get('gather_ms_unoverlapped'),
                      'gather_bytes_per_rank': sinfo.get('gather_bytes_per_rank'),
                      'frames_sha256': sinfo.get('frames_sha256'),
                      'note': 'BASELINE configs[3]: 64 frames per step sharded over the GPUs, one RCCL gather per step (overlapped with '
                              'the next step in the timed loop; gather_ms_unoverlapped = the same exchange once on its own); '
                              'frames_sha256 = digest of the 64 gathered frames of the last step: the same for every N, and equal '
                              'to configs3_projection.frames_sha256 of an N = 1 line'}
    # N = 1: what configs[3] would take -- 64 pairs per step over 8 GPUs is 8 pairs per GPU: the batch-8 step of THIS GPU
    # against its own time for all 64 pairs (two 32-pair calls); the gather is not in it
    proj = None
    if world == 1 and not strong and not args.shared_style and args.batch == 32 and not args.no_latency:      # (an extra leg, like the latency: profile passes skip it)
        dt8 = measure(8, args.steps, 2, False)[0]
        dt64, _, _, _, i64 = measure(64, max(1, args.steps // 2), 1, False)
        ms8, ms64 = 1e3 * dt8 / args.steps, 1e3 * dt64 / max(1, args.steps // 2)
        proj = {'batch8_ms_per_step': ms8, 'batch8_frames_per_s': 8e3 / ms8, 'one_gpu_64_pairs_ms': ms64,
                'projected_speedup_8_gpus': ms64 / ms8, 'target': 6.0, 'frames_sha256': i64.get('frames_sha256'),
                'note': 'BASELINE configs[3] (64 frames over 8 GPUs = 8 pairs per GPU) projected from ONE GPU: time of 64 pairs here / '
                        'time of an 8-pair step here, before the 6.3 MB-per-rank gather; frames_sha256 = digest of those 64 frames '
                        '(what the `strong` record of an N > 1 line must reproduce)'}

    if rank == 0:
        frames = total_pairs * args.steps
        fps = frames / dt
        per_gpu = '%d' % (total_pairs // world) if total_pairs % world == 0 else '%d-%d' % (total_pairs // world, total_pairs // world + 1)
        line = {
            'metric': 'stylized frames/sec @512x512, 5-level relu5->1 pipeline, alpha=0.8',
            'value': fps, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'strong' if strong else 'weak',
            'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': 'configs[2]: full 5-level relu5_1->relu1_1, %dx%d content+style, alpha %.1f, '
                                   'wct_tf semantics, %s' % (S, S, args.alpha, 'ONE style per step (fixed-style video mode, not the headline '
                                   'metric)' if args.shared_style else 'style features recomputed per frame'),
                       'pairs_per_gpu_per_step': n_local if world == 1 else per_gpu, 'global_batch': total_pairs,
                       'parallelism': ('strong scaling: %d pairs per step sharded over %d GPU(s) (%s per GPU; BASELINE configs[3] is 64 over 8)'
                                       if strong else 'weak scaling: %d pairs per step = %d GPU(s) x %s') % (total_pairs, world, per_gpu)
                                      + ', no data-path collective, one RCCL gather of the uint8 frames per step overlapped with the next step',
                       'weights': 'synthetic He-normal seed 42 (no pre-trained weights offline)'},
        }
        if dt_np is not None:
            line['no_prof'] = {'value': frames / dt_np, 'unit': 'frames/s', 'ms_per_step': 1e3 * dt_np / args.steps,
                               'note': 'the same %d steps timed again without the per-class HIP events' % args.steps}
        if strong_rec is not None:
            line['strong'] = strong_rec
        if proj is not None:
            line['configs3_projection'] = proj
        line['frames_sha256'] = info.get('frames_sha256')
        if world > 1:
            try:
                rccl = '.'.join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rccl = None
            line['dist'] = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(), 'rccl_version': rccl,
                            'rank_ms_per_step_min_max': [min(info['rank_ms_per_step']), max(info['rank_ms_per_step'])],
                            'gather_ms_unoverlapped': info.get('gather_ms_unoverlapped'),
                            'gather_bytes_per_rank': info.get('gather_bytes_per_rank')}
        if prof is not None:
            conv = prof['conv3x3']
            ach = conv['flops'] / (conv['ms'] * 1e-3) / 1e12 if conv['ms'] > 0 else 0.0
            traffic, traffic_src = pmc_traffic(n_local, S)
            line['roofline'] = {
                'bound': 'mfma', 'achieved': ach, 'peak': MFMA_F16_DENSE_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': ach / MFMA_F16_DENSE_PEAK_TFLOPS, 'traffic': traffic,
                'traffic_unit': 'HBM bytes per launch; NOT measured in this run: the committed rocprofv3 PMC passes of this '
                                'workload (%s)' % traffic_src if traffic else 'no committed PMC pass for this batch/size',
                'algorithmic_bytes_per_launch': conv['bytes'] / max(1, conv['launches']),
                'kernel': 'conv3x3_mfma_kernel, the generic instantiations <..., false> (all their launches: the largest time class of '
                          'the step; since round 6 the >= 256-channel layers without a tap run on conv3x3_wino_kernel -- field conv_wino, '
                          'and conv3x3_direct_and_wino for the 61 launches per frame set that were this class up to round 5; the tap '
                          'launches -- 8 of them -- also take the transform\'s per-channel sums and maxima in their epilogue).  Up to round 3 / profiles/r04_final the class had 65 launches: the four 64->64 '
                          '@512^2 pooled conv1_2 launches of the content passes (0.25 of peak) now run as the instantiation '
                          '<32,64,4,1,true> with conv1_1 inside the patch loader -- class conv12, next field -- and '
                          'all_conv3x3_instantiations gives the figure over both for comparison with the earlier rounds',
                'launches': conv['launches'], 'avg_launch_ms': conv['ms'] / max(1, conv['launches']),
                'algorithmic_flops_per_frame': conv_flops_per_frame(S),
                'algorithmic_gbytes_per_s': conv['bytes'] / (conv['ms'] * 1e-3) / 1e9 if conv['ms'] > 0 else 0.0,
            }
            cw = prof.get('conv_wino')
            if cw and cw['ms'] > 0:
                aw = cw['flops'] / (cw['ms'] * 1e-3) / 1e12
                line['roofline']['conv_wino'] = {
                    'kernel': 'conv3x3_wino_kernel (csrc/conv_wino.hip): the >= 256-channel 3x3 layers without a feature tap on Winograd F(2,3) '
                              'along y x direct along x -- 12 MFMA products per 2 outputs instead of 18.  achieved / frac are quoted on the '
                              'DIRECT convolution\'s FLOPs (comparable with the class above); executed = the MFMA FLOPs the kernel issues (2/3)',
                    'achieved': aw, 'frac': aw / MFMA_F16_DENSE_PEAK_TFLOPS,
                    'executed_tflops': aw / 1.5, 'executed_frac': aw / 1.5 / MFMA_F16_DENSE_PEAK_TFLOPS,
                    'launches': cw['launches'], 'avg_launch_ms': cw['ms'] / max(1, cw['launches'])}
                allc = (conv['flops'] + cw['flops']) / ((conv['ms'] + cw['ms']) * 1e-3) / 1e12
                line['roofline']['conv3x3_direct_and_wino'] = {
                    'achieved': allc, 'frac': allc / MFMA_F16_DENSE_PEAK_TFLOPS, 'launches': conv['launches'] + cw['launches'],
                    'ms_per_step': (conv['ms'] + cw['ms']) / args.steps,
                    'note': 'the 61 launches per frame set that were ONE class up to round 5 (direct-convolution FLOPs / time): compare with roofline.frac of the earlier rounds'}
                # (roofline.frac above is the DIRECT kernel's class alone: since round 6 it has lost its most efficient layers to the
                #  reduced-FLOP kernel, so it reads lower than round 5's 0.471 although no layer got slower; this is the like-for-like figure)
                line['roofline']['frac_over_the_round5_class'] = allc / MFMA_F16_DENSE_PEAK_TFLOPS
            c12 = prof.get('conv12')
            if c12 and c12['ms'] > 0:
                a12 = c12['flops'] / (c12['ms'] * 1e-3) / 1e12
                both = (conv['flops'] + c12['flops']) / ((conv['ms'] + c12['ms']) * 1e-3) / 1e12
                line['roofline']['conv12'] = {
                    'kernel': 'conv3x3_mfma_kernel<32,64,4,1,true>: conv1_1 (3->64, split-fp16 MFMA) computed into the halo patch, '
                              'conv1_2 (64->64), 2x2 max-pool -- one launch instead of conv_first_kernel + conv3x3; algorithmic FLOPs of '
                              'the two layers (the halo recomputation of conv1_1 is not counted)',
                    'achieved': a12, 'frac': a12 / MFMA_F16_DENSE_PEAK_TFLOPS, 'launches': c12['launches'],
                    'avg_launch_ms': c12['ms'] / max(1, c12['launches'])}
                line['roofline']['all_conv3x3_instantiations'] = {
                    'achieved': both, 'frac': both / MFMA_F16_DENSE_PEAK_TFLOPS, 'launches': conv['launches'] + c12['launches']}
            step_ms = 1e3 * dt / args.steps
            jac = prof['jacobi']
            jflops = sum(v['sweeps'] * jacobi_flops_per_sweep(c) for c, v in eig.items())
            jtf = jflops / (jac['ms'] * 1e-3) / 1e12 if jac['ms'] > 0 else 0.0
            line['eigensolver'] = {
                'ms_per_step': jac['ms'] / args.steps, 'share_of_step': jac['ms'] / args.steps / step_ms,
                'matrices_per_step': 2 * 5 * n_local,
                'sweeps': {str(c): {'mean': v['sweeps'] / max(1, v['matrices']), 'max': v['max_sweeps'], 'budget': 16} for c, v in sorted(eig.items())},
                'achieved_tflops': jtf, 'peak_tflops': F32_MFMA_PEAK_TFLOPS, 'frac_of_f32_mfma_peak': jtf / F32_MFMA_PEAK_TFLOPS,
                'flops_note': 'ALGORITHMIC fp32 FLOPs of the tile updates the sweeps executed (two-sided A tiles + V Q), from the sweep '
                              'counts the library reports (wct_eig_stats); the rotation sets themselves run on the VALU/LDS.  Since round 6 the '
                              'A-tile updates of this (batched transform) path run as split-fp16 products on the fp16 MFMA pipe (48 '
                              'v_mfma_f32_16x16x32_f16 per wave and tile where 128 v_mfma_f32_16x16x4_f32 stood), like V Q since round 4: the '
                              'fp32-MFMA peak is kept as the yardstick for comparison with the earlier rounds',
                'bound': 'VALU issue of the rotation sets (one wave per SIMD and pair problem; the chain of a set runs through the '
                         'pivot wave) + the tile updates (split-fp16 MFMA)',
                'note': 'batched two-sided block Jacobi on the %d-level covariances (C = 512, 512, 256, 128, 64; content and style); '
                        'look-ahead launches {pair problems of step s, tile update of step s-1}; from 256 channels on the 64 x 64 pair '
                        'problems are resident in REGISTERS (256 threads, 1 x W strips of cells, rim exchange through LDS, scaled '
                        'rotations: one fma per output, cells as separate S / Q scalars: 60 v_fma per lane and set; four blocks of a launch per CU); V '
                        'resident in registers per launch segment from 24 matrices per solve on (rotation log by LDS-DMA into a ring of 5 tiles; its blocks leave room on their CU for the blocks of the solver); the intra step of a sweep one wave per 32-wide block (odd-even transposition ordering, nothing in LDS during the sets); second-order completion of the '
                        'spectral functions; second-largest time class' % len(LEVELS)}
            line['breakdown_ms_per_step'] = {k: v['ms'] / args.steps for k, v in prof.items()}
        if world == 1 and not args.no_latency and not args.shared_style:
            line.update(latency_leg(ctx, S, args.alpha))
            if 'eigensolver' in line:
                line['eigensolver']['hard_spectrum'] = hard_spectrum_leg(ctx)
            if S == 512:
                real = real_image_leg(ctx, args.alpha)
                if real is not None:
                    line['real_image'] = real
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(S, weights, args.alpha)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == '__main__':
    main()

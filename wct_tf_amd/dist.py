"""Multi-GPU layer of the stylize path: one process per GPU, independent content/style pairs
sharded statically, ONE exchange step -- a gather of the finished uint8 frames to rank 0
(RCCL over xGMI when the tensors live on GPUs, gloo on CPU in the tests).

The reference has no multi-device code at all (single device string, wct.py:17,31); units
are the (content, style) pairs it already processes one at a time (stylize.py:70-100)."""
import os
import subprocess
import sys

import torch
import torch.distributed as dist


def resolve_world(n_gpus, environ=None, share_gpu=False, device_count=None):
    """What `--gpus N` means for this process.  Returns ('launch', N) when N > 1 ranks still have to be started (no
    launcher in the environment: the caller re-executes itself under launch_ranks), or ('rank', rank, world, local_rank)
    when this process IS a rank (N = 1, or a launcher set WORLD_SIZE).  Exits non-zero -- never runs fewer ranks than
    asked for -- when the launcher's WORLD_SIZE differs from N, or when N exceeds the GPUs present and ranks may not
    share one (`share_gpu`: dry runs only)."""
    env = os.environ if environ is None else environ
    if n_gpus < 1:
        raise SystemExit('--gpus %d: need at least one GPU' % n_gpus)
    if 'WORLD_SIZE' in env:
        world = int(env['WORLD_SIZE'])
        if world != n_gpus:
            raise SystemExit('--gpus %d but the launcher set WORLD_SIZE=%d: refusing to run a different number of ranks' % (n_gpus, world))
    if device_count is None:
        device_count = torch.cuda.device_count()
    if n_gpus > device_count and not share_gpu:
        raise SystemExit('--gpus %d but %d GPU(s) visible (WCT_BENCH_SHARE_GPU=1 wraps ranks onto the GPUs for dry runs)' % (n_gpus, device_count))
    if 'WORLD_SIZE' in env:
        rank = int(env.get('RANK', '0'))
        return ('rank', rank, n_gpus, int(env.get('LOCAL_RANK', str(rank))))
    if n_gpus == 1:
        return ('rank', 0, 1, 0)
    return ('launch', n_gpus)


# environment a rank gets unless the caller's environment already says otherwise: policy of the HOST this was built on, kept
# in one visible place (callers pass their own dict to launch_ranks(env_defaults=...) to change it)
RANK_ENV_DEFAULTS = {
    'HSA_ENABLE_IPC_MODE_LEGACY': '0',      # dmabuf IPC: what RCCL needs on this host driver
    'OMP_NUM_THREADS': '8',                 # N ranks on one node: do not let each of them start a thread per core
}


def launch_ranks(n_gpus, argv, script=None, module=None, timeout=None, env_defaults=None):
    """Start N ranks of this program on ONE node, one per GPU, and wait for them: `python -m torch.distributed.run
    --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node N <script | -m module> argv` -- the driver's own
    launch (`--master-addr 127.0.0.1 --master-port P`) with the port left to the launcher: `--standalone` has its c10d
    rendezvous bind port 0 itself, so two self-launched jobs on one node cannot race for a port number picked here
    (bind-close-reuse was a TOCTOU window).  The ranks see the same RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    variables either way.  stdout/stderr pass through (rank 0 prints the result).  Returns the launcher's exit code."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
           '--nproc-per-node', str(n_gpus)]
    cmd += ['-m', module] if module else [script]
    cmd += list(argv)
    env = dict(os.environ)
    for k, v in (RANK_ENV_DEFAULTS if env_defaults is None else env_defaults).items():
        env.setdefault(k, v)
    return subprocess.call(cmd, env=env, timeout=timeout)


def shard_range(n_items, world_size, rank):
    """Contiguous static shard [lo, hi) of `n_items` pairs for `rank`; sizes differ by at
    most one and cover every item exactly once."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError('bad rank/world_size')
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def gather_frames(local_frames, world_size, rank, dst=0, n_items=None):
    """Gather every rank's [b, H, W, 3] uint8 frames on `dst`; returns the [sum b, H, W, 3]
    tensor there (rank order == global pair order of shard_range) and None elsewhere.

    The shard map is static, so every rank knows every shard size: with `n_items` (the global
    number of pairs) given, the exchange is ONE dist.gather and nothing else -- no size
    exchange, no host synchronisation.  Ragged shards pad to the largest.  Without `n_items`
    the sizes are exchanged first (an all_gather and a host read-back: only for callers whose
    shards do not come from shard_range)."""
    if world_size == 1:
        return local_frames
    if n_items is not None:
        sizes = [shard_range(n_items, world_size, r)[1] - shard_range(n_items, world_size, r)[0] for r in range(world_size)]
        if sizes[rank] != local_frames.shape[0]:
            raise ValueError('rank %d holds %d frames, its shard of %d items has %d' % (rank, local_frames.shape[0], n_items, sizes[rank]))
    else:
        b = torch.tensor([local_frames.shape[0]], dtype=torch.int64, device=local_frames.device)
        got = [torch.zeros_like(b) for _ in range(world_size)]
        dist.all_gather(got, b)
        sizes = [int(s.item()) for s in got]
    bmax = max(sizes)
    send = local_frames
    if send.shape[0] != bmax:
        pad = torch.zeros((bmax - send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad], 0)
    send = send.contiguous()
    if rank == dst:
        parts = [torch.empty_like(send) for _ in range(world_size)]
        dist.gather(send, gather_list=parts, dst=dst)
        if all(n == bmax for n in sizes):
            return torch.cat(parts, 0)
        return torch.cat([p[:n] for p, n in zip(parts, sizes)], 0)
    dist.gather(send, gather_list=None, dst=dst)
    return None

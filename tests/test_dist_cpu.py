"""N>1 path on CPU: world_size-2 gloo processes exercising the shard map and the frame gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wct_tf_amd.dist import shard_range, gather_frames


def test_shard_range_covers_everything_once():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_range(n, world, r)
                assert 0 <= lo <= hi <= n
                seen += list(range(lo, hi))
            assert seen == list(range(n))
    assert shard_range(64, 8, 3) == (24, 32)        # config 4: 8 frames per GPU
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q, static):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = shard_range(n_items, world, rank)
    # "stylized frames" of this shard: frame i is filled with value i
    frames = torch.stack([torch.full((4, 6, 3), i, dtype=torch.uint8) for i in range(lo, hi)]) \
        if hi > lo else torch.zeros((0, 4, 6, 3), dtype=torch.uint8)
    out = gather_frames(frames, world, rank, n_items=n_items if static else None)
    if rank == 0:
        q.put(out.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_items,static', [(8, True), (5, True), (5, False)])
def test_gather_frames_world2_gloo(n_items, static):
    """static: the sizes follow from the shard map (one gather, nothing else: what bench.py and the CLI use);
    not static: sizes exchanged first."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q, static)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got.shape == (n_items, 4, 6, 3)
    for i in range(n_items):
        assert np.all(got[i] == i)

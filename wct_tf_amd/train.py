"""`python -m wct_tf_amd.train ...`: the reference's train.py (train.py:12-201) on the MI355X path -- trains the
decoder for one relu target to invert the frozen VGG encoder (model.py:123-223).  Same flags.

What runs where: every optimiser step is ONE call into the library (`wct_train_step`): forward on the inference
kernels, losses (feature MSE, pixel MSE, total variation), fp32 backward, Adam, re-packing of the fp16 forward
weights.  The host only produces batches (random 256x256 crops of images resized to 512, train.py:60-87) on a
loader thread, logs, and writes checkpoints.

Multi-GPU (`torchrun --nproc-per-node N --master-addr 127.0.0.1 -m wct_tf_amd.train ...`): data parallel, one
process per GPU, `--batch-size` images PER GPU.  Every rank computes the gradients of its own batch
(`wct_train_step` with lr = 0), the ranks average the decoder's gradient buffer -- ONE contiguous device buffer, ONE
all-reduce over RCCL per step -- and every rank applies the same Adam step (`wct_train_apply`), so the replicas stay
bit-identical.  Rank 0 logs (losses averaged over ranks) and writes the checkpoints.

Differences from the reference, forced by the environment: no TensorFlow, so no summaries / FIFOQueue / Saver.
Checkpoints are `decoder_<relu>.npz` files in --checkpoint (the layout `WCT(checkpoints=[dir])` reads back) plus a
`train_state.json` with the step counter; `--max-to-keep` numbered snapshots are rotated like the Saver does.
`--synthetic-weights SEED` / `--synthetic-data N` stand in for the absent VGG file / image folder.

Optimiser state: a checkpoint holds the decoder weights and the global step, NOT Adam's moments (tf.train.Saver also
stores the Adam slots).  After a restore -- at start-up or after a failed step -- the moments restart at zero, and so
does the step count of Adam's bias correction (`opt_step` below): with a large global step and zero moments the
correction would be ~1 and the first updates ~3x too large.  The learning-rate decay keeps following the global step.
A fresh decoder starts from seeded He-normal filters with zero-mean kernels and a 0.5 bias on the output conv
(`weights.synthetic_weights(0)`), not from Keras' glorot_uniform / zero bias: there is no Keras here to reproduce
its random stream, and the start only matters for the first few hundred steps.

Data-parallel failures: a rank-local exception inside the local phase of a step (gradients of the rank's own batch)
is agreed on by ALL ranks (MIN all-reduce of an ok flag) before any of them acts, so every rank reloads the same
checkpoint together and the collectives stay paired; a failure inside the collective phase (all-reduce, apply)
cannot be recovered rank-locally and aborts the job.
"""
from __future__ import division, print_function

import argparse
import json
import os
import queue
import threading
import time

import numpy as np

from .context import Context
from .utils import get_files, get_img_random_crop
from .weights import load_weights, save_weights, synthetic_weights, synthetic_image


def build_parser():
    parser = argparse.ArgumentParser()
    # Directories
    parser.add_argument('--checkpoint', type=str, dest='checkpoint', help='Checkpoint save dir', required=True)
    parser.add_argument('--log-path', type=str, dest='log_path', help='Logging dir path')
    parser.add_argument('--relu-target', type=str, required=True, help='Target VGG19 relu layer to decode from, e.g. relu4_1')
    parser.add_argument('--content-path', type=str, dest='content_path', help='Content images folder')
    parser.add_argument('--val-path', type=str, default=None, dest='val_path', help='Validation images folder')
    parser.add_argument('--vgg-path', type=str, dest='vgg_path', help='Path to vgg_normalised.t7 (or a .npz)', default='models/vgg_normalised.t7')
    # Loss weights
    parser.add_argument('--feature-weight', type=float, dest='feature_weight', help='Feature loss weight', default=1)
    parser.add_argument('--pixel-weight', type=float, dest='pixel_weight', help='Pixel reconstruction loss weight', default=1)
    parser.add_argument('--tv-weight', type=float, dest='tv_weight', help='Total variation loss weight', default=0)
    # Train opts
    parser.add_argument('--learning-rate', type=float, dest='learning_rate', help='Learning rate', default=1e-4)
    parser.add_argument('--lr-decay', type=float, dest='lr_decay', help='Learning rate decay', default=0)
    parser.add_argument('--max-iter', type=int, dest='max_iter', help='Max # of training iterations', default=16000)
    parser.add_argument('--batch-size', type=int, dest='batch_size', help='Batch size', default=8)
    parser.add_argument('--save-iter', type=int, dest='save_iter', help='Checkpoint save frequency', default=200)
    parser.add_argument('--summary-iter', type=int, dest='summary_iter', help='Validation loss frequency', default=20)
    parser.add_argument('--max-to-keep', type=int, dest='max_to_keep', help='Max # of checkpoints to keep around', default=10)
    # additions of this path
    parser.add_argument('--device', type=int, default=0)
    parser.add_argument('--crop', type=int, default=256, help='training crop size (reference: 256)')
    parser.add_argument('--synthetic-weights', type=int, default=None, metavar='SEED',
                        help='seeded synthetic encoder instead of --vgg-path')
    parser.add_argument('--synthetic-data', type=int, default=None, metavar='N',
                        help='train on N seeded synthetic images instead of --content-path')
    return parser


def torch_decay(learning_rate, global_step, decay_rate):
    """ops.py:298-309: lr / (1 + step * decay)"""
    return learning_rate / (1.0 + global_step * decay_rate)


def batch_gen(folder, batch_shape, synthetic=None, seed=0):
    """train.py:60-87: resize to 512, random 256 crop, [0,1]; or seeded synthetic images"""
    rng = np.random.default_rng(seed)
    files = None if synthetic else np.asarray(get_files(folder))
    b, h, w, _ = batch_shape
    while True:
        x = np.zeros(batch_shape, np.float32)
        idx = 0
        while idx < b:
            try:
                if synthetic:
                    img = synthetic_image(int(rng.integers(0, synthetic)), h, w)
                else:
                    img = get_img_random_crop(str(rng.choice(files)), resize=2 * h, crop=h)
                x[idx] = np.float32(img) / 255.
                assert not np.isnan(x[idx].min())
            except Exception as e:      # noqa: BLE001  (the reference skips unreadable files the same way)
                print(e)
                continue
            idx += 1
        yield x


def _start_loader(gen, depth=4):
    q = queue.Queue(maxsize=depth)

    def work():
        for x in gen:
            q.put(x)
    t = threading.Thread(target=work, daemon=True)
    t.start()
    return q


def save_checkpoint(ctx, relu_target, directory, step, max_to_keep):
    os.makedirs(directory, exist_ok=True)
    layers = ctx.get_decoder(relu_target)
    w = {'encoder': {}, 'decoder': {relu_target: layers}}
    latest = os.path.join(directory, 'decoder_%s.npz' % relu_target)
    snap = os.path.join(directory, 'decoder_%s-%d.npz' % (relu_target, step))
    save_weights(snap, w)
    save_weights(latest, w)
    state_path = os.path.join(directory, 'train_state.json')
    state = {'step': step, 'snapshots': []}
    if os.path.exists(state_path):
        with open(state_path) as f:
            state['snapshots'] = json.load(f).get('snapshots', [])
    state['snapshots'] = [s for s in state['snapshots'] if s != os.path.basename(snap)] + [os.path.basename(snap)]
    while len(state['snapshots']) > max(1, max_to_keep):
        old = os.path.join(directory, state['snapshots'].pop(0))
        if os.path.exists(old):
            os.remove(old)
    with open(state_path, 'w') as f:
        json.dump(state, f)
    return snap


def load_latest(directory, relu_target):
    """(decoder layers, step) of the newest checkpoint in `directory`, or (None, 0)"""
    latest = os.path.join(directory, 'decoder_%s.npz' % relu_target)
    state_path = os.path.join(directory, 'train_state.json')
    if not (os.path.exists(latest) and os.path.exists(state_path)):
        return None, 0
    with open(state_path) as f:
        step = int(json.load(f).get('step', 0))
    return load_weights(latest)['decoder'][relu_target], step


class _DevArray(object):
    """A device buffer owned by the library, exposed through __cuda_array_interface__ so that torch can wrap it
    without a copy (torch.as_tensor) and hand it to RCCL."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {'shape': (count,), 'typestr': '<f4', 'data': (ptr, False), 'version': 2}


def _dist_setup():
    """(rank, world, torch.distributed or None).  Under torchrun the device follows LOCAL_RANK."""
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world <= 1:
        return 0, 1, None
    import torch
    import torch.distributed as dist
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    if os.environ.get('WCT_TRAIN_SHARE_GPU'):      # dry run on a box with fewer GPUs than ranks
        local %= torch.cuda.device_count()
        os.environ['LOCAL_RANK'] = str(local)
    torch.cuda.set_device(local)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if not dist.is_initialized():
        # 'nccl' = RCCL over xGMI; WCT_TRAIN_BACKEND=gloo stages the all-reduce through the host (dry run)
        dist.init_process_group(os.environ.get('WCT_TRAIN_BACKEND', 'nccl'), rank=rank, world_size=world)
    return rank, world, dist


def train(argv=None):
    args = build_parser().parse_args(argv)
    relu = args.relu_target
    rank, world, dist = _dist_setup()
    if dist is not None:
        args.device = int(os.environ.get('LOCAL_RANK', str(rank)))
    if args.synthetic_weights is not None:
        weights = synthetic_weights(args.synthetic_weights, relu_targets=[relu])
    else:
        if args.vgg_path.endswith('.t7'):
            from .t7 import vgg_weights_from_t7
            enc = vgg_weights_from_t7(args.vgg_path)
        else:
            enc = load_weights(args.vgg_path)['encoder']
        # fresh decoder: He-normal, as Keras' default glorot/he initialisers would give a trainable start
        weights = {'encoder': enc, 'decoder': synthetic_weights(0, relu_targets=[relu])['decoder']}
    restored, step0 = load_latest(args.checkpoint, relu)
    if restored is not None:
        print('Restoring checkpoint (step %d)' % step0)
        weights['decoder'][relu] = restored
    ctx = Context(args.device)
    ctx.set_encoder(weights['encoder'])
    ctx.set_decoder(relu, weights['decoder'][relu])

    shape = (args.batch_size, args.crop, args.crop, 3)
    if args.synthetic_data is None and not args.content_path:
        raise SystemExit('--content-path (or --synthetic-data N) is required')
    train_q = _start_loader(batch_gen(args.content_path, shape, args.synthetic_data, seed=1 + 1000 * rank))
    val_folder = args.val_path if args.val_path is not None else args.content_path
    val_q = _start_loader(batch_gen(val_folder, shape, args.synthetic_data, seed=2 + 1000 * rank))
    log_path = args.log_path if args.log_path is not None else os.path.join(args.checkpoint, 'log')
    log = None
    if rank == 0:
        os.makedirs(log_path, exist_ok=True)
        log = open(os.path.join(log_path, 'train_log.jsonl'), 'a')
    grad = [None]
    if dist is not None:
        import torch

    def bind_grad():
        if dist is not None:
            ptr, count = ctx.train_grad_buffer(relu)
            grad[0] = torch.as_tensor(_DevArray(ptr, count), device='cuda:%d' % args.device)     # a view, no copy
    bind_grad()

    class StepFailed(Exception):
        """a step every rank agreed to abandon (recoverable: reload the latest checkpoint everywhere)"""

    def agree(res, err):
        """data parallel: ONE small all-reduce per step carries the four losses and a failure flag (round 3; the separate
        agreement collective + host sync of round 2 is gone).  Returns the losses averaged over the ranks; every rank
        raises StepFailed together if the local phase failed anywhere -- before the gradients are exchanged."""
        vals = [res['feature_loss'], res['pixel_loss'], res['tv_loss'], res['total_loss']] if res is not None else [0.0] * 4
        t = torch.tensor(vals + [0.0 if err is None else 1.0], dtype=torch.float64,
                         device='cuda:%d' % args.device if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t = t.tolist()
        if t[4] > 0:
            raise StepFailed(str(err) if err is not None else 'another rank failed this step')
        return {'feature_loss': t[0] / world, 'pixel_loss': t[1] / world, 'tv_loss': t[2] / world, 'total_loss': t[3] / world}

    def one_step(x, step, lr):
        """single GPU: one fused call; data parallel: gradients, one all-reduce (average), the same Adam everywhere.
        `step` is the optimiser's own step count (Adam bias correction), see the module docstring."""
        if dist is None:
            try:
                return ctx.train_step(relu, x, step=step, learning_rate=lr, feature_weight=args.feature_weight,
                                      pixel_weight=args.pixel_weight, tv_weight=args.tv_weight)
            except Exception as e:                  # noqa: BLE001
                raise StepFailed(str(e))
        res, err = None, None
        try:                                        # local phase: may fail on one rank only
            res = ctx.train_step(relu, x, step=step, learning_rate=0.0, feature_weight=args.feature_weight,
                                 pixel_weight=args.pixel_weight, tv_weight=args.tv_weight)
        except Exception as e:                      # noqa: BLE001
            err = e
        res = agree(res, err)
        if lr == 0.0:
            return res                              # evaluation only (validation batch): no update
        if dist.get_backend() == 'nccl':
            dist.all_reduce(grad[0], op=dist.ReduceOp.SUM)
            grad[0].div_(world)
        else:                                       # host-staged dry run
            host = grad[0].cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            grad[0].copy_(host / world)
        torch.cuda.synchronize()
        ctx.train_apply(relu, step, lr)             # collective phase: an exception here ends the job (not caught below)
        return res

    step = step0
    opt_step = 0                                    # steps since Adam's moments were last zero (see the module docstring)
    results = None
    for iteration in range(args.max_iter):
        start = time.time()
        x = train_q.get()
        lr = torch_decay(args.learning_rate, step, args.lr_decay)
        step += 1
        opt_step += 1
        try:
            results = one_step(x, opt_step, lr)
            if not np.isfinite(results['total_loss']):     # the loss is averaged over ranks: every rank sees the same value
                raise StepFailed('non-finite loss at step %d' % step)
        except StepFailed as e:                    # train.py:168-174: reload the latest checkpoint and go on -- all ranks together
            print(e)
            print('Exception encountered, re-loading latest checkpoint')
            # rank 0 owns the checkpoint directory (only it writes there): it loads, every rank receives the same
            # weights and step -- on a filesystem that is not shared the ranks would otherwise restore different (or no)
            # files and diverge silently
            if dist is None or rank == 0:
                restored, step_saved = load_latest(args.checkpoint, relu)
            else:
                restored, step_saved = None, None
            if dist is not None:
                box = [restored, step_saved]
                dist.broadcast_object_list(box, src=0)
                restored, step_saved = box
            if restored is None:
                raise
            ctx.set_decoder(relu, restored)        # zero moments: the bias correction restarts with them
            bind_grad()                            # the gradient buffer was re-created with the decoder
            step, opt_step = step_saved, 0
            if dist is not None:
                dist.barrier()
            continue
        rec = dict(results, step=step, lr=lr, time=time.time() - start)
        if iteration % args.summary_iter == 0:          # a validation batch, evaluated without an update
            try:
                val = one_step(val_q.get(), max(opt_step, 1), 0.0)
                rec['val_total_loss'] = val['total_loss']
            except StepFailed as e:                    # evaluation only: nothing to undo, every rank skips it together
                print('validation batch skipped:', e)
        if rank == 0:
            log.write(json.dumps(rec) + '\n')
            log.flush()
            if iteration % args.save_iter == 0:
                print('Model saved in file: %s' % save_checkpoint(ctx, relu, args.checkpoint, step, args.max_to_keep))
            print('Step: {}  LR: {:.7f}  Feature: {:.5f}  Pixel: {:.5f}  TV: {:.5f}  Time: {:.5f}'.format(
                step, lr, results['feature_loss'], results['pixel_loss'], results['tv_loss'], time.time() - start))
    if rank == 0:
        print('Model saved in file: %s' % save_checkpoint(ctx, relu, args.checkpoint, step, args.max_to_keep))
    if dist is not None:
        import hashlib
        digest = hashlib.sha1(b''.join(w.tobytes() + b.tobytes() for w, b in ctx.get_decoder(relu))).hexdigest()
        print('rank %d decoder digest %s' % (rank, digest))       # identical on every rank: the replicas never diverge
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    return results


if __name__ == '__main__':
    train()

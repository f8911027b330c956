"""`python -m wct_tf_amd.stylize ...`: the command line of the reference's stylize.py (stylize.py:14-126) on the
MI355X path.  Flag names, defaults and the output naming `{content}_{style}{ext}` are the reference's; `--checkpoints`
takes TF checkpoint directories or .npz files (see wct.py), `--vgg-path` the .t7 or a .npz, and
`--synthetic-weights SEED` stands in when neither exists.  Under torchrun each rank takes a shard of the content
files (rank_shard); `--gpus N` without a launcher starts the N ranks itself."""
import argparse
import os
import time

import numpy as np

from . import utils
from .wct import WCT

# (flags, keyword arguments) -- the interface of stylize.py:16-37, then the additions of this path
_FLAGS = [
    (('--checkpoints',), dict(nargs='+', default=None, help='one decoder checkpoint (directory or .npz) per relu target')),
    (('--relu-targets',), dict(nargs='+', required=True, help='relu layers to stylize at, in pipeline order')),
    (('--vgg-path',), dict(default='models/vgg_normalised.t7', help='encoder weights: vgg_normalised.t7 or .npz (stylize.py:19 default)')),
    (('--content-path',), dict(dest='content_path', help='content image, or a folder of them')),
    (('--style-path',), dict(dest='style_path', help='style image, or a folder of them')),
    (('--out-path',), dict(dest='out_path', help='folder the results are written to')),
    (('--keep-colors',), dict(action='store_true', default=False, help='CORAL: give the style the colours of the content first')),
    (('--device',), dict(default='/gpu:0', help='e.g. /gpu:0')),
    (('--style-size',), dict(type=int, default=0, help='short side of the style image (0: as is)')),
    (('--crop-size',), dict(type=int, default=0, help='centre-crop the style image to a square of this side (0: no)')),
    (('--content-size',), dict(type=int, default=0, help='short side of the content image (0: as is)')),
    (('--passes',), dict(type=int, default=1, help='feed the result back in this many times')),
    (('-r', '--random'), dict(type=int, default=0, help='use this many randomly chosen styles of the style folder')),
    (('--alpha',), dict(type=float, default=1, help='style strength: blend of transformed and content features')),
    (('--concat',), dict(action='store_true', default=False, help='put the style image to the left of every result')),
    (('--adain',), dict(action='store_true', default=False, help='AdaIN instead of WCT at every level')),
    (('--swap5',), dict(action='store_true', default=False, help='style-swap at relu5_1')),
    (('--ss-alpha',), dict(type=float, default=0.6, help='style-swap blend')),
    (('--ss-patch-size',), dict(type=int, default=3, help='style-swap patch size')),
    (('--ss-stride',), dict(type=int, default=1, help='style-swap stride')),
    (('--synthetic-weights',), dict(type=int, default=None, metavar='SEED', help='seeded synthetic weights instead of files')),
    (('--wct-mode',), dict(choices=['tf', 'np'], default='tf', help='wct_tf (the graph) or wct_np semantics')),
    (('--gpus',), dict(type=int, default=0, metavar='N',
                       help='shard the content files over N GPUs of this node, one process per GPU: started here when no '
                            'launcher did (0: whatever the launcher set, else one GPU)')),
]


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    for names, kw in _FLAGS:
        parser.add_argument(*names, **kw)
    return parser


def rank_shard(items, environ=None):
    """One process per GPU (`torchrun --nproc-per-node N -m wct_tf_amd.stylize ...`): rank r of WORLD_SIZE takes a
    contiguous shard of the content files and writes its own outputs -- independent pairs, no collective.
    Returns (shard, device or None): the device string follows LOCAL_RANK when the launcher set it."""
    env = os.environ if environ is None else environ
    world, rank = int(env.get('WORLD_SIZE', '1')), int(env.get('RANK', '0'))
    if world <= 1:
        return list(items), None
    from .dist import shard_range
    lo, hi = shard_range(len(items), world, rank)
    local = int(env.get('LOCAL_RANK', str(rank)))
    if env.get('WCT_BENCH_SHARE_GPU'):                       # dry run on a box with fewer GPUs than ranks
        import torch
        local %= max(1, torch.cuda.device_count())
    return list(items)[lo:hi], '/gpu:%d' % local


def _listing(path):
    return utils.get_files(path) if os.path.isdir(path) else [path]


def _stem(path):
    return os.path.splitext(os.path.basename(path))[0]


def load_style(path, args):
    """style image after --style-size / --crop-size (stylize.py:76-83)"""
    img = utils.get_img(path)
    if args.style_size > 0:
        img = utils.resize_to(img, args.style_size)
    if args.crop_size > 0:
        img = utils.center_crop(img, args.crop_size)
    return img


def stylize_pair(model, content, style, args):
    """one output image: optional CORAL, `--passes` predictions, optional `--concat` (stylize.py:85-110)"""
    if args.keep_colors:
        from .ops import preserve_colors_np
        style = preserve_colors_np(style, content, ctx=model.sess)
    out = content
    for _ in range(max(1, args.passes)):
        out = model.predict(out, style, args.alpha, args.swap5, args.ss_alpha, args.adain)
    if args.concat:
        side = out.shape[0]
        out = np.hstack([utils._imresize(style, (side, side)), out])
    return out


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    if args.synthetic_weights is None and not args.checkpoints:
        parser.error('--checkpoints is required (stylize.py:17) unless --synthetic-weights SEED is given')
    if args.gpus > 0:
        # --gpus N means N ranks: agree with a launcher's WORLD_SIZE, or start the ranks from here (dist.resolve_world)
        from .dist import resolve_world, launch_ranks
        import sys
        role = resolve_world(args.gpus, share_gpu=bool(os.environ.get('WCT_BENCH_SHARE_GPU')))
        if role[0] == 'launch':
            rc = launch_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv), module='wct_tf_amd.stylize')
            if rc:
                raise SystemExit(rc)
            return None
    t0 = time.time()
    weights = None
    if args.synthetic_weights is not None:
        from .weights import synthetic_weights
        weights = synthetic_weights(args.synthetic_weights, relu_targets=args.relu_targets)
    contents, device = rank_shard(sorted(_listing(args.content_path)))
    model = WCT(checkpoints=args.checkpoints, relu_targets=args.relu_targets, vgg_path=args.vgg_path,
                device=device or args.device, ss_patch_size=args.ss_patch_size, ss_stride=args.ss_stride,
                weights=weights, wct_mode=args.wct_mode)
    styles = _listing(args.style_path)
    if os.path.isdir(args.style_path) and args.random > 0:
        styles = list(np.random.choice(styles, args.random))
    os.makedirs(args.out_path, exist_ok=True)
    written = 0
    for cpath in contents:
        content = utils.get_img(cpath)
        if args.content_size > 0:
            content = utils.resize_to(content, args.content_size)
        for spath in styles:
            result = stylize_pair(model, content, load_style(spath, args), args)
            target = os.path.join(args.out_path, '%s_%s%s' % (_stem(cpath), _stem(spath), os.path.splitext(cpath)[1]))
            utils.save_img(target, result)
            written += 1
            print('%d: wrote %s' % (written, target))
    print('%d outputs in %.1f s' % (written, time.time() - t0))
    return written


if __name__ == '__main__':
    main()

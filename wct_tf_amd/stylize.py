"""`python -m wct_tf_amd.stylize ...`: the reference's stylize.py CLI (stylize.py:14-126) on the MI355X
path.  Same flags and output naming ({content}_{style}{ext}); `--checkpoints` / `--vgg-path` take .npz
files written by wct_tf_amd.weights.save_weights (see wct.py), or `--synthetic-weights SEED` stands in
for the absent pre-trained files."""
from __future__ import division, print_function

import argparse
import os
import time

import numpy as np

from .utils import get_files, get_img, save_img, resize_to, center_crop, _imresize
from .wct import WCT


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--checkpoints', nargs='+', type=str, help='List of decoder weight files/dirs', default=None)
    parser.add_argument('--relu-targets', nargs='+', type=str, help='List of reluX_1 layers, corresponding to --checkpoints', required=True)
    parser.add_argument('--vgg-path', type=str, help='Path to the encoder weights (.npz)', default=None)
    parser.add_argument('--content-path', type=str, dest='content_path', help='Content image or folder of images')
    parser.add_argument('--style-path', type=str, dest='style_path', help='Style image or folder of images')
    parser.add_argument('--out-path', type=str, dest='out_path', help='Output folder path')
    parser.add_argument('--keep-colors', action='store_true', help="Preserve the colors of the style image", default=False)
    parser.add_argument('--device', type=str, help='Device to perform compute on, e.g. /gpu:0', default='/gpu:0')
    parser.add_argument('--style-size', type=int, help="Resize style image to this size before cropping", default=0)
    parser.add_argument('--crop-size', type=int, help="Crop square size", default=0)
    parser.add_argument('--content-size', type=int, help="Resize short side of content image to this", default=0)
    parser.add_argument('--passes', type=int, help="# of stylization passes per content image", default=1)
    parser.add_argument('-r', '--random', type=int, help="Choose # of random subset of images from style folder", default=0)
    parser.add_argument('--alpha', type=float, help="Alpha blend value", default=1)
    parser.add_argument('--concat', action='store_true', help="Concatenate style image and stylized output", default=False)
    parser.add_argument('--adain', action='store_true', help="Use AdaIN instead of WCT", default=False)
    # Style swap args
    parser.add_argument('--swap5', action='store_true', help="Swap style on layer relu5_1", default=False)
    parser.add_argument('--ss-alpha', type=float, help="Style swap alpha blend", default=0.6)
    parser.add_argument('--ss-patch-size', type=int, help="Style swap patch size", default=3)
    parser.add_argument('--ss-stride', type=int, help="Style swap stride", default=1)
    # additions of this path
    parser.add_argument('--synthetic-weights', type=int, default=None, metavar='SEED',
                        help='use seeded synthetic weights instead of --checkpoints/--vgg-path')
    parser.add_argument('--wct-mode', choices=['tf', 'np'], default='tf', help='wct_tf (graph) or wct_np semantics')
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
    return list(items)[lo:hi], '/gpu:%d' % int(env.get('LOCAL_RANK', str(rank)))


def main(argv=None):
    args = build_parser().parse_args(argv)
    start = time.time()

    weights = None
    if args.synthetic_weights is not None:
        from .weights import synthetic_weights
        weights = synthetic_weights(args.synthetic_weights, relu_targets=args.relu_targets)
    _, rank_device = rank_shard([])
    wct_model = WCT(checkpoints=args.checkpoints, relu_targets=args.relu_targets, vgg_path=args.vgg_path,
                    device=rank_device or args.device, ss_patch_size=args.ss_patch_size, ss_stride=args.ss_stride,
                    weights=weights, wct_mode=args.wct_mode)

    content_files = get_files(args.content_path) if os.path.isdir(args.content_path) else [args.content_path]
    content_files, _ = rank_shard(sorted(content_files))
    if os.path.isdir(args.style_path):
        style_files = get_files(args.style_path)
        if args.random > 0:
            style_files = np.random.choice(style_files, args.random)
    else:
        style_files = [args.style_path]

    os.makedirs(args.out_path, exist_ok=True)
    count = 0
    for content_fullpath in content_files:
        content_prefix, content_ext = os.path.splitext(content_fullpath)
        content_prefix = os.path.basename(content_prefix)
        content_img = get_img(content_fullpath)
        if args.content_size > 0:
            content_img = resize_to(content_img, args.content_size)

        for style_fullpath in style_files:
            style_prefix, _ = os.path.splitext(style_fullpath)
            style_prefix = os.path.basename(style_prefix)
            style_img = get_img(style_fullpath)
            if args.style_size > 0:
                style_img = resize_to(style_img, args.style_size)
            if args.crop_size > 0:
                style_img = center_crop(style_img, args.crop_size)
            if args.keep_colors:
                from .ops import preserve_colors_np
                style_img = preserve_colors_np(style_img, content_img, ctx=wct_model.sess)

            stylized_rgb = wct_model.predict(content_img, style_img, args.alpha, args.swap5, args.ss_alpha, args.adain)
            for _ in range(args.passes - 1):
                stylized_rgb = wct_model.predict(stylized_rgb, style_img, args.alpha, args.swap5, args.ss_alpha, args.adain)

            if args.concat:
                style_img_resized = _imresize(style_img, (stylized_rgb.shape[0], stylized_rgb.shape[0]))
                stylized_rgb = np.hstack([style_img_resized, stylized_rgb])

            out_f = os.path.join(args.out_path, '{}_{}{}'.format(content_prefix, style_prefix, content_ext))
            save_img(out_f, stylized_rgb)
            count += 1
            print("{}: Wrote stylized output image to {}".format(count, out_f))

    print("Finished stylizing {} outputs in {}s".format(count, time.time() - start))
    return count


if __name__ == '__main__':
    main()

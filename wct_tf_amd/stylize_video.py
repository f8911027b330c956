"""`python -m wct_tf_amd.stylize_video ...`: the reference's stylize_video.py (stylize_video.py:16-160) on the
MI355X path.  Same flags.  Differences, all forced by the environment or by the hardware:

* `--in-path` may be a DIRECTORY of frames (ffmpeg is not in this image).  A video file works
  when an `ffmpeg` binary is on PATH: frames are extracted to `--tmp-dir` and the stylized frames are
  re-encoded, exactly as the reference does (stylize_video.py:73-80,136-150).  Without ffmpeg the stylized
  frames are left in `<out-path>/<video>_<style>/`.
* The reference calls `predict()` once per frame, which re-runs the style encoder, the style statistics and
  the style eigendecompositions for every frame (stylize_video.py:112-121).  Here the frames of a video go
  through `WCT.predict_frames` in batches that share ONE style: the style side runs once per batch
  (WCT_FLAG_STYLE_SHARED) and every frame is bit-identical to `predict(frame, style)`.  With
  `--keep-colors` the style image is CORAL-matched to every frame (stylize_video.py:116-119) and therefore
  differs per frame: those frames go through the per-pair batch instead.
"""
from __future__ import division, print_function

import argparse
import os
import random
import re
import shutil
import subprocess
import time

import numpy as np

from .utils import get_files, get_img, save_img, resize_to, center_crop, _imresize
from .wct import WCT

TMP_DIR = '_____fns_frames_%s/' % random.randint(0, 99999)


# (flags, keyword arguments): the interface of stylize_video.py:18-41, then the additions of this path
_FLAGS = [
    (('--checkpoints',), dict(nargs='+', default=None, help='one decoder checkpoint (directory or .npz) per relu target')),
    (('--relu-targets',), dict(nargs='+', required=True, help='relu layers to stylize at, in pipeline order')),
    (('--vgg-path',), dict(default='models/vgg_normalised.t7', help='encoder weights: vgg_normalised.t7 or .npz (stylize.py:19 default)')),
    (('--in-path',), dict(required=True, help='a video file (needs ffmpeg on PATH) or a directory of frames')),
    (('--out-path',), dict(required=True, help='folder the results are written to')),
    (('--style-path',), dict(required=True, help='style image, or a folder of them (one output per style)')),
    (('--tmp-dir',), dict(dest='tmp_dir', default=TMP_DIR, help='scratch folder for extracted / stylized frames')),
    (('--keep-tmp',), dict(action='store_true', default=False, help='leave the scratch folder in place')),
    (('--keep-colors',), dict(action='store_true', default=False, help='CORAL: give the style the colours of each frame first')),
    (('--style-size',), dict(type=int, default=0, help='short side of the style image (0: as is)')),
    (('--crop-size',), dict(type=int, default=0, help='centre-crop the style image to a square of this side (0: no)')),
    (('--content-size',), dict(type=int, default=0, help='short side of every frame (0: as is)')),
    (('--passes',), dict(type=int, default=1, help='feed the result back in this many times')),
    (('--device',), dict(default='/gpu:0', help='e.g. /gpu:0')),
    (('--alpha',), dict(type=float, default=1, help='style strength')),
    (('--concat',), dict(action='store_true', default=False, help='put the style image to the left of every frame')),
    (('--swap5',), dict(action='store_true', default=False, help='style-swap at relu5_1')),
    (('--ss-alpha',), dict(type=float, default=0.6, help='style-swap blend')),
    (('--ss-patch-size',), dict(type=int, default=3, help='style-swap patch size')),
    (('--ss-stride',), dict(type=int, default=1, help='style-swap stride')),
    (('--adain',), dict(action='store_true', default=False, help='AdaIN instead of WCT at every level')),
    (('--batch',), dict(type=int, default=16, help='frames per device batch (<= 32)')),
    (('--fps',), dict(type=int, default=30, help='frame rate of the re-encoded video (reference: 30)')),
    (('--synthetic-weights',), dict(type=int, default=None, metavar='SEED', help='seeded synthetic weights instead of files')),
    (('--wct-mode',), dict(choices=['tf', 'np'], default='tf', help='wct_tf (the graph) or wct_np semantics')),
]


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    for names, kw in _FLAGS:
        parser.add_argument(*names, **kw)
    return parser


def natural_key(path):
    """frame_2.png before frame_10.png (ffmpeg's %d numbering, stylize_video.py:76)."""
    return [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', os.path.basename(path))]


def list_frames(in_dir):
    return sorted(get_files(in_dir), key=natural_key)


def stylize_frames(wct_model, frame_files, style_img, args):
    """Yield (frame_file, stylized uint8 image) in order; consecutive same-sized frames are batched."""
    def load(f):
        img = get_img(f)
        if args.content_size > 0:
            img = resize_to(img, args.content_size)
        return img

    def run(frames, style):
        out = wct_model.predict_frames(frames, style, args.alpha, args.swap5, args.ss_alpha, args.adain, batch=args.batch)
        for _ in range(args.passes - 1):                      # later passes: plain WCT, as stylize_video.py:124-126
            out = wct_model.predict_frames(out, style, args.alpha, adain=args.adain, batch=args.batch)
        return out

    i = 0
    while i < len(frame_files):
        first = load(frame_files[i])
        group_files, group = [frame_files[i]], [first]
        i += 1
        while i < len(frame_files) and len(group) < args.batch:
            nxt = load(frame_files[i])
            if nxt.shape != first.shape:
                break
            group_files.append(frame_files[i])
            group.append(nxt)
            i += 1
        frames = np.stack(group)
        if args.keep_colors:
            # the style differs per frame (CORAL towards each frame): per-pair batch
            from .ops import preserve_colors_np
            styles = np.stack([preserve_colors_np(style_img, f, ctx=wct_model.sess) for f in group])
            outs = wct_model.sess.stylize_batch(frames, styles, wct_model.relu_targets, alpha=args.alpha,
                                                adain=args.adain, wct_mode=wct_model.wct_mode)
            for _ in range(args.passes - 1):
                outs = wct_model.sess.stylize_batch(outs, styles, wct_model.relu_targets, alpha=args.alpha,
                                                    adain=args.adain, wct_mode=wct_model.wct_mode)
            style_per_frame = list(styles)
        else:
            outs = run(frames, style_img)
            style_per_frame = [style_img] * len(group)
        for f, o, s in zip(group_files, outs, style_per_frame):
            if args.concat:                                   # stylize_video.py:129-132
                o = np.hstack([_imresize(s, (o.shape[0], o.shape[0])), o])
            yield f, o


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    if args.synthetic_weights is None and not args.checkpoints:
        parser.error('--checkpoints is required (stylize.py:17) unless --synthetic-weights SEED is given')
    start = time.time()
    weights = None
    if args.synthetic_weights is not None:
        from .weights import synthetic_weights
        weights = synthetic_weights(args.synthetic_weights, relu_targets=args.relu_targets)
    wct_model = WCT(checkpoints=args.checkpoints, relu_targets=args.relu_targets, vgg_path=args.vgg_path,
                    device=args.device, ss_patch_size=args.ss_patch_size, ss_stride=args.ss_stride,
                    weights=weights, wct_mode=args.wct_mode)
    if args.keep_colors and args.swap5:
        raise SystemExit('--keep-colors with --swap5 is not batched here: use stylize.py per frame')

    ffmpeg = shutil.which('ffmpeg')
    is_video = not os.path.isdir(args.in_path)
    if is_video and ffmpeg is None:
        raise SystemExit('--in-path is a file and there is no ffmpeg on PATH: pass a directory of frames')
    in_dir = args.in_path
    if is_video:
        in_dir = os.path.join(args.tmp_dir, 'input')
        os.makedirs(in_dir, exist_ok=True)
        subprocess.check_call([ffmpeg, '-i', args.in_path, '%s/frame_%%d.png' % in_dir])
    frame_files = list_frames(in_dir)
    style_files = get_files(args.style_path) if os.path.isdir(args.style_path) else [args.style_path]
    os.makedirs(args.out_path, exist_ok=True)

    content_prefix, content_ext = os.path.splitext(os.path.basename(os.path.normpath(args.in_path)))
    count = 0
    for style_fullpath in style_files:
        style_img = get_img(style_fullpath)
        if args.style_size > 0:
            style_img = resize_to(style_img, args.style_size)
        if args.crop_size > 0:
            style_img = center_crop(style_img, args.crop_size)
        style_prefix = os.path.splitext(os.path.basename(style_fullpath))[0]
        out_dir = os.path.join(args.tmp_dir, 'sytlized') if is_video else \
            os.path.join(args.out_path, '{}_{}'.format(content_prefix, style_prefix))
        os.makedirs(out_dir, exist_ok=True)
        for f, stylized in stylize_frames(wct_model, frame_files, style_img, args):
            out_f = os.path.join(out_dir, os.path.basename(f))
            save_img(out_f, stylized)
            count += 1
        if is_video:
            out_v = os.path.join(args.out_path, '{}_{}{}'.format(content_prefix, style_prefix, content_ext))
            subprocess.check_call([ffmpeg, '-i', '%s/frame_%%d.png' % out_dir, '-f', 'mp4', '-q:v', '0', '-vcodec', 'mpeg4',
                                   '-r', str(args.fps), '-y', out_v])
            print('Video at: %s' % out_v)
            if not args.keep_tmp:
                shutil.rmtree(out_dir)
    if is_video and not args.keep_tmp:
        shutil.rmtree(args.tmp_dir, ignore_errors=True)
    dt = time.time() - start
    print('Finished stylizing {} frames in {:.1f}s ({:.1f} frames/s incl. image I/O)'.format(count, dt, count / max(dt, 1e-9)))
    return count


if __name__ == '__main__':
    main()

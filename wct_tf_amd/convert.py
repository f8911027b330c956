"""`python -m wct_tf_amd.convert --vgg-path vgg_normalised.t7 --checkpoints DIR... --relu-targets ... --out weights.npz`

Reads the reference's own weight files -- the Torch7 encoder (vgg_normalised.py:16) and the TensorFlow decoder
checkpoint directories (wct.py:46-58) -- with the from-scratch readers of this package (no torchfile / TensorFlow
import) and writes the flat `.npz` container of wct_tf_amd.weights (one file for everything, or per-decoder files that
`WCT(checkpoints=[...])` picks up as `decoder_<relu>.npz`)."""
import argparse
import os

from .weights import save_weights


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--vgg-path', type=str, default=None, help='vgg_normalised.t7')
    p.add_argument('--checkpoints', nargs='*', default=[], help='decoder checkpoint dirs, one per --relu-targets entry')
    p.add_argument('--relu-targets', nargs='*', default=[])
    p.add_argument('--out', type=str, required=True, help='output .npz (or a directory for per-decoder files)')
    return p


def convert(vgg_path=None, checkpoints=(), relu_targets=(), verify_crc=False):
    weights = {'encoder': {}, 'decoder': {}}
    if vgg_path:
        from .t7 import vgg_weights_from_t7
        weights['encoder'] = vgg_weights_from_t7(vgg_path)
    if len(checkpoints) != len(relu_targets):
        raise ValueError('--checkpoints and --relu-targets must pair up')
    from .tf_ckpt import decoder_weights_from_checkpoint
    for relu, ck in zip(relu_targets, checkpoints):
        weights['decoder'][relu] = decoder_weights_from_checkpoint(ck, relu, verify_crc)
    return weights


def main(argv=None):
    args = build_parser().parse_args(argv)
    w = convert(args.vgg_path, args.checkpoints, args.relu_targets)
    if args.out.endswith('.npz'):
        save_weights(args.out, w)
        print('wrote', args.out)
    else:
        os.makedirs(args.out, exist_ok=True)
        if w['encoder']:
            save_weights(os.path.join(args.out, 'vgg_normalised.npz'), {'encoder': w['encoder'], 'decoder': {}})
        for relu, layers in w['decoder'].items():
            save_weights(os.path.join(args.out, 'decoder_%s.npz' % relu), {'encoder': {}, 'decoder': {relu: layers}})
        print('wrote', args.out)
    return w


if __name__ == '__main__':
    main()

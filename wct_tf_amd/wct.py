"""WCT: the inference facade of wct.py:14-106 on the MI355X path.

    WCT(checkpoints, relu_targets, vgg_path, device='/gpu:0', ss_patch_size=3, ss_stride=1)
    WCT.predict(content, style, alpha=1, swap5=False, ss_alpha=1, adain=False) -> uint8 HxWx3

`checkpoints` / `vgg_path`: the reference restores TF checkpoints and a .t7 file
(wct.py:46-58, vgg_normalised.py:16).  Per decoder this class takes either of
  * a TF checkpoint directory (a `checkpoint` state file + V2 bundle, read by
    wct_tf_amd/tf_ckpt.py without TensorFlow) -- the reference's own format,
  * a .npz written by `wct_tf_amd.weights.save_weights` (file, or `decoder_<relu>.npz` in a directory),
or a weights dict via `weights=`; `vgg_path` is the reference's `.t7` or a .npz.  A missing
decoder raises like the reference does (wct.py:58).
"""
import os
import re

import numpy as np

from .context import Context
from .model import WCTModel
from .weights import load_weights


def _device_index(device):
    if isinstance(device, int):
        return device
    m = re.search(r'(\d+)\s*$', str(device))        # '/gpu:0' -> 0
    return int(m.group(1)) if m else 0


class WCT(object):
    '''Stylize images with the WCT model on one MI355X'''

    def __init__(self, checkpoints, relu_targets, vgg_path, device='/gpu:0',
                 ss_patch_size=3, ss_stride=1, weights=None, wct_mode='tf'):
        self.ss_patch_size = ss_patch_size
        self.ss_stride = ss_stride
        self.relu_targets = list(relu_targets)
        self.wct_mode = wct_mode
        self.model = WCTModel(mode='test', relu_targets=relu_targets, vgg_path=vgg_path,
                              ss_patch_size=ss_patch_size, ss_stride=ss_stride)
        self.content_input = self.model.content_input
        self.decoded_output = self.model.decoded_output
        self.sess = Context(_device_index(device))

        if weights is None:
            weights = {'encoder': None, 'decoder': {}}
            if vgg_path is not None:
                if not os.path.exists(vgg_path):
                    raise Exception('No VGG weights found at {}'.format(vgg_path))
                if vgg_path.endswith('.t7'):                 # the reference's own format (vgg_normalised.py:16)
                    from .t7 import vgg_weights_from_t7
                    weights['encoder'] = vgg_weights_from_t7(vgg_path)
                else:
                    weights['encoder'] = load_weights(vgg_path)['encoder']
            for relu_target, checkpoint_dir in zip(relu_targets, checkpoints or []):
                path = checkpoint_dir
                if os.path.isdir(path) and os.path.exists(os.path.join(path, 'checkpoint')):
                    from .tf_ckpt import decoder_weights_from_checkpoint     # tf.train.Saver layout (wct.py:46-58)
                    weights['decoder'][relu_target] = decoder_weights_from_checkpoint(path, relu_target)
                    continue
                if os.path.isdir(path):
                    path = os.path.join(path, 'decoder_{}.npz'.format(relu_target))
                if not os.path.exists(path):
                    raise Exception('No checkpoint found for target {} in dir {}'.format(relu_target, checkpoint_dir))
                dec = load_weights(path)['decoder']
                if relu_target not in dec:
                    raise Exception('No checkpoint found for target {} in dir {}'.format(relu_target, checkpoint_dir))
                weights['decoder'][relu_target] = dec[relu_target]
        if weights.get('encoder') is None:
            raise Exception('No VGG weights given')
        self.sess.set_encoder(weights['encoder'])
        for relu_target in relu_targets:
            if relu_target not in weights['decoder']:
                raise Exception('No checkpoint found for target {}'.format(relu_target))
            self.sess.set_decoder(relu_target, weights['decoder'][relu_target])

    @staticmethod
    def preprocess(image):
        if len(image.shape) == 3:
            image = np.expand_dims(image, 0)
        return image / 255.

    @staticmethod
    def postprocess(image):
        return np.uint8(np.clip(image, 0, 1) * 255)

    def predict(self, content, style, alpha=1, swap5=False, ss_alpha=1, adain=False):
        '''Stylize a single content/style pair; arrays in [0,255], returns uint8 HxWx3.
           The /255 preprocess and the clip*255 postprocess run inside the library
           (fused at the ends of the kernel chain).'''
        content = np.asarray(content)
        style = np.asarray(style)
        # If doing style swap and stride > 1 the content might need to be resized for the filter to fit
        if swap5 is True and self.ss_stride != 1:
            from .utils import swap_filter_fit, center_crop_to
            should_refit, H, W = swap_filter_fit(content.shape[0], content.shape[1], self.ss_patch_size, self.ss_stride)
            if should_refit:
                content = center_crop_to(content, H, W)
        if swap5:
            self.sess.set_style_swap(ss_alpha, self.ss_patch_size, self.ss_stride)
        # uint8 arrays take the fused /255 on the device; float arrays are divided by 255 WITHOUT rounding, as the
        # reference's preprocess does (wct.py:60-64) -- Context.stylize hands them over as float32 images
        return self.sess.stylize(content, style, self.relu_targets, alpha=alpha, adain=adain,
                                 wct_mode=self.wct_mode, swap5=bool(swap5))

    def predict_frames(self, frames, style, alpha=1, swap5=False, ss_alpha=1, adain=False, batch=16):
        '''Stylize same-sized frames [F][H][W][3] with ONE style image (the loop of stylize_video.py:112-121,
           which calls predict() once per frame and so re-runs the style encoder, the style statistics and the
           style eigendecompositions every frame).  Here the style side runs once per batch of `batch` frames;
           every frame equals predict(frame, style) bit for bit.  Returns uint8 [F][Ho][Wo][3].'''
        frames = np.asarray(frames)
        style = np.asarray(style)
        assert frames.ndim == 4 and style.ndim == 3
        if frames.dtype != np.uint8:
            frames = np.uint8(np.clip(frames, 0, 255))
        if style.dtype != np.uint8:
            style = np.uint8(np.clip(style, 0, 255))
        if swap5 is True and self.ss_stride != 1:
            from .utils import swap_filter_fit, center_crop_to
            should_refit, H, W = swap_filter_fit(frames.shape[1], frames.shape[2], self.ss_patch_size, self.ss_stride)
            if should_refit:
                frames = np.stack([center_crop_to(f, H, W) for f in frames])
        if swap5:
            self.sess.set_style_swap(ss_alpha, self.ss_patch_size, self.ss_stride)
        batch = max(1, min(32, int(batch)))
        outs = [self.sess.stylize_batch(frames[i:i + batch], style, self.relu_targets, alpha=alpha, adain=adain,
                                        wct_mode=self.wct_mode, swap5=bool(swap5))
                for i in range(0, len(frames), batch)]
        return np.concatenate(outs, axis=0)

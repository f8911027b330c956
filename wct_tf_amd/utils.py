"""Host-side image helpers with the reference's names (utils.py:11-85), on Pillow instead of the
removed scipy.misc functions.  Not kernel targets: file I/O and resizing around `predict`."""
import os
import random

import numpy as np
from PIL import Image


def get_files(img_dir):
    return [os.path.join(img_dir, x) for x in sorted(os.listdir(img_dir))]


def save_img(out_path, img):
    img = np.clip(img, 0, 255).astype(np.uint8)
    Image.fromarray(img).save(out_path)


def get_img(src):
    img = np.asarray(Image.open(src).convert('RGB'))
    return img


def _imresize(img, shape_hw):
    """scipy.misc.imresize(img, (h, w, 3), interp='bilinear') stand-in."""
    h, w = int(shape_hw[0]), int(shape_hw[1])
    return np.asarray(Image.fromarray(np.uint8(img)).resize((w, h), Image.BILINEAR))


def center_crop_to(img, H_target, W_target):
    '''Center crop a rectangle of given dimensions and resize if necessary (utils.py:40-53)'''
    height, width = img.shape[0], img.shape[1]
    if height < H_target or width < W_target:
        rat = max(H_target / height, W_target / width)
        img = _imresize(img, (int(round(height * rat)), int(round(width * rat))))
        height, width = img.shape[0], img.shape[1]
    h_off = (height - H_target) // 2
    w_off = (width - W_target) // 2
    return img[h_off:h_off + H_target, w_off:w_off + W_target]


def swap_filter_fit(H, W, patch_size, stride, n_pools=4):
    '''Style swap may not output same size encoding if filter size > 1, calculate a new size to avoid
       this (utils.py:115-138): returns (should_refit, H_out, W_out)'''
    H_pool_out, W_pool_out = H, W
    for _ in range(n_pools):
        H_pool_out, W_pool_out = (H_pool_out + 1) // 2, (W_pool_out + 1) // 2
    H_conv_out = (H_pool_out - patch_size) // stride + 1
    W_conv_out = (W_pool_out - patch_size) // stride + 1
    H_deconv_out = (H_conv_out - 1) * stride + patch_size
    W_deconv_out = (W_conv_out - 1) * stride + patch_size
    H_out = H_deconv_out * 2 ** n_pools
    W_out = W_deconv_out * 2 ** n_pools
    should_refit = (H_pool_out != H_deconv_out) or (W_pool_out != W_deconv_out)
    return should_refit, H_out, W_out


def resize_to(img, resize=512):
    '''Resize short side to target size and preserve aspect ratio (utils.py:55-67)'''
    height, width = img.shape[0], img.shape[1]
    if height < width:
        ratio = height / resize
        long_side = round(width / ratio)
        resize_shape = (resize, long_side, 3)
    else:
        ratio = width / resize
        long_side = round(height / ratio)
        resize_shape = (long_side, resize, 3)
    return _imresize(img, resize_shape)


def center_crop(img, size=256):
    '''utils.py:27-38: upscale if a side is too small, then centre crop a square'''
    height, width = img.shape[0], img.shape[1]
    if height < size or width < size:
        img = resize_to(img, resize=size)
        height, width = img.shape[0], img.shape[1]
    h_off = (height - size) // 2
    w_off = (width - size) // 2
    return img[h_off:h_off + size, w_off:w_off + size]


def get_img_crop(src, resize=512, crop=256):
    return center_crop(resize_to(get_img(src), resize), crop)


def get_img_random_crop(src, resize=512, crop=256):
    img = resize_to(get_img(src), resize=resize)
    offset_h = random.randint(0, (img.shape[0] - crop))
    offset_w = random.randint(0, (img.shape[1] - crop))
    return img[offset_h:offset_h + crop, offset_w:offset_w + crop, :]


def preserve_colors_np(style_rgb, content_rgb):
    """utils.py:87-90 on the GPU path (see ops.preserve_colors_np)."""
    from .ops import preserve_colors_np as _p
    return _p(style_rgb, content_rgb)

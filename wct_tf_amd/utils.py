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

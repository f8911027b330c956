"""Host-side image helpers under the reference's names (utils.py:11-138), on Pillow (the scipy.misc functions the
reference calls no longer exist).  File I/O and resizing around `predict`: not kernel targets."""
import os
import random

import numpy as np
from PIL import Image


def get_files(img_dir):
    """paths of a folder's entries, sorted (utils.py:11-13)"""
    return [os.path.join(img_dir, name) for name in sorted(os.listdir(img_dir))]


def get_img(src):
    """RGB uint8 array of an image file (utils.py:19-25)"""
    with Image.open(src) as im:
        return np.array(im.convert('RGB'))


def save_img(out_path, img):
    Image.fromarray(np.uint8(np.clip(img, 0, 255))).save(out_path)


def _imresize(img, shape_hw):
    """bilinear resize to (h, w): what scipy.misc.imresize(img, (h, w, 3)) did for the reference"""
    h, w = (int(v) for v in shape_hw[:2])
    return np.array(Image.fromarray(np.uint8(img)).resize((w, h), Image.BILINEAR))


def _centre_window(img, h, w):
    top, left = (img.shape[0] - h) // 2, (img.shape[1] - w) // 2
    return img[top:top + h, left:left + w]


def resize_to(img, resize=512):
    """short side -> `resize`, aspect kept (utils.py:55-67)"""
    h, w = img.shape[:2]
    scale = resize / min(h, w)
    if h < w:
        return _imresize(img, (resize, round(w * scale)))
    return _imresize(img, (round(h * scale), resize))


def center_crop(img, size=256):
    """square centre crop; an image with a side below `size` is scaled up first (utils.py:27-38)"""
    if min(img.shape[:2]) < size:
        img = resize_to(img, resize=size)
    return _centre_window(img, size, size)


def center_crop_to(img, H_target, W_target):
    """centre crop of a given rectangle, scaling up first if the image is smaller (utils.py:40-53)"""
    h, w = img.shape[:2]
    if h < H_target or w < W_target:
        grow = max(H_target / h, W_target / w)
        img = _imresize(img, (int(round(h * grow)), int(round(w * grow))))
    return _centre_window(img, H_target, W_target)


def get_img_crop(src, resize=512, crop=256):
    return center_crop(resize_to(get_img(src), resize), crop)


def get_img_random_crop(src, resize=512, crop=256):
    """a random crop x crop window of the image resized to short side `resize` (utils.py:75-85)"""
    img = resize_to(get_img(src), resize=resize)
    top = random.randint(0, img.shape[0] - crop)
    left = random.randint(0, img.shape[1] - crop)
    return img[top:top + crop, left:left + crop]


def swap_filter_fit(H, W, patch_size, stride, n_pools=4):
    """Does a style-swap with this patch size / stride reproduce the relu5_1 map size of an H x W image?  If not,
    the content is cropped to the size that does (utils.py:115-138).  Returns (should_refit, H_out, W_out)."""
    def through(n):
        for _ in range(n_pools):
            n = (n + 1) // 2                                    # 'same' pooling
        covered = ((n - patch_size) // stride) * stride + patch_size   # conv then transposed conv
        return n, covered
    (hp, hc), (wp, wc) = through(H), through(W)
    return (hp != hc) or (wp != wc), hc * 2 ** n_pools, wc * 2 ** n_pools


def preserve_colors_np(style_rgb, content_rgb):
    """utils.py:87-90 on the GPU path (see ops.preserve_colors_np)"""
    from .ops import preserve_colors_np as impl
    return impl(style_rgb, content_rgb)

"""On-device reward preprocessing (host side: coefficient tables + kernel calls).

CLIP/PickScore: the reference round-trips every image through the CPU (uint8 -> PIL -> CLIPProcessor,
adv_grpo/rewards.py:567-571 + adv_grpo/pickscore_scorer.py:21-27).  Pillow's 8-bit resample is integer
arithmetic on coefficients that depend only on (in_size, out_size), so the tables are computed here once
(float64, same formulas and truncations as Pillow's Resample.c precompute_coeffs / normalize_coeffs_8bpc)
and the resize itself runs on the GPU, bit-exact.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
PRECISION_BITS = 32 - 8 - 2


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_tables(in_size, out_size):
    """(bounds int32 [out,2], coefs int32 [out,ksize], ksize) of Pillow's antialiased BICUBIC resample."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coefs = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        for x, w in enumerate(k):
            v = w * (1 << PRECISION_BITS)
            coefs[xx, x] = int(-0.5 + v) if w < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, coefs, ksize


_TABLES = {}


def _tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    if key not in _TABLES:
        b, c, k = pil_bicubic_tables(in_size, out_size)
        _TABLES[key] = (torch.from_numpy(b).to(device), torch.from_numpy(c).to(device), k)
    return _TABLES[key]


def _f3(v):
    return (ctypes.c_float * 3)(*v)


def clip_patches(images, size=224, x3=False):
    """images [B,3,H,W] in [0,1] (f32 or bf16, device) -> bf16 [B*(size/14)^2, 640] patch rows (CLIPProcessor)."""
    return pil_patches(images, size, CLIP_MEAN, CLIP_STD, trunc=False, x3=x3)


def pil_patches(images, size, mean, std, trunc=False, x3=False):
    """uint8 quantisation (round, or truncation as np.astype does) -> Pillow antialiased BICUBIC resize to
    size x size -> /255 -> (x-mean)/std -> 14x14 patch rows, all on the device and bit-exact with PIL.
    x3: the normalised pixels stay f32 and are returned as the split-bf16 left operand [B*P, 3*640] (vit_x3.py)."""
    lib = _lib.load()
    B, C, H, W = images.shape
    assert C == 3
    dev = images.device
    bh, ch, kh = _tables(W, size, dev)
    bv, cv, kv = _tables(H, size, dev)
    P = (size // 14) ** 2
    patches = torch.empty(B * P, 3 * 640 if x3 else 640, dtype=torch.bfloat16, device=dev)
    tmp = torch.empty(B * 3 * H * size, dtype=torch.uint8, device=dev)
    images = images.contiguous()
    _lib.check((lib.advgrpo_clip_preprocess_patches_x3 if x3 else lib.advgrpo_clip_preprocess_patches)(
        _lib.ptr(images), _lib.dtype_code(images.dtype), patches.data_ptr(), tmp.data_ptr(), B, H, W, size,
        size, bh.data_ptr(), ch.data_ptr(), kh, bv.data_ptr(), cv.data_ptr(), kv, _f3(mean), _f3(std), int(trunc),
        _lib.stream_ptr()))
    return patches


def dino_patches(images, size=518, x3=False):
    """x3: the fp32 pipeline (no bf16 rounding), patches returned as the split-bf16 left operand [B*P, 3*640] (vit_x3.py)."""
    lib = _lib.load()
    B, C, H, W = images.shape
    P = (size // 14) ** 2
    patches = torch.empty(B * P, 3 * 640 if x3 else 640, dtype=torch.bfloat16, device=images.device)
    images = images.contiguous()
    _lib.check((lib.advgrpo_dino_preprocess_patches_x3 if x3 else lib.advgrpo_dino_preprocess_patches)(_lib.ptr(images), _lib.dtype_code(images.dtype),
                                                   patches.data_ptr(), B, H, W, size, size, _f3(IMAGENET_MEAN),
                                                   _f3(IMAGENET_STD), _lib.stream_ptr()))
    return patches

"""CPU restatement of the image front-end of the data path (image side) -- TEST INFRASTRUCTURE ONLY, like
myriad_ref.py: imported by tests/ and tools/ only, never by the product path.

What the reference does to every image before it reaches the model:
  * training samples  datasets/datasets/anomaly_detection.py:118-122, 246, 330-333:
        torchvision Resize(224, BICUBIC) on the PIL image (shorter side -> 224) -> CenterCrop(224) -> np.asarray ->
        (NSA augmentation on the uint8 array) -> vis_processor = ToTensor + Normalize(CLIP mean / std)
        (processors/blip_processors.py:21-29, 120-147 with identity=True)
  * evaluation images processors/blip_processors.py:189-203: Resize((224, 224), BICUBIC) -> ToTensor -> Normalize.
torchvision is a pip dependency that is absent from the build container; its Resize / CenterCrop / ToTensor / Normalize are a
few lines each (size rule, crop offsets, /255, (x - mean) / std in float32) and are restated here from its published
functional code.  The resampling itself is Pillow's (torchvision calls PIL.Image.resize for PIL inputs), and Pillow IS
present: `resize_bicubic_u8` restates src/libImaging/Resample.c (8 bits per channel: double-precision coefficients of the
support-scaled Keys cubic a = -0.5, normalised, rounded to 22-bit fixed point; horizontal pass to a uint8 intermediate, then
vertical pass; round-half-up and clip per pass) and is pinned bit-for-bit against PIL.Image.resize itself in
tests/test_image_frontend.py (live, random images) and by tests/golden/image_frontend.npz (tools/make_golden_image.py).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # blip_processors.py:23-26
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: float, a: float = -0.5) -> float:
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full-image box: (kk [out, ksize] int32 fixed-point
    weights, bounds [out, 2] int32 = first input index and tap count)."""
    scale = in_size / out_size
    fscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    ss = 1.0 / fscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def _resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    kk, b = resample_coeffs(img.shape[axis], out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, xmax = int(b[xx, 0]), int(b[xx, 1])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[xx, :xmax].astype(np.int64), src[xmin:xmin + xmax], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """PIL.Image.resize((out_w, out_h), BICUBIC) of an HWC uint8 image: horizontal pass, then vertical; an axis whose size
    does not change is skipped (ImagingResample need_horizontal / need_vertical)."""
    t = _resample_axis(img, out_w, 1) if out_w != img.shape[1] else img
    return _resample_axis(t, out_h, 0) if out_h != img.shape[0] else t


def resized_size(h: int, w: int, size: int) -> Tuple[int, int]:
    """torchvision.transforms.functional._compute_resized_output_size for an int size: the shorter edge becomes `size`, the
    longer int(size * long / short).  Returns (new_h, new_w)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def center_crop_offsets(h: int, w: int, crop: int) -> Tuple[int, int]:
    """torchvision center_crop: top = int(round((h - crop) / 2.0)), left likewise (Python round: half to even)."""
    return int(round((h - crop) / 2.0)), int(round((w - crop) / 2.0))


def normalize_lut(mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """ToTensor + Normalize for every byte value: ((v / 255) - mean) / std evaluated in float32 like torch does -> [3, 256]."""
    v = np.arange(256, dtype=np.float32) / np.float32(255.0)
    return np.stack([(v - np.float32(m)) / np.float32(s) for m, s in zip(mean, std)]).astype(np.float32)


def train_image(img: np.ndarray, size: int = 224, mean=CLIP_MEAN, std=CLIP_STD) -> Tuple[np.ndarray, np.ndarray]:
    """Resize(size, BICUBIC) -> CenterCrop(size) -> ToTensor -> Normalize.  Returns (uint8 crop [size, size, 3] -- what the
    NSA augmentation sees -- and the normalised float32 [3, size, size])."""
    nh, nw = resized_size(img.shape[0], img.shape[1], size)
    r = resize_bicubic_u8(img, nw, nh)
    top, left = center_crop_offsets(nh, nw, size)
    c = r[top:top + size, left:left + size]
    lut = normalize_lut(mean, std)
    return c, np.stack([lut[ch][c[..., ch]] for ch in range(3)])


def eval_image(img: np.ndarray, size: int = 224, mean=CLIP_MEAN, std=CLIP_STD) -> Tuple[np.ndarray, np.ndarray]:
    """Resize((size, size), BICUBIC) -> ToTensor -> Normalize (blip2_image_eval)."""
    r = resize_bicubic_u8(img, size, size)
    lut = normalize_lut(mean, std)
    return r, np.stack([lut[ch][r[..., ch]] for ch in range(3)])

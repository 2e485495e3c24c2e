"""Image front-end of the data path on the GPU (image side).

Mirrors what the reference does on CPU workers for every image (datasets/datasets/anomaly_detection.py:118-122, 246, 330-333;
processors/blip_processors.py:21-29, 120-147, 189-203):

    train:  Resize(224, BICUBIC) -> CenterCrop(224) -> [uint8 crop, edited by the NSA augmentation] -> ToTensor -> Normalize
    eval:   Resize((224, 224), BICUBIC) -> ToTensor -> Normalize

with the decoded uint8 image resident in HBM and two HIP kernels per image (csrc/image.hip).  The resampling is Pillow's 8-bit
resampler to the bit (torchvision's Resize on a PIL image is PIL.Image.resize): the fixed-point weight tables are built here on
the host in double precision the way Resample.c builds them (once per distinct image size, cached on the device), the kernels
do the integer arithmetic, and ToTensor + Normalize is a 3 x 256 float32 table evaluated with torch's own float32 expressions.
At 130-140 images/s per GPU the reference's 8 CPU workers per GPU spend ~5 ms per 1000 x 1500 image in PIL alone; here it is
two launches.  The NSA / CutPaste blending itself (cv2.seamlessClone, datasets/self_sup_tasks.py) is not built: OpenCV is
absent from the build container, so its Poisson solver could not be pinned.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib, ops

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # blip_processors.py:23-26
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _cubic(x: float) -> float:
    a = -0.5
    x = -x if x < 0 else x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_tables(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (full-image box, bicubic): int32 weights [out, ksize] and
    (first, count) bounds [out, 2].  Python floats are IEEE doubles, like the C code's."""
    scale = in_size / out_size
    fscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    ss = 1.0 / fscale
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ws = [_cubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in ws:
            ww += w
        for x, w in enumerate(ws):
            if ww != 0.0:
                w = w / ww
            kk[xx, x] = int(-0.5 + w * one) if w < 0 else int(0.5 + w * one)
        bounds[xx, 0], bounds[xx, 1] = xmin, xmax
    return kk, bounds


def resized_size(h: int, w: int, size: int) -> Tuple[int, int]:
    """torchvision Resize(int): shorter edge -> size, longer -> int(size * long / short); (new_h, new_w)."""
    short, long = (w, h) if w <= h else (h, w)
    new_long = int(size * long / short)
    return (new_long, size) if w <= h else (size, new_long)


class ImageFrontEndHIP:
    """`mode="train"`: Resize(size) + CenterCrop(size); `mode="eval"`: Resize((size, size)).  Call with a list of uint8 HWC RGB
    images (numpy arrays, CPU or CUDA tensors; sizes may differ) -> float32 [B, 3, size, size] on the device.
    `return_u8=True` also returns the uint8 crops [B, size, size, 3] (what the reference hands to its augmentation);
    `normalize_u8` turns (edited) uint8 crops into model inputs."""

    def __init__(self, device, size: int = 224, mode: str = "train", mean: Sequence[float] = CLIP_MEAN,
                 std: Sequence[float] = CLIP_STD):
        if mode not in ("train", "eval"):
            raise ValueError("mode must be 'train' or 'eval'")
        _lib.load()                                   # no CPU path: fail loudly without the HIP library
        self.dev = torch.device(device)
        self.size, self.mode = int(size), mode
        v = np.arange(256, dtype=np.float32) / np.float32(255.0)
        lut = np.stack([(v - np.float32(m)) / np.float32(s) for m, s in zip(mean, std)]).astype(np.float32)
        self.lut = torch.from_numpy(lut).to(self.dev)
        self._tables: Dict[Tuple[int, int], tuple] = {}
        self._tmp: Optional[torch.Tensor] = None
        self._stage, self._stage_ev = None, None

    def _plan(self, H: int, W: int):
        key = (H, W)
        if key not in self._tables:
            S = self.size
            if self.mode == "train":
                rh, rw = resized_size(H, W, S)
                top, left = int(round((rh - S) / 2.0)), int(round((rw - S) / 2.0))   # torchvision center_crop
            else:
                rh, rw, top, left = S, S, 0, 0
            kh, bh = resample_tables(W, rw)
            kv, bv = resample_tables(H, rh)
            y0 = int(bv[top, 0])
            y1 = int(bv[top + S - 1, 0] + bv[top + S - 1, 1])
            dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
            self._tables[key] = (dev(kh), dev(bh), kh.shape[1], dev(kv), dev(bv), kv.shape[1], top, left, y0, y1 - y0)
        return self._tables[key]

    def _one(self, img, out: torch.Tensor, u8_out: Optional[torch.Tensor]):
        if isinstance(img, np.ndarray):
            img = torch.from_numpy(np.ascontiguousarray(img))
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise _lib.MyriadHipError(f"image front-end: need uint8 [H, W, 3], got {img.dtype} {tuple(img.shape)}")
        if not img.is_cuda:
            img = img.to(self.dev)                      # __call__ stages host images through one pinned buffer instead
        if img.stride(2) != 1 or img.stride(1) != 3:
            img = img.contiguous()
        H, W = int(img.shape[0]), int(img.shape[1])
        kh, bh, ksh, kv, bv, ksv, top, left, y0, rows = self._plan(H, W)
        S = self.size
        need = rows * S * 3
        if self._tmp is None or self._tmp.numel() < need:
            self._tmp = torch.empty((need,), dtype=torch.uint8, device=self.dev)
        rc = _lib.load().mh_image_resize_crop_norm(
            img.data_ptr(), H, W, img.stride(0), kh.data_ptr(), bh.data_ptr(), ksh, kv.data_ptr(), bv.data_ptr(), ksv, top, left,
            S, S, y0, rows, self._tmp.data_ptr(), self.lut.data_ptr(), out.data_ptr(),
            None if u8_out is None else u8_out.data_ptr(), ops._s())
        _lib.check(rc, f"mh_image_resize_crop_norm {H}x{W}")

    def _upload(self, images):
        """Host-resident images go through ONE pinned staging buffer and ONE asynchronous copy per call."""
        host = [(i, torch.from_numpy(np.ascontiguousarray(im)) if isinstance(im, np.ndarray) else im.contiguous())
                for i, im in enumerate(images) if isinstance(im, np.ndarray) or not im.is_cuda]
        if not host:
            return images
        total = sum(t.numel() for _, t in host)
        if getattr(self, "_stage", None) is None or self._stage.numel() < total:
            self._stage = torch.empty((total,), dtype=torch.uint8).pin_memory()
            self._stage_ev = None
        if self._stage_ev is not None:
            self._stage_ev.synchronize()               # the previous call's copy has left the buffer
        off, views = 0, []
        for _, t in host:
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
                raise _lib.MyriadHipError(f"image front-end: need uint8 [H, W, 3], got {t.dtype} {tuple(t.shape)}")
            self._stage[off:off + t.numel()].view(t.shape).copy_(t)
            views.append((off, t.shape))
            off += t.numel()
        dev = torch.empty((total,), dtype=torch.uint8, device=self.dev)
        dev.copy_(self._stage[:total], non_blocking=True)
        self._stage_ev = torch.cuda.Event()
        self._stage_ev.record()
        images = list(images)
        for (i, _), (o, shp) in zip(host, views):
            images[i] = dev[o:o + int(np.prod(shp))].view(shp)
        return images

    def __call__(self, images: List[Union[np.ndarray, torch.Tensor]], return_u8: bool = False):
        images = self._upload(images)
        B, S = len(images), self.size
        out = torch.empty((B, 3, S, S), dtype=torch.float32, device=self.dev)
        u8 = torch.empty((B, S, S, 3), dtype=torch.uint8, device=self.dev) if return_u8 else None
        for i, img in enumerate(images):
            self._one(img, out[i], None if u8 is None else u8[i])
        return (out, u8) if return_u8 else out

    def normalize_u8(self, u8: torch.Tensor) -> torch.Tensor:
        """ToTensor + Normalize of uint8 crops [B, S, S, 3] (device) -> float32 [B, 3, S, S]."""
        if u8.dtype != torch.uint8 or u8.dim() != 4 or u8.shape[3] != 3 or not u8.is_cuda or not u8.is_contiguous():
            raise _lib.MyriadHipError("normalize_u8: need a contiguous CUDA uint8 [B, S, S, 3] tensor")
        B, H, W, _ = u8.shape
        out = torch.empty((B, 3, H, W), dtype=torch.float32, device=u8.device)
        for i in range(B):
            _lib.check(_lib.load().mh_image_u8_normalize(u8[i].data_ptr(), H * W, self.lut.data_ptr(), out[i].data_ptr(),
                                                         ops._s()), "mh_image_u8_normalize")
        return out

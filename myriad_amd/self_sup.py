"""NSA / CutPaste self-supervised anomaly augmentation of the training data path: the reference's
`patch_ex` (minigpt4/datasets/self_sup_tasks.py:11-292, called from datasets/datasets/anomaly_detection.py:262-265).

Split the way the work splits on this machine:
  * `plan(...)`       HOST.  Everything random: the reference's np.random draws in the reference's order (gamma half-widths,
                      centres, shift search, coin flips for extra patches, CutPaste area / aspect, ellipse masks) and the
                      object-mask tests of `skip_background`.  Output: a list of patch operations (source box, destination
                      box, byte mask, factor).  A few hundred scalar operations per image.
  * `apply_numpy`     HOST, for DataLoader workers: the blends + the label on one uint8 crop.
  * `PatchExHIP`      DEVICE: the same blends + label for a whole batch of uint8 crops resident in HBM
                      (csrc/selfsup.hip), so the image never returns to the host between resize and normalise
                      (`image_frontend.ImageFrontEndHIP`: resize -> [this] -> normalise).
Blending modes: 'swap' and 'uniform' (arithmetic).  The shipped recipes use Poisson blending (`cv2.seamlessClone`,
self_sup_tasks.py:269-288) and `resize=True` (`cv2.resize`): both are OpenCV algorithms that can be neither run nor pinned
in the build container -- `mode='poisson'` delegates to cv2 when it is importable and raises otherwise; `resize=True`
raises.  Pinned against the reference's own function for everything else (tests/golden/self_sup.npz).
"""
from __future__ import annotations

import ctypes
import struct
from typing import List, Optional, Sequence, Tuple

import numpy as np

# the training split's arguments for MVTec-style data (anomaly_detection.py:133-143) minus the two OpenCV-only ones
NSA_ARGS = dict(num_patches=2, min_object_pct=0, min_overlap_pct=0.25, gamma_params=(2, 0.05, 0.03), resize=False, shift=True,
                same=False, label_mode="logistic-intensity")


class PatchOp:
    __slots__ = ("src_box", "dst_box", "mask", "factor", "mode")

    def __init__(self, src_box, dst_box, mask, factor, mode):
        self.src_box, self.dst_box, self.mask, self.factor, self.mode = src_box, dst_box, mask, factor, mode


def _median_blur(u8: np.ndarray, k: int) -> np.ndarray:
    import scipy.ndimage as ndi
    return ndi.median_filter(u8, size=k, mode="nearest")          # cv2.medianBlur: replicated borders


def _object_masks(ima_dest, ima_src, skip_background):
    if isinstance(skip_background, tuple):
        skip_background = [skip_background]
    so = np.ones_like(ima_src[..., 0:1])
    do = np.ones_like(ima_dest[..., 0:1])
    for background, threshold in skip_background:
        so &= np.uint8(np.abs(ima_src.mean(axis=-1, keepdims=True) - background) > threshold)
        do &= np.uint8(np.abs(ima_dest.mean(axis=-1, keepdims=True) - background) > threshold)
    so[..., 0] = _median_blur(so[..., 0], 7)
    do[..., 0] = _median_blur(do[..., 0], 7)
    return so, do


def plan(ima_dest: np.ndarray, ima_src: Optional[np.ndarray] = None, same: bool = False, num_patches: int = 1, mode="swap",
         width_bounds_pct=((0.05, 0.2), (0.05, 0.2)), min_object_pct=0.25, min_overlap_pct=0.25, shift: bool = True,
         label_mode: str = "binary", skip_background=None, resize: bool = False, gamma_params=None, num_ellipses=None,
         cutpaste_patch_generation: bool = False, rng=np.random) -> Tuple[List[PatchOp], float]:
    """The random part of `patch_ex` (self_sup_tasks.py:44-95, 116-256): which patches go where.  `rng` is np.random (the
    reference's generator) or a RandomState.  Returns (operations in application order, interpolation factor)."""
    if mode not in ("swap", "uniform"):
        raise NotImplementedError("only the arithmetic blends are planned here; Poisson: see patch_ex(mode='poisson')")
    if cutpaste_patch_generation:
        width_bounds_pct, resize, skip_background = None, False, None
        min_overlap_pct = min_object_pct = gamma_params = None
        num_patches = 1
    if resize:
        raise NotImplementedError("resize=True resamples the patch with cv2.resize: not available / not pinnable here")
    src_img = ima_dest if (same or ima_src is None) else ima_src
    so = do = None
    if skip_background is not None and not cutpaste_patch_generation:
        so, do = _object_masks(ima_dest, src_img, skip_background)
    factor = rng.uniform(0.05, 0.95) if label_mode == "continuous" else 1
    ops: List[PatchOp] = []
    Hh, Ww = ima_dest.shape[0], ima_dest.shape[1]
    for i in range(num_patches):
        if not (i == 0 or rng.randint(2) > 0):
            continue
        op = _plan_one(Hh, Ww, so, do, shift, width_bounds_pct, gamma_params, min_object_pct, min_overlap_pct, num_ellipses,
                       cutpaste_patch_generation, rng)
        if op is not None:
            op.factor, op.mode = factor, mode
            ops.append(op)
    return ops, factor


def _plan_one(Hh, Ww, so, do, shift, width_bounds_pct, gamma_params, min_object_pct, min_overlap_pct, num_ellipses, cutpaste,
              rng) -> Optional[PatchOp]:
    dims = np.array([Hh, Ww, 3])
    if cutpaste:
        skip_bg = False
        if Hh != Ww:
            raise ValueError("CutPaste patch generation only works for square images")
        area_ratio = rng.uniform(0.02, 0.15) / 4.0
        aspect = rng.uniform(0.3, 1) if rng.randint(2) > 0 else rng.uniform(1, 3.3)
        w1 = int(np.rint(np.clip(np.sqrt(area_ratio * aspect * Hh ** 2), 0, Hh)))
        w2 = int(np.rint(np.clip(area_ratio * Hh ** 2 / w1, 0, Ww)))
        c1, c2 = rng.randint(w1, Hh - w1), rng.randint(w2, Ww - w2)
        a1, a2 = int(np.clip(c1 - w1, 0, Hh)), int(np.clip(c2 - w2, 0, Ww))
        b1, b2 = int(np.clip(c1 + w1, 0, Hh)), int(np.clip(c2 + w2, 0, Ww))
        pm = np.ones((b1 - a1, b2 - a2, 1), dtype=np.uint8)
    else:
        skip_bg = so is not None and do is not None
        lo1, hi1 = (width_bounds_pct[0][0] * dims[0]).round().astype(int), (width_bounds_pct[0][1] * dims[0]).round().astype(int)
        lo2, hi2 = (width_bounds_pct[1][0] * dims[1]).round().astype(int), (width_bounds_pct[1][1] * dims[1]).round().astype(int)
        if gamma_params is not None:
            shape, scale, lower = gamma_params
            w1 = int(np.clip((lower + rng.gamma(shape, scale)) * dims[0], lo1, hi1))
            w2 = int(np.clip((lower + rng.gamma(shape, scale)) * dims[1], lo2, hi2))
        else:
            w1, w2 = rng.randint(lo1, hi1), rng.randint(lo2, hi2)
        for attempt in range(200):
            c1, c2 = rng.randint(lo1, dims[0] - lo1), rng.randint(lo2, dims[1] - lo2)
            a1, a2 = int(np.clip(c1 - w1, 0, Hh)), int(np.clip(c2 - w2, 0, Ww))
            b1, b2 = int(np.clip(c1 + w1, 0, Hh)), int(np.clip(c2 + w2, 0, Ww))
            if num_ellipses is not None:
                pm2 = np.zeros((b1 - a1, b2 - a2), dtype=np.uint8)
                x = np.arange(pm2.shape[0]).reshape(-1, 1)
                y = np.arange(pm2.shape[1]).reshape(1, -1)
                for _ in range(num_ellipses):
                    theta = rng.uniform(0, np.pi)
                    x0, y0 = rng.randint(0, pm2.shape[0]), rng.randint(0, pm2.shape[1])
                    ea = rng.randint(lo1, max(lo1 + 1, w1 // 2))
                    eb = rng.randint(lo2, max(lo2 + 1, w2 // 2))
                    pm2 |= (((x - x0) * np.cos(theta) + (y - y0) * np.sin(theta)) / ea) ** 2 + \
                           (((x - x0) * np.sin(theta) + (y - y0) * np.cos(theta)) / eb) ** 2 <= 1
                pm = pm2[..., None]
            else:
                pm = np.ones((b1 - a1, b2 - a2, 1), dtype=np.uint8)
            if not skip_bg:
                break
            area = np.sum(pm) if num_ellipses is not None else pm.shape[0] * pm.shape[1]
            if np.sum(pm & so[a1:b1, a2:b2]) / area > min_object_pct:
                break
        else:
            return None                                   # 200 attempts without a patch on the object (:206-209)
    sa1, sa2 = a1, a2
    height, width = b1 - a1, b2 - a2
    so_p = so[a1:b1, a2:b2, 0].copy()[..., None] if skip_bg else None
    if shift:
        for attempt in range(200):
            c1 = rng.randint(height // 2 + 1, Hh - height // 2 - 1)
            c2 = rng.randint(width // 2 + 1, Ww - width // 2 - 1)
            a1, b1 = c1 - height // 2, c1 + (height + 1) // 2
            a2, b2 = c2 - width // 2, c2 + (width + 1) // 2
            if not skip_bg:
                break
            both = do[a1:b1, a2:b2] & so_p & pm
            if np.sum(so_p) / (pm.shape[0] * pm.shape[1]) > min_object_pct and np.sum(both) / np.sum(so_p) > min_overlap_pct:
                break
        else:
            return None
    if skip_bg:
        pm = pm & (so_p | do[a1:b1, a2:b2])
    return PatchOp((sa1, sa2), (int(a1), int(a2), int(height), int(width)), np.ascontiguousarray(pm[..., 0]), 1.0, "swap")


def apply_numpy(ima_dest: np.ndarray, ima_src: np.ndarray, ops: Sequence[PatchOp], factor: float, label_mode: str = "binary",
                tol: int = 1, intensity_logistic_params=(1 / 6, 20)):
    """Blends + label on the host (self_sup_tasks.py:254-268, 97-113)."""
    import scipy.ndimage as ndi
    out = ima_dest.copy()
    mask = np.zeros_like(ima_dest[..., 0:1])
    boxes = []
    lo1, hi1, lo2, hi2 = mask.shape[0] - 1, 0, mask.shape[1] - 1, 0
    for op in ops:
        (sy, sx), (y0, x0, h, w) = op.src_box, op.dst_box
        pm = op.mask[..., None]
        src = ima_src[sy:sy + h, sx:sx + w]
        if op.mode == "swap":
            out[y0:y0 + h, x0:x0 + w] = np.where(pm.astype(bool), src, out[y0:y0 + h, x0:x0 + w])
        else:
            f = 1.0 * out
            before = f[y0:y0 + h, x0:x0 + w]
            before -= op.factor * pm * before
            before += op.factor * pm * src
            out = np.uint8(np.floor(f))
        mask[y0:y0 + h, x0:x0 + w] = pm
        lo1, hi1, lo2, hi2 = min(lo1, y0), max(hi1, y0 + h), min(lo2, x0), max(hi2, x0 + w)
        boxes.append([lo2, lo1, hi2, hi1])
    diff = np.abs(mask.astype(np.int32) * ima_dest - mask.astype(np.int32) * out).sum(-1, keepdims=True)
    lm = np.uint8(diff > 3 * tol)
    lm[..., 0] = _median_blur(lm[..., 0], 5)
    if label_mode == "binary":
        label = lm
    elif label_mode == "continuous":
        label = lm * factor
    elif label_mode in ("intensity", "logistic-intensity"):
        k, x0 = intensity_logistic_params
        y, x = np.mgrid[-5:6, -5:6]
        label = (lm.astype(np.int32) * np.abs(ima_dest.astype(np.int32) - out)).sum(-1, keepdims=True) / 3.0
        label[..., 0] = ndi.median_filter(label[..., 0], footprint=(x * x + y * y <= 25), mode="nearest")
        if label_mode == "logistic-intensity":
            label = lm / (1 + np.exp(-k * (label - x0)))
    else:
        raise ValueError("label_mode not supported" + str(label_mode))
    return out, label, boxes


def patch_ex(ima_dest: np.ndarray, ima_src: Optional[np.ndarray] = None, mode="swap", tol: int = 1,
             intensity_logistic_params=(1 / 6, 20), rng=np.random, **kw):
    """`patch_ex` (self_sup_tasks.py:11-113) on the host: plan + apply.  mode='poisson' hands the whole call to OpenCV's
    seamlessClone through the reference-equivalent path when cv2 is importable."""
    if mode == "poisson":
        try:
            import cv2  # noqa: F401
        except ImportError as e:
            raise NotImplementedError("Poisson blending is cv2.seamlessClone (self_sup_tasks.py:269-288): OpenCV is not installed") from e
        raise NotImplementedError("with OpenCV present call the reference's patch_ex directly; this build pins only the arithmetic blends")
    ops, factor = plan(ima_dest, ima_src, mode=mode, rng=rng, **kw)
    src = ima_dest if (kw.get("same") or ima_src is None) else ima_src
    return apply_numpy(ima_dest, src, ops, factor, kw.get("label_mode", "binary"), tol, intensity_logistic_params)


# ------------------------------------------------------------------------------------------------ device path
_LABEL_MODES = {"binary": 0, "continuous": 1, "intensity": 2, "logistic-intensity": 3}


class PatchExHIP:
    """Blends + labels for a batch of uint8 crops on the GPU (csrc/selfsup.hip).  `plans[b]` = (ops, factor) from `plan`."""

    def __init__(self, device="cuda"):
        import torch
        from . import _lib
        self.torch, self.lib, self.dev = torch, _lib.load(), torch.device(device)

    def __call__(self, dest_u8, src_u8, plans, label_mode: str = "binary", tol: int = 1, intensity_logistic_params=(1 / 6, 20)):
        """dest_u8, src_u8: [B,H,W,3] uint8 device tensors.  Returns (patchex [B,H,W,3] u8, label [B,H,W] f32, union mask)."""
        torch = self.torch
        from . import _lib, ops as O
        B, H, W, _ = dest_u8.shape
        out = dest_u8.clone()
        union = torch.zeros((B, H, W), dtype=torch.uint8, device=self.dev)
        recs, masks, hs, ws, off = [], [], [], [], 0
        for b, (ops_b, _f) in enumerate(plans):
            for op in ops_b:
                (sy, sx), (y0, x0, h, w) = op.src_box, op.dst_box
                recs.append(struct.pack("<8iqd", b, y0, x0, h, w, sy, sx, 0 if op.mode == "swap" else 1, off, float(op.factor)))
                masks.append(np.ascontiguousarray(op.mask, dtype=np.uint8).reshape(-1))
                hs.append(h); ws.append(w)
                off += h * w
        if recs:
            ops_dev = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(self.dev)
            pool = torch.from_numpy(np.concatenate(masks)).to(self.dev)
            hs_c, ws_c = (ctypes.c_int * len(hs))(*hs), (ctypes.c_int * len(ws))(*ws)
            _lib.check(self.lib.mh_patch_blend_u8(out.data_ptr(), src_u8.data_ptr(), pool.data_ptr(), ops_dev.data_ptr(),
                                                  ctypes.cast(hs_c, ctypes.c_void_p), ctypes.cast(ws_c, ctypes.c_void_p), len(recs), B, H,
                                                  W, union.data_ptr(), O._s()), "mh_patch_blend_u8")
        sums = torch.empty((B, H, W), dtype=torch.int32, device=self.dev)
        lm = torch.empty((B, H, W), dtype=torch.uint8, device=self.dev)
        label = torch.empty((B, H, W), dtype=torch.float32, device=self.dev)
        factor = torch.tensor([float(f) for _, f in plans], dtype=torch.float64).to(self.dev)
        k, x0 = intensity_logistic_params
        _lib.check(self.lib.mh_patch_label(dest_u8.data_ptr(), out.data_ptr(), union.data_ptr(), sums.data_ptr(), lm.data_ptr(),
                                           label.data_ptr(), factor.data_ptr(), B, H, W, _LABEL_MODES[label_mode], int(tol), float(k),
                                           float(x0), O._s()), "mh_patch_label")
        return out, label, union

"""ORACLE (test infrastructure, never on the product path): CPU restatement of the reference's NSA / CutPaste
self-supervised augmentation `patch_ex` / `_patch_ex` (minigpt4/datasets/self_sup_tasks.py:11-292) for the branches that
need neither cv2.resize nor cv2.seamlessClone.

Pinned against tests/golden/self_sup.npz, produced by the reference's own function (tools/make_golden_selfsup.py, scipy
shims for cv2.medianBlur / skimage.filters.median).  np.random is consumed in exactly the reference's order, so with the same
seed the same patches are drawn.  Parity UNPINNED (neither cv2 nor the reference's dependency can run here): `resize=True`
(cv2.resize, bilinear) and the Poisson modes (cv2.seamlessClone) -- both raise here.
"""
from __future__ import annotations

import numpy as np
import scipy.ndimage as ndi


def _median_blur(u8: np.ndarray, k: int) -> np.ndarray:
    """cv2.medianBlur(uint8, k): k x k median with replicated borders (self_sup_tasks.py:67-68, 98)."""
    return ndi.median_filter(u8, size=k, mode="nearest")


def _disk(r: int) -> np.ndarray:
    y, x = np.mgrid[-r:r + 1, -r:r + 1]
    return x * x + y * y <= r * r


def patch_ex(ima_dest, ima_src=None, same=False, num_patches=1, mode="swap", width_bounds_pct=((0.05, 0.2), (0.05, 0.2)),
             min_object_pct=0.25, min_overlap_pct=0.25, shift=True, label_mode="binary", skip_background=None, tol=1,
             resize=False, gamma_params=None, intensity_logistic_params=(1 / 6, 20), resize_bounds=(0.7, 1.3),
             num_ellipses=None, cutpaste_patch_generation=False):
    """self_sup_tasks.py:11-113.  Returns (patchex uint8, label, label_boxes)."""
    if mode not in ("swap", "uniform"):
        raise NotImplementedError("Poisson blending (cv2.seamlessClone) cannot be restated or pinned without OpenCV")
    if cutpaste_patch_generation:                                   # :47-54
        width_bounds_pct, resize, skip_background = None, False, None
        min_overlap_pct = min_object_pct = gamma_params = None
        num_patches = 1
    if resize:
        raise NotImplementedError("resize=True needs cv2.resize; unpinned")
    ima_src = ima_dest.copy() if same or ima_src is None else ima_src   # :56
    src_obj = dest_obj = None
    if skip_background is not None and not cutpaste_patch_generation:   # :58-68
        if isinstance(skip_background, tuple):
            skip_background = [skip_background]
        src_obj = np.ones_like(ima_src[..., 0:1])
        dest_obj = np.ones_like(ima_dest[..., 0:1])
        for background, threshold in skip_background:
            src_obj &= np.uint8(np.abs(ima_src.mean(axis=-1, keepdims=True) - background) > threshold)
            dest_obj &= np.uint8(np.abs(ima_dest.mean(axis=-1, keepdims=True) - background) > threshold)
        src_obj[..., 0] = _median_blur(src_obj[..., 0], 7)
        dest_obj[..., 0] = _median_blur(dest_obj[..., 0], 7)
    boxes = []
    mask = np.zeros_like(ima_dest[..., 0:1])
    patchex = ima_dest.copy()
    c1lo, c1hi, c2lo, c2hi = mask.shape[0] - 1, 0, mask.shape[1] - 1, 0      # :77
    factor = np.random.uniform(0.05, 0.95) if label_mode == "continuous" else 1   # :78-81
    for i in range(num_patches):                                    # :82-95
        if i == 0 or np.random.randint(2) > 0:
            patchex, ((a1, b1), (a2, b2)), pm = _patch_ex(patchex, ima_src, dest_obj, src_obj, mode, shift, width_bounds_pct,
                                                          gamma_params, min_object_pct, min_overlap_pct, factor, num_ellipses,
                                                          cutpaste_patch_generation)
            if pm is not None:
                mask[a1:b1, a2:b2] = pm
                c1lo, c1hi, c2lo, c2hi = min(c1lo, a1), max(c1hi, b1), min(c2lo, a2), max(c2hi, b2)
                boxes.append([c2lo, c1lo, c2hi, c1hi])
    # label (:97-111)
    label_mask = np.uint8(np.mean(np.abs(1.0 * mask * ima_dest - 1.0 * mask * patchex), axis=-1, keepdims=True) > tol)
    label_mask[..., 0] = _median_blur(label_mask[..., 0], 5)
    if label_mode == "continuous":
        label = label_mask * factor
    elif label_mode in ("logistic-intensity", "intensity"):
        k, x0 = intensity_logistic_params
        label = np.mean(np.abs(label_mask * ima_dest * 1.0 - label_mask * patchex * 1.0), axis=-1, keepdims=True)
        label[..., 0] = ndi.median_filter(label[..., 0], footprint=_disk(5), mode="nearest")   # skimage.filters.median
        if label_mode == "logistic-intensity":
            label = label_mask / (1 + np.exp(-k * (label - x0)))
    elif label_mode == "binary":
        label = label_mask
    else:
        raise ValueError("label_mode not supported" + str(label_mode))
    return patchex, label, boxes


def _patch_ex(ima_dest, ima_src, dest_obj, src_obj, mode, shift, width_bounds_pct, gamma_params, min_object_pct,
              min_overlap_pct, factor, num_ellipses, cutpaste):
    """self_sup_tasks.py:116-292 (resize=False, arithmetic blends)."""
    dims = np.array(ima_dest.shape)
    if cutpaste:                                                    # :118-146
        skip_bg = False
        if dims[0] != dims[1]:
            raise ValueError("CutPaste patch generation only works for square images")
        area_ratio = np.random.uniform(0.02, 0.15) / 4.0
        aspect = np.random.uniform(0.3, 1) if np.random.randint(2) > 0 else np.random.uniform(1, 3.3)
        w1 = int(np.rint(np.clip(np.sqrt(area_ratio * aspect * dims[0] ** 2), 0, dims[0])))
        w2 = int(np.rint(np.clip(area_ratio * dims[0] ** 2 / w1, 0, dims[1])))
        ce1 = np.random.randint(w1, dims[0] - w1)
        ce2 = np.random.randint(w2, dims[1] - w2)
        a1, a2 = np.clip(ce1 - w1, 0, dims[0]), np.clip(ce2 - w2, 0, dims[1])
        b1, b2 = np.clip(ce1 + w1, 0, dims[0]), np.clip(ce2 + w2, 0, dims[1])
        pm = np.ones((b1 - a1, b2 - a2, 1), dtype=np.uint8)
    else:                                                           # :147-209
        skip_bg = (src_obj is not None) and (dest_obj is not None)
        lo1 = (width_bounds_pct[0][0] * dims[0]).round().astype(int)
        hi1 = (width_bounds_pct[0][1] * dims[0]).round().astype(int)
        lo2 = (width_bounds_pct[1][0] * dims[1]).round().astype(int)
        hi2 = (width_bounds_pct[1][1] * dims[1]).round().astype(int)
        if gamma_params is not None:
            shape, scale, lower = gamma_params
            w1 = int(np.clip((lower + np.random.gamma(shape, scale)) * dims[0], lo1, hi1))
            w2 = int(np.clip((lower + np.random.gamma(shape, scale)) * dims[1], lo2, hi2))
        else:
            w1 = np.random.randint(lo1, hi1)
            w2 = np.random.randint(lo2, hi2)
        attempts = 0
        while True:
            ce1 = np.random.randint(lo1, dims[0] - lo1)
            ce2 = np.random.randint(lo2, dims[1] - lo2)
            a1, a2 = np.clip(ce1 - w1, 0, dims[0]), np.clip(ce2 - w2, 0, dims[1])
            b1, b2 = np.clip(ce1 + w1, 0, dims[0]), np.clip(ce2 + w2, 0, dims[1])
            if num_ellipses is not None:
                emin1, emin2 = lo1, lo2
                emax1, emax2 = max(lo1 + 1, w1 // 2), max(lo2 + 1, w2 // 2)
                pm = np.zeros((b1 - a1, b2 - a2), dtype=np.uint8)
                x = np.arange(pm.shape[0]).reshape(-1, 1)
                y = np.arange(pm.shape[1]).reshape(1, -1)
                for _ in range(num_ellipses):
                    theta = np.random.uniform(0, np.pi)
                    x0 = np.random.randint(0, pm.shape[0])
                    y0 = np.random.randint(0, pm.shape[1])
                    ea = np.random.randint(emin1, emax1)
                    eb = np.random.randint(emin2, emax2)
                    ell = (((x - x0) * np.cos(theta) + (y - y0) * np.sin(theta)) / ea) ** 2 + \
                          (((x - x0) * np.sin(theta) + (y - y0) * np.cos(theta)) / eb) ** 2 <= 1
                    pm |= ell
                pm = pm[..., None]
            else:
                pm = np.ones((b1 - a1, b2 - a2, 1), dtype=np.uint8)
            if skip_bg:
                bg_area = np.sum(pm & src_obj[a1:b1, a2:b2])
                area = np.sum(pm) if num_ellipses is not None else pm.shape[0] * pm.shape[1]
                found = bg_area / area > min_object_pct
            else:
                found = True
            attempts += 1
            if found:
                break
            if attempts == 200:
                return ima_dest.copy(), ((0, 0), (0, 0)), None
    src = ima_src[a1:b1, a2:b2]                                     # :211-212
    height, width, _ = src.shape
    so = None
    if skip_bg:                                                     # :225-227 (cv2.resize to the same size: identity)
        so = src_obj[a1:b1, a2:b2, 0].copy()[..., None]
    if shift:                                                       # :230-252
        attempts = 0
        while True:
            ce1 = np.random.randint(height // 2 + 1, ima_dest.shape[0] - height // 2 - 1)
            ce2 = np.random.randint(width // 2 + 1, ima_dest.shape[1] - width // 2 - 1)
            a1, b1 = ce1 - height // 2, ce1 + (height + 1) // 2
            a2, b2 = ce2 - width // 2, ce2 + (width + 1) // 2
            if skip_bg:
                both = dest_obj[a1:b1, a2:b2] & so & pm
                found = (np.sum(so) / (pm.shape[0] * pm.shape[1]) > min_object_pct and np.sum(both) / np.sum(so) > min_overlap_pct)
            else:
                found = True
            attempts += 1
            if found:
                break
            if attempts == 200:
                return ima_dest.copy(), ((0, 0), (0, 0)), None
    if skip_bg:                                                     # :255-256
        pm = pm & (so | dest_obj[a1:b1, a2:b2])
    if mode == "swap":                                              # :258-262 (uint8 arithmetic, wraps like numpy's)
        out = ima_dest.copy()
        before = out[a1:b1, a2:b2]
        out[a1:b1, a2:b2] -= pm * before
        out[a1:b1, a2:b2] += pm * src
    else:                                                           # 'uniform' :263-268
        out = 1.0 * ima_dest
        before = out[a1:b1, a2:b2]
        out[a1:b1, a2:b2] -= factor * pm * before
        out[a1:b1, a2:b2] += factor * pm * src
        out = np.uint8(np.floor(out))
    return out, ((a1, b1), (a2, b2)), pm

"""NSA / CutPaste self-supervised anomaly augmentation of the training data path: the reference's
`patch_ex` (minigpt4/datasets/self_sup_tasks.py:11-292, called from datasets/datasets/anomaly_detection.py:254-264).

Split the way the work splits on this machine:
  * `plan(...)`       HOST.  Everything random: the reference's np.random draws in the reference's order (gamma half-widths,
                      centres, the resize scale, shift search, coin flips for extra patches, CutPaste area / aspect, ellipse
                      masks) and the object-mask tests of `skip_background`.  Output: a list of patch operations (source box,
                      resampled size, destination box, byte masks, factor).  A few hundred scalar operations per image.
  * `apply_numpy`     HOST, for DataLoader workers: resample + blend + label on one uint8 crop.
  * `PatchExHIP`      DEVICE: the same for a whole batch of uint8 crops resident in HBM (csrc/selfsup.hip), so the image never
                      returns to the host between resize and normalise (`image_frontend.ImageFrontEndHIP`).
Blending modes: 'swap', 'uniform' (arithmetic) and 'normal_clone' (= cv2.NORMAL_CLONE = 1, Poisson blending, the mode the
shipped recipe trains with, anomaly_detection.py:118-141); `resize=True` resamples the patch as `cv2.resize` does.
The two OpenCV algorithms are built from their published form -- 8-bit INTER_LINEAR with 11-bit fixed-point weights
(modules/imgproc/src/resize.cpp) and gradient-domain cloning solved by a discrete sine transform (modules/photo/src/
seamless_cloning_impl.cpp) -- and are PARITY UNPINNED: OpenCV is not installed, the reference pins no version, no OpenCV
output exists here.  Pinned: the reference's own code around them (tests/golden/self_sup.npz: the reference ran with the
oracle's stand-ins), and properties of the result (tests/test_self_sup.py).  Stated deviation: the Poisson solve runs in
float64 and truncates `floor(v + 1e-6)` where OpenCV truncates a float32 `v`.
"""
from __future__ import annotations

import ctypes
import struct
from typing import List, Optional, Sequence, Tuple

import numpy as np

NORMAL_CLONE = 1                      # cv2.NORMAL_CLONE
MIXED_CLONE = 2                       # cv2.MIXED_CLONE
TRUNC_EPS = 1e-6
# the training split's base arguments (anomaly_detection.py:118-141): MVTec-style and VisA
NSA_ARGS = dict(num_patches=2, min_object_pct=0, min_overlap_pct=0.25, gamma_params=(2, 0.05, 0.03), resize=True, shift=True,
                same=False, mode="normal_clone", label_mode="logistic-intensity")
NSA_ARGS_VISA = dict(NSA_ARGS, width_bounds_pct=((0.03, 0.4), (0.03, 0.4)), intensity_logistic_params=(1 / 12, 24),
                     skip_background=None, resize_bounds=(.5, 2))
# per-class arguments of the MVTec branch (anomaly_detection.py:50-65,254-259)
MVTEC_WIDTH_BOUNDS_PCT = {'bottle': ((0.03, 0.4), (0.03, 0.4)), 'cable': ((0.05, 0.4), (0.05, 0.4)), 'capsule': ((0.03, 0.15), (0.03, 0.4)),
                          'hazelnut': ((0.03, 0.35), (0.03, 0.35)), 'metal_nut': ((0.03, 0.4), (0.03, 0.4)), 'pill': ((0.03, 0.2), (0.03, 0.4)),
                          'screw': ((0.03, 0.12), (0.03, 0.12)), 'toothbrush': ((0.03, 0.4), (0.03, 0.2)), 'transistor': ((0.03, 0.4), (0.03, 0.4)),
                          'zipper': ((0.03, 0.4), (0.03, 0.2)), 'carpet': ((0.03, 0.4), (0.03, 0.4)), 'grid': ((0.03, 0.4), (0.03, 0.4)),
                          'leather': ((0.03, 0.4), (0.03, 0.4)), 'tile': ((0.03, 0.4), (0.03, 0.4)), 'wood': ((0.03, 0.4), (0.03, 0.4))}
MVTEC_INTENSITY_LOGISTIC_PARAMS = {'bottle': (1 / 12, 24), 'cable': (1 / 12, 24), 'capsule': (1 / 2, 4), 'hazelnut': (1 / 12, 24),
                                   'metal_nut': (1 / 3, 7), 'pill': (1 / 3, 7), 'screw': (1, 3), 'toothbrush': (1 / 6, 15),
                                   'transistor': (1 / 6, 15), 'zipper': (1 / 6, 15), 'carpet': (1 / 3, 7), 'grid': (1 / 3, 7),
                                   'leather': (1 / 3, 7), 'tile': (1 / 3, 7), 'wood': (1 / 6, 15)}
MVTEC_BACKGROUND = {'bottle': (200, 60), 'screw': (200, 60), 'capsule': (200, 60), 'zipper': (200, 60), 'hazelnut': (20, 20),
                    'pill': (20, 20), 'toothbrush': (20, 20), 'metal_nut': (20, 20)}


def self_sup_args(dataset: str, class_name: str, visa_base: Optional[bool] = None) -> dict:
    """The keyword arguments `AnomalyDetectionDataset.__getitem__` passes to patch_ex (anomaly_detection.py:254-264), which are
    two independent choices in the reference:
      * the BASE set, fixed per dataset object by `'VISA' in ann_paths[0]` (anomaly_detection.py:118-141): the VisA set, else the
        MVTec-style set WITHOUT width bounds / logistic parameters / resize bounds / background (patch_ex's own defaults apply);
      * the per-item extras, by get_class_name's split (:224-230): 'mvtec' adds that class's width bounds / logistic parameters /
        background (None for a class that is in no table, as dict.get gives); anything else adds nothing.
    `visa_base=None` keeps the two tied (VisA base for a non-'mvtec' item), which is what every shipped annotation file gives.
    A VisA base with MVTec extras passes width_bounds_pct twice in the reference (a TypeError there): the same error here."""
    if visa_base is None:
        visa_base = dataset != "mvtec"
    base = dict(NSA_ARGS_VISA) if visa_base else dict(NSA_ARGS)
    if dataset == "mvtec":
        if visa_base:
            raise TypeError("patch_ex() got multiple values for keyword argument 'width_bounds_pct' (VisA base set + MVTec per-class "
                            "arguments: anomaly_detection.py:118-141 with :254-259)")
        base.update(width_bounds_pct=MVTEC_WIDTH_BOUNDS_PCT.get(class_name),
                    intensity_logistic_params=MVTEC_INTENSITY_LOGISTIC_PARAMS.get(class_name),
                    skip_background=MVTEC_BACKGROUND.get(class_name))
    return base


class PatchOp:
    """One patch: source box (sy, sx) of size `src_size` (= the patch size unless resampled), destination box
    (y0, x0, h, w), the byte mask written into the label's union mask, the interpolation factor, the blend mode; for
    'normal_clone' also `pms` (the mask handed to seamlessClone: scaled, background added, border cleared) and the ROI
    geometry derived from it."""
    __slots__ = ("src_box", "src_size", "dst_box", "mask", "factor", "mode", "pms", "roi")

    def __init__(self, src_box, dst_box, mask, factor, mode, src_size=None, pms=None, roi=None):
        self.src_box, self.dst_box, self.mask, self.factor, self.mode = src_box, dst_box, mask, factor, mode
        self.src_size = src_size if src_size is not None else (dst_box[2], dst_box[3])
        self.pms, self.roi = pms, roi

    @property
    def resized(self) -> bool:
        return tuple(self.src_size) != (self.dst_box[2], self.dst_box[3])


# ---- cv2.resize(uint8, INTER_LINEAR) from its published algorithm (PARITY UNPINNED, module docstring) ---------------------
def linear_resize_tables(ssize: int, dsize: int):
    """Left source index and the two 11-bit weights per destination index (resize.cpp: fx = (dx + 0.5) * scale - 0.5 in
    float, floor, clamp at both ends, weights cvRound((1 - fx) * 2048), cvRound(fx * 2048))."""
    scale = 1.0 / (float(dsize) / float(ssize))
    f = ((np.arange(dsize, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    sx = np.floor(f).astype(np.int64)
    f = (f - sx.astype(np.float32)).astype(np.float32)
    lo, hi = sx < 0, sx >= ssize - 1
    f = np.where(lo | hi, np.float32(0), f).astype(np.float32)
    sx = np.where(lo, 0, np.where(hi, ssize - 1, sx))
    w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int32)
    w1 = np.rint(f * np.float32(2048)).astype(np.int32)
    return sx.astype(np.int32), np.stack([w0, w1], 1)


def resize_linear_u8(img: np.ndarray, dsize) -> np.ndarray:
    dw, dh = int(dsize[0]), int(dsize[1])
    a = img if img.ndim == 3 else img[..., None]
    sh, sw, _ = a.shape
    if (dw, dh) == (sw, sh):
        return img.copy()
    q = a.astype(np.int32)
    if sw == 2 * dw and sh == 2 * dh:                     # exact 2 x 2 decimation: OpenCV switches to the area mean
        out = (q[0::2, 0::2] + q[0::2, 1::2] + q[1::2, 0::2] + q[1::2, 1::2] + 2) >> 2
    else:
        xi, xw = linear_resize_tables(sw, dw)
        yi, yw = linear_resize_tables(sh, dh)
        x1, y1 = np.minimum(xi + 1, sw - 1), np.minimum(yi + 1, sh - 1)
        rows = q[:, xi] * xw[None, :, 0:1] + q[:, x1] * xw[None, :, 1:2]
        out = (((yw[:, 0, None, None] * (rows[yi] >> 4)) >> 16) + ((yw[:, 1, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    out = out.astype(np.uint8)
    return out if img.ndim == 3 else out[..., 0]


def clone_roi(pms: np.ndarray, center, dest_hw):
    """seamless_cloning.cpp: roi_s = bounding box of the non-zero mask (its 1-pixel border cleared), roi_d = the same size
    around `center` = (x, y).  Returns (y0s, x0s, dy0, dx0, h, w) or None when the clone must be skipped (empty mask; the ROI
    leaves the image -- OpenCV raises and the reference's except-branch returns the input unchanged)."""
    m = pms if pms.ndim == 2 else pms[..., 0]
    ys, xs = np.nonzero(m)
    if ys.size == 0:
        return None
    y0, x0 = int(ys.min()), int(xs.min())
    h, w = int(ys.max()) + 1 - y0, int(xs.max()) + 1 - x0
    dx0, dy0 = int(center[0]) - w // 2, int(center[1]) - h // 2
    if dx0 < 0 or dy0 < 0 or dx0 + w > dest_hw[1] or dy0 + h > dest_hw[0]:
        return None
    return y0, x0, dy0, dx0, h, w


def eroded_mask(pms: np.ndarray, roi) -> np.ndarray:
    """binaryMask of Cloning::computeDerivatives: the ROI of the mask after erode(3 x 3) x 3 (outside the ROI never erodes)."""
    import scipy.ndimage as ndi
    y0, x0, _, _, h, w = roi
    m = (pms if pms.ndim == 2 else pms[..., 0])[y0:y0 + h, x0:x0 + w].astype(np.int32)
    for _ in range(3):
        m = ndi.minimum_filter(m, size=3, mode="constant", cval=255)
    return m.astype(np.uint8)


def dst_tables(n: int):
    """Sine matrix S[k, j] = sin(pi (k+1)(j+1) / (n+1)) and the eigenvalue terms 2 cos(pi (k+1) / (n+1)) of the 1-D Laplacian."""
    k = np.arange(1, n + 1, dtype=np.float64)
    return np.sin(np.pi * np.outer(k, k) / (n + 1)), 2.0 * np.cos(np.pi * k / (n + 1))


def poisson_clone_numpy(out: np.ndarray, patch: np.ndarray, pms: np.ndarray, roi, mixed: bool = False) -> None:
    """Cloning::normalClone (NORMAL_CLONE) -- or, with `mixed`, Cloning::mixedClone (MIXED_CLONE: per element the patch's
    gradient pair is kept where |Px - Py| > |Dx - Dy|, else the destination's pair, seamless_cloning_impl.cpp) -- on `out` in
    place, float64, the transform through scipy's DST-I."""
    import scipy.fft as sfft
    y0, x0, dy0, dx0, h, w = roi
    if h < 3 or w < 3:
        return
    m = (pms if pms.ndim == 2 else pms[..., 0])
    D = out[dy0:dy0 + h, dx0:dx0 + w].astype(np.float64)
    P = np.where(m[y0:y0 + h, x0:x0 + w, None] != 0, patch[y0:y0 + h, x0:x0 + w], 0).astype(np.float64)
    me = eroded_mask(pms, roi).astype(np.float64)[..., None]
    mf, mi = me / 255.0, (255.0 - me) / 255.0
    dgx, pgx = D[:, 1:] - D[:, :-1], P[:, 1:] - P[:, :-1]                                    # forward differences, columns 0..w-2
    dgy, pgy = D[1:] - D[:-1], P[1:] - P[:-1]                                                # rows 0..h-2
    if mixed:                                                                                # defined where both components are: [h-1, w-1]
        keep = np.abs(pgx[:-1] - pgy[:, :-1]) > np.abs(dgx[:-1] - dgy[:, :-1])
        pgx, pgy = pgx.copy(), pgy.copy()
        pgx[:-1] = np.where(keep, pgx[:-1], dgx[:-1])
        pgy[:, :-1] = np.where(keep, pgy[:, :-1], dgy[:, :-1])
    gx = dgx * mi[:, :-1] + pgx * mf[:, :-1]
    gy = dgy * mi[:-1] + pgy * mf[:-1]
    lap = (gx[1:-1, 1:] - gx[1:-1, :-1]) + (gy[1:, 1:-1] - gy[:-1, 1:-1])                    # interior (h-2) x (w-2)
    ring = D.copy()
    ring[1:-1, 1:-1] = 0
    rhs = lap - (ring[1:-1, :-2] + ring[1:-1, 2:] + ring[:-2, 1:-1] + ring[2:, 1:-1])
    nh, nw = h - 2, w - 2
    den = (2.0 * np.cos(np.pi * np.arange(1, nh + 1) / (nh + 1)))[:, None, None] + \
          (2.0 * np.cos(np.pi * np.arange(1, nw + 1) / (nw + 1)))[None, :, None] - 4.0
    u = sfft.idstn(sfft.dstn(rhs, type=1, axes=(0, 1)) / den, type=1, axes=(0, 1))
    out[dy0 + 1:dy0 + h - 1, dx0 + 1:dx0 + w - 1] = np.floor(np.clip(u, 0.0, 255.0) + TRUNC_EPS).astype(np.uint8)




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


def _mode(mode, rng=np.random):
    """self_sup_tasks.py:22,47-48: 'mix' flips a coin between the two Poisson modes -- one np.random.randint(2) draw, the FIRST
    draw of patch_ex."""
    if mode == "mix":
        mode = (NORMAL_CLONE, MIXED_CLONE)[int(rng.randint(2))]
    if mode in (NORMAL_CLONE, "normal_clone", "poisson"):
        return "normal_clone"
    if mode in (MIXED_CLONE, "mixed_clone"):
        return "mixed_clone"
    if mode in ("swap", "uniform"):
        return mode
    raise ValueError("mode not supported" + str(mode))          # self_sup_tasks.py:290-291


def plan(ima_dest: np.ndarray, ima_src: Optional[np.ndarray] = None, same: bool = False, num_patches: int = 1, mode="swap",
         width_bounds_pct=((0.05, 0.2), (0.05, 0.2)), min_object_pct=0.25, min_overlap_pct=0.25, shift: bool = True,
         label_mode: str = "binary", skip_background=None, resize: bool = False, gamma_params=None, num_ellipses=None,
         cutpaste_patch_generation: bool = False, resize_bounds=(0.7, 1.3), rng=np.random) -> Tuple[List[PatchOp], float]:
    """The random part of `patch_ex` (self_sup_tasks.py:44-95, 116-256, 271-279): which patches go where.  `rng` is np.random
    (the reference's generator) or a RandomState.  Returns (operations in application order, interpolation factor)."""
    mode = _mode(mode, rng)
    if cutpaste_patch_generation:
        width_bounds_pct, resize, skip_background = None, False, None
        min_overlap_pct = min_object_pct = gamma_params = None
        num_patches = 1
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
                       cutpaste_patch_generation, rng, resize, resize_bounds, mode, factor)
        if op is not None:
            op.factor, op.mode = factor, mode
            ops.append(op)
    return ops, factor


def _plan_one(Hh, Ww, so, do, shift, width_bounds_pct, gamma_params, min_object_pct, min_overlap_pct, num_ellipses, cutpaste,
              rng, resize=False, resize_bounds=(0.7, 1.3), mode="swap", factor=1) -> Optional[PatchOp]:
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
    h0, w0 = height, width
    if resize:                                            # :213-224 (the patch itself is resampled by apply_numpy / the device)
        lb, ub = resize_bounds
        scale = np.clip(rng.normal(1, 0.5), lb, ub)
        new_h = np.clip(scale * height, lo1, hi1)
        new_w = np.clip(int(new_h / height * width), lo2, hi2)
        new_h = np.clip(int(new_w / width * height), lo1, hi1)
        height, width = int(new_h), int(new_w)
        pm = resize_linear_u8(pm[..., 0], (width, height))[..., None]
    so_p = resize_linear_u8(so[a1:b1, a2:b2, 0], (width, height))[..., None] if skip_bg else None      # :225-227
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
    pms = roi = None
    if mode in ("normal_clone", "mixed_clone"):          # :267-279
        int_factor = np.uint8(np.ceil(factor * 255))
        pms = int_factor * (pm | ((1 - so_p) & (1 - do[a1:b1, a2:b2]))) if skip_bg else int_factor * pm
        pms[0], pms[-1], pms[:, 0], pms[:, -1] = 0, 0, 0, 0
        center = (b2 - (b2 - a2) // 2, a1 + (b1 - a1) // 2)
        if np.sum(pms > 0) < 50:
            return None
        pms = np.ascontiguousarray(pms[..., 0])
        roi = clone_roi(pms, center, (Hh, Ww))
        if roi is None:                                   # cv2.error in the reference: the patch is dropped
            return None
    return PatchOp((sa1, sa2), (int(a1), int(a2), int(height), int(width)), np.ascontiguousarray(pm[..., 0]), 1.0, "swap",
                   src_size=(int(h0), int(w0)), pms=pms, roi=roi)


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
        src = ima_src[sy:sy + op.src_size[0], sx:sx + op.src_size[1]]
        if op.resized:
            src = resize_linear_u8(src, (w, h))
        if op.mode in ("normal_clone", "mixed_clone"):
            poisson_clone_numpy(out, src, op.pms, op.roi, mixed=op.mode == "mixed_clone")
        elif op.mode == "swap":
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
    """`patch_ex` (self_sup_tasks.py:11-113) on the host: plan + apply."""
    ops, factor = plan(ima_dest, ima_src, mode=mode, rng=rng, **kw)
    src = ima_dest if (kw.get("same") or ima_src is None) else ima_src
    return apply_numpy(ima_dest, src, ops, factor, kw.get("label_mode", "binary"), tol, intensity_logistic_params)


# ------------------------------------------------------------------------------------------------ device path
_LABEL_MODES = {"binary": 0, "continuous": 1, "intensity": 2, "logistic-intensity": 3}


class PatchExHIP:
    """Resample + blend + label for a batch of uint8 crops on the GPU (csrc/selfsup.hip).  `plans[b]` = (ops, factor) from
    `plan`.  Operations run in plan order (a later patch sees an earlier one's pixels); per operation: the patch is resampled
    into a pool when its size changes (mh_patch_resize_u8), 'normal_clone' solves the Poisson problem in place
    (mh_patch_poisson_u8), and one blend launch writes the pixels ('swap' / 'uniform') and the label's union mask."""

    def __init__(self, device="cuda"):
        import torch
        from . import _lib
        self.torch, self.lib, self.dev = torch, _lib.load(), torch.device(device)
        self._dst = {}                                   # n -> (sine matrix, 2 cos terms) on the device

    def _dst_tables(self, n: int):
        if n not in self._dst:
            S, c = dst_tables(n)
            self._dst[n] = (self.torch.from_numpy(S).to(self.dev), self.torch.from_numpy(c).to(self.dev))
        return self._dst[n]

    def __call__(self, dest_u8, src_u8, plans, label_mode: str = "binary", tol: int = 1, intensity_logistic_params=(1 / 6, 20)):
        """dest_u8, src_u8: [B,H,W,3] uint8 device tensors.  Returns (patchex [B,H,W,3] u8, label [B,H,W] f32, union mask)."""
        torch = self.torch
        from . import _lib, ops as O
        B, H, W, _ = dest_u8.shape
        out = dest_u8.clone()
        union = torch.zeros((B, H, W), dtype=torch.uint8, device=self.dev)
        recs, masks, flat, off, poff, tabs, toff = [], [], [], 0, 0, [], 0
        for b, (ops_b, _f) in enumerate(plans):
            for op in ops_b:
                (sy, sx), (y0, x0, h, w) = op.src_box, op.dst_box
                clone = op.mode in ("normal_clone", "mixed_clone")
                pooled = op.resized or clone
                mode = {"swap": 0, "uniform": 1, "normal_clone": 2, "mixed_clone": 2}[op.mode] | (4 if pooled and not clone else 0)
                recs.append(struct.pack("<8iqdq", b, y0, x0, h, w, sy, sx, mode, off, float(op.factor), poff))
                masks.append(np.ascontiguousarray(op.mask, dtype=np.uint8).reshape(-1))
                t = None
                if pooled:
                    xi, xw = linear_resize_tables(op.src_size[1], w)
                    yi, yw = linear_resize_tables(op.src_size[0], h)
                    t = (toff, toff + w, toff + 3 * w, toff + 3 * w + h)
                    tabs += [xi, xw.reshape(-1), yi, yw.reshape(-1)]
                    toff += 3 * w + 3 * h
                flat.append((b, op, off, poff if pooled else None, t))
                off += h * w
                poff += h * w * 3 if pooled else 0
        if recs:
            ops_dev = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(self.dev)
            pool = torch.from_numpy(np.concatenate(masks)).to(self.dev)
            patches = torch.empty(max(poff, 1), dtype=torch.uint8, device=self.dev)
            tab_dev = torch.from_numpy(np.concatenate(tabs).astype(np.int32)).to(self.dev) if tabs else None
            one = (ctypes.c_int * 1)
            keep = []                                     # device buffers stay alive until the launches are queued
            for o, (b, op, moff, p_off, t) in enumerate(flat):
                (sy, sx), (y0, x0, h, w) = op.src_box, op.dst_box
                if p_off is not None:
                    tp = tab_dev.data_ptr()
                    _lib.check(self.lib.mh_patch_resize_u8(src_u8.data_ptr(), b, H, W, sy, sx, op.src_size[0], op.src_size[1],
                                                           tp + 4 * t[0], tp + 4 * t[1], tp + 4 * t[2], tp + 4 * t[3],
                                                           patches.data_ptr() + p_off, h, w, O._s()), "mh_patch_resize_u8")
                if op.mode in ("normal_clone", "mixed_clone"):
                    y0s, x0s, dy0, dx0, rh, rw = op.roi
                    if rh >= 3 and rw >= 3:
                        pms = torch.from_numpy(np.ascontiguousarray(op.pms, dtype=np.uint8)).to(self.dev)
                        er = torch.from_numpy(np.ascontiguousarray(eroded_mask(op.pms, op.roi))).to(self.dev)
                        Sh, cy = self._dst_tables(rh - 2)
                        Sw, cx = self._dst_tables(rw - 2)
                        ws = torch.empty(self.lib.mh_patch_poisson_ws_doubles(rh, rw), dtype=torch.float64, device=self.dev)
                        keep += [pms, er, ws]
                        _lib.check(self.lib.mh_patch_poisson_u8(out.data_ptr(), b, H, W, patches.data_ptr() + p_off, h, w,
                                                                pms.data_ptr(), er.data_ptr(), y0s, x0s, dy0, dx0, rh, rw, Sh.data_ptr(),
                                                                cy.data_ptr(), Sw.data_ptr(), cx.data_ptr(), ws.data_ptr(),
                                                                1 if op.mode == "mixed_clone" else 0, O._s()),
                                   "mh_patch_poisson_u8")
                _lib.check(self.lib.mh_patch_blend_u8(out.data_ptr(), src_u8.data_ptr(), pool.data_ptr(), ops_dev.data_ptr() + 56 * o,
                                                      ctypes.cast(one(h), ctypes.c_void_p), ctypes.cast(one(w), ctypes.c_void_p), 1, B, H,
                                                      W, union.data_ptr(), patches.data_ptr(), O._s()), "mh_patch_blend_u8")
            torch.cuda.current_stream().synchronize() if keep else None
        sums = torch.empty((B, H, W), dtype=torch.int32, device=self.dev)
        lm = torch.empty((B, H, W), dtype=torch.uint8, device=self.dev)
        label = torch.empty((B, H, W), dtype=torch.float32, device=self.dev)
        factor = torch.tensor([float(f) for _, f in plans], dtype=torch.float64).to(self.dev)
        k, x0 = intensity_logistic_params
        _lib.check(self.lib.mh_patch_label(dest_u8.data_ptr(), out.data_ptr(), union.data_ptr(), sums.data_ptr(), lm.data_ptr(),
                                           label.data_ptr(), factor.data_ptr(), B, H, W, _LABEL_MODES[label_mode], int(tol), float(k),
                                           float(x0), O._s()), "mh_patch_label")
        return out, label, union

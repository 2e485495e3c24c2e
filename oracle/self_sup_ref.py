"""ORACLE (test infrastructure, never on the product path): CPU restatement of the reference's NSA / CutPaste
self-supervised augmentation `patch_ex` / `_patch_ex` (minigpt4/datasets/self_sup_tasks.py:11-292), every branch.

Pinned against tests/golden/self_sup.npz, produced by the reference's own function (tools/make_golden_selfsup.py).
np.random is consumed in exactly the reference's order, so with the same seed the same patches are drawn.
What the golden pins, and what it cannot:
  * arithmetic blends ('swap', 'uniform'), patch geometry, object masks, labels: the reference ran with scipy stand-ins
    for cv2.medianBlur / skimage.filters.median only -- PINNED;
  * `resize=True` and `mode=cv2.NORMAL_CLONE` (the shipped recipe, anomaly_detection.py:118-141): the reference's own code
    around the two OpenCV calls (size draws and clipping :213-224, mask scaling / border / centre / the 50-pixel rule
    :271-279) is PINNED -- the reference ran, with `cv2.resize` and `cv2.seamlessClone` supplied by `resize_linear_u8` and
    `seamless_clone` below.  Those two functions restate OpenCV's published algorithms (modules/imgproc/src/resize.cpp:
    8-bit INTER_LINEAR with 11-bit fixed-point coefficients; modules/photo/src/seamless_cloning*.cpp: gradient-domain
    NORMAL_CLONE solved with a discrete sine transform) and are **PARITY UNPINNED**: OpenCV is not installed, is pinned to no
    version anywhere in the reference tree, and no vector of its output exists here.
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
    """self_sup_tasks.py:11-113.  Returns (patchex uint8, label, label_boxes).  mode: 'swap', 'uniform', or
    NORMAL_CLONE (= cv2.NORMAL_CLONE = 1, also spelled 'normal_clone'), MIXED_CLONE (= 2, 'mixed_clone'), or 'mix' (a coin flip
    between the two Poisson modes, :47-48: the first random draw of the call)."""
    if mode == "mix":
        mode = (NORMAL_CLONE, MIXED_CLONE)[np.random.randint(2)]
    if mode == "normal_clone":
        mode = NORMAL_CLONE
    if mode == "mixed_clone":
        mode = MIXED_CLONE
    if mode not in ("swap", "uniform", NORMAL_CLONE, MIXED_CLONE):
        raise ValueError("mode not supported" + str(mode))          # :290-291
    if cutpaste_patch_generation:                                   # :47-54
        width_bounds_pct, resize, skip_background = None, False, None
        min_overlap_pct = min_object_pct = gamma_params = None
        num_patches = 1
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
                                                          cutpaste_patch_generation, resize, resize_bounds)
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
              min_overlap_pct, factor, num_ellipses, cutpaste, resize=False, resize_bounds=(0.7, 1.3)):
    """self_sup_tasks.py:116-292."""
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
    if resize:                                                      # :213-224
        lb, ub = resize_bounds
        scale = np.clip(np.random.normal(1, 0.5), lb, ub)
        new_h = np.clip(scale * height, lo1, hi1)
        new_w = np.clip(int(new_h / height * width), lo2, hi2)
        new_h = np.clip(int(new_w / width * height), lo1, hi1)      # in case there was clipping
        src = resize_linear_u8(src, (new_w, new_h))
        height, width, _ = src.shape
        pm = resize_linear_u8(pm[..., 0], (width, height))[..., None]
    so = None
    if skip_bg:                                                     # :225-227
        so = resize_linear_u8(src_obj[a1:b1, a2:b2, 0], (width, height))[..., None]
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
    if mode in (NORMAL_CLONE, MIXED_CLONE):                         # :267-288 Poisson interpolation
        int_factor = np.uint8(np.ceil(factor * 255))
        if skip_bg:                                                 # background added to the mask to avoid artefacts
            pms = int_factor * (pm | ((1 - so) & (1 - dest_obj[a1:b1, a2:b2])))
        else:
            pms = int_factor * pm
        pms[0], pms[-1], pms[:, 0], pms[:, -1] = 0, 0, 0, 0         # zero border
        center = (b2 - (b2 - a2) // 2, a1 + (b1 - a1) // 2)         # (x, y)
        if np.sum(pms > 0) < 50:
            return ima_dest.copy(), ((0, 0), (0, 0)), None
        try:
            out = seamless_clone(src, ima_dest, pms, center, mode)
        except ValueError:                                          # cv2.error in the reference
            return ima_dest.copy(), ((0, 0), (0, 0)), None
    elif mode == "swap":                                            # :258-262 (uint8 arithmetic, wraps like numpy's)
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


# ---------------------------------------------------------------------------------------------------------------------
# OpenCV stand-ins (PARITY UNPINNED, see the module docstring)
# ---------------------------------------------------------------------------------------------------------------------
NORMAL_CLONE, MIXED_CLONE = 1, 2


def _linear_coeffs(ssize: int, dsize: int):
    """resize.cpp (INTER_LINEAR, 8U): per destination index the left source index and the two 11-bit weights."""
    scale = 1.0 / (float(dsize) / float(ssize))
    idx = np.zeros(dsize, np.int64)
    w = np.zeros((dsize, 2), np.int64)
    for d in range(dsize):
        f = np.float32((d + 0.5) * scale - 0.5)
        sx = int(np.floor(f))
        f = np.float32(f - np.float32(sx))
        if sx < 0:
            f, sx = np.float32(0), 0
        if sx >= ssize - 1:
            f, sx = np.float32(0), ssize - 1
        idx[d] = sx
        w[d, 0] = int(np.rint(np.float32(np.float32(1) - f) * np.float32(2048)))      # cvRound: half to even
        w[d, 1] = int(np.rint(f * np.float32(2048)))
    return idx, w


def resize_linear_u8(img: np.ndarray, dsize) -> np.ndarray:
    """cv2.resize(img, (width, height)) for uint8, default INTER_LINEAR (self_sup_tasks.py:219-227).  Horizontal pass in
    int32 (pixel x 11-bit weight), vertical pass  ((b0 (S0 >> 4)) >> 16) + ((b1 (S1 >> 4)) >> 16) + 2) >> 2;  an exact 2 x 2
    down-scale is the 2x2 mean, rounded (resize.cpp hands that case to INTER_AREA)."""
    dw, dh = int(dsize[0]), int(dsize[1])
    a = img if img.ndim == 3 else img[..., None]
    sh, sw, c = a.shape
    if (dw, dh) == (sw, sh):
        return img.copy()
    if sw == 2 * dw and sh == 2 * dh:
        q = a.astype(np.int64)
        out = ((q[0::2, 0::2] + q[0::2, 1::2] + q[1::2, 0::2] + q[1::2, 1::2] + 2) >> 2).astype(np.uint8)
        return out if img.ndim == 3 else out[..., 0]
    xi, xw = _linear_coeffs(sw, dw)
    yi, yw = _linear_coeffs(sh, dh)
    q = a.astype(np.int64)
    x1 = np.minimum(xi + 1, sw - 1)
    rows = q[:, xi, :] * xw[None, :, 0, None] + q[:, x1, :] * xw[None, :, 1, None]          # [sh, dw, c]
    y1 = np.minimum(yi + 1, sh - 1)
    s0, s1 = rows[yi], rows[y1]
    out = ((((yw[:, 0, None, None] * (s0 >> 4)) >> 16) + ((yw[:, 1, None, None] * (s1 >> 4)) >> 16) + 2) >> 2)
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out if img.ndim == 3 else out[..., 0]


def _sine_matrix(n: int) -> np.ndarray:
    k = np.arange(1, n + 1, dtype=np.float64)
    return np.sin(np.pi * np.outer(k, k) / (n + 1))


def poisson_dirichlet(rhs: np.ndarray, h: int, w: int) -> np.ndarray:
    """Solve  u(y,x-1) + u(y,x+1) + u(y-1,x) + u(y+1,x) - 4 u(y,x) = rhs  on the (h-2) x (w-2) interior with zero boundary,
    by the discrete sine transform (Cloning::solve: divide by 2 cos(pi (i+1)/(w-1)) + 2 cos(pi (j+1)/(h-1)) - 4)."""
    nh, nw = h - 2, w - 2
    Sh, Sw = _sine_matrix(nh), _sine_matrix(nw)
    t = Sh @ rhs @ Sw
    den = (2.0 * np.cos(np.pi * np.arange(1, nh + 1) / (nh + 1)))[:, None] + (2.0 * np.cos(np.pi * np.arange(1, nw + 1) / (nw + 1)))[None, :] - 4.0
    return (Sh @ (t / den) @ Sw) * (2.0 / (nh + 1)) * (2.0 / (nw + 1))


TRUNC_EPS = 1e-6    # OpenCV truncates (static_cast<uchar>); a solution that is an exact integer must not fall to the level below
                    # through 1e-12 of round-off -- every implementation here adds this before truncating (stated in DESIGN.md)


def seamless_clone(src: np.ndarray, dst: np.ndarray, mask: np.ndarray, center, flags: int = NORMAL_CLONE) -> np.ndarray:
    """cv2.seamlessClone(src, dst, mask, (x, y), cv2.NORMAL_CLONE) for uint8 3-channel images, float64 arithmetic.
    seamless_cloning.cpp: the mask's 1-pixel border is cleared; roi_s = bounding box of the non-zero mask; roi_d = the same
    size around `center`; seamless_cloning_impl.cpp (normalClone): forward-difference gradients of the destination ROI and
    of the masked source ROI, mixed with the mask eroded 3 x (3 x 3, border never erodes), divergence, minus the Laplacian
    of the ROI's boundary ring, Poisson solve with that ring as Dirichlet data, clamp to [0, 255] and truncate."""
    if flags not in (NORMAL_CLONE, MIXED_CLONE):
        raise NotImplementedError("only NORMAL_CLONE (the shipped recipe) and MIXED_CLONE are restated")
    m = np.array(mask, copy=True)
    if m.ndim == 3:
        m = m[..., 0]
    m[0, :] = m[-1, :] = 0
    m[:, 0] = m[:, -1] = 0
    ys, xs = np.nonzero(m)
    out = dst.copy()
    if ys.size == 0:
        return out
    y0, y1, x0, x1 = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
    h, w = y1 - y0, x1 - x0
    dx0, dy0 = int(center[0]) - w // 2, int(center[1]) - h // 2
    if dx0 < 0 or dy0 < 0 or dx0 + w > dst.shape[1] or dy0 + h > dst.shape[0]:
        raise ValueError("seamless_clone: the destination ROI leaves the image (OpenCV raises cv2.error)")
    if h < 3 or w < 3:
        return out
    mroi = m[y0:y1, x0:x1] != 0
    D = dst[dy0:dy0 + h, dx0:dx0 + w].astype(np.float64)
    P = np.where(mroi[..., None], src[y0:y1, x0:x1], 0).astype(np.float64)
    me = m[y0:y1, x0:x1].astype(np.int64)
    for _ in range(3):                                                # erode(3 x 3) x 3; outside the ROI counts as the maximum
        me = ndi.minimum_filter(me, size=3, mode="constant", cval=255)
    mf = (me / 255.0)[..., None]                                      # binaryMaskFloat
    mi = ((255 - me) / 255.0)[..., None]                              # bitwise_not, then / 255

    def grad_x(a):                                                    # filter2D [0, -1, 1], BORDER_REFLECT_101
        g = np.empty_like(a)
        g[:, :-1] = a[:, 1:] - a[:, :-1]
        g[:, -1] = a[:, -2] - a[:, -1]
        return g

    def grad_y(a):
        g = np.empty_like(a)
        g[:-1] = a[1:] - a[:-1]
        g[-1] = a[-2] - a[-1]
        return g

    pgx, pgy, dgx, dgy = grad_x(P), grad_y(P), grad_x(D), grad_y(D)
    if flags == MIXED_CLONE:
        # Cloning::mixedClone (seamless_cloning_impl.cpp): per element the patch's gradient pair is kept where
        # |Px - Py| > |Dx - Dy| (OpenCV compares the DIFFERENCE of the two components, not their magnitude), else the
        # destination's pair takes its place -- both then weighted by the eroded mask like the patch gradients of normalClone
        keep = np.abs(pgx - pgy) > np.abs(dgx - dgy)
        pgx, pgy = np.where(keep, pgx, dgx), np.where(keep, pgy, dgy)
    gx = dgx * mi + pgx * mf
    gy = dgy * mi + pgy * mf
    lap = np.zeros_like(D)                                            # filter2D [-1, 1, 0]: g(x) - g(x-1), interior only is used
    lap[:, 1:] += gx[:, 1:] - gx[:, :-1]
    lap[1:, :] += gy[1:, :] - gy[:-1, :]
    ring = D.copy()
    ring[1:-1, 1:-1] = 0
    ring_lap = np.zeros_like(D)                                       # Laplacian(bound): 4-neighbour stencil
    ring_lap[1:-1, 1:-1] = (ring[1:-1, :-2] + ring[1:-1, 2:] + ring[:-2, 1:-1] + ring[2:, 1:-1] - 4 * ring[1:-1, 1:-1])
    rhs = (lap - ring_lap)[1:-1, 1:-1]
    res = D.copy()
    for ch in range(D.shape[2]):
        u = poisson_dirichlet(rhs[..., ch], h, w)
        res[1:-1, 1:-1, ch] = np.floor(np.clip(u, 0.0, 255.0) + TRUNC_EPS)
    out[dy0:dy0 + h, dx0:dx0 + w] = np.clip(res, 0, 255).astype(np.uint8)
    return out

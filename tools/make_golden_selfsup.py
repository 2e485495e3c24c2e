#!/usr/bin/env python3
"""Golden vectors for the NSA / CutPaste self-supervised augmentation: runs the REFERENCE's own
`patch_ex` (minigpt4/datasets/self_sup_tasks.py:11-292) in this container and commits inputs + outputs.

The reference module imports cv2 and skimage, neither of which is installed here.  What it needs from them is supplied by
import shims:
    cv2.NORMAL_CLONE / MIXED_CLONE     the two integer constants (1, 2)
    cv2.medianBlur(u8, k)              scipy.ndimage.median_filter(size=k, mode='nearest')   [OpenCV: BORDER_REPLICATE]
    skimage.morphology.disk(r)         boolean x^2 + y^2 <= r^2 footprint
    skimage.filters.median(img, fp)    scipy.ndimage.median_filter(footprint=fp, mode='nearest')  [what skimage itself calls]
    cv2.resize(u8, (w, h))             oracle/self_sup_ref.resize_linear_u8   -- a restatement of OpenCV's 8-bit INTER_LINEAR
    cv2.seamlessClone(..NORMAL_CLONE / MIXED_CLONE)  oracle/self_sup_ref.seamless_clone  -- a restatement of OpenCV's Poisson cloning
The first four are independent library implementations: cases that only use them pin the whole function.  The last two are
this repository's own restatements of published OpenCV algorithms (PARITY UNPINNED, see oracle/self_sup_ref.py): the cases
with resize=True / mode=cv2.NORMAL_CLONE -- the shipped recipe, datasets/datasets/anomaly_detection.py:118-141,254-264 --
pin the REFERENCE'S OWN CODE around those two calls (the np.random sequence, the size arithmetic and clipping :213-224, the
object-mask logic, mask scaling / zeroed border / centre / the 50-pixel rule :271-279, the label), not OpenCV's pixels.

python tools/make_golden_selfsup.py [--ref /root/reference]   ->  tests/golden/self_sup.npz
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import scipy.ndimage as ndi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_reference(ref):
    cv2 = types.ModuleType("cv2")
    cv2.NORMAL_CLONE, cv2.MIXED_CLONE = 1, 2
    cv2.medianBlur = lambda img, k: ndi.median_filter(img, size=k, mode="nearest")

    class error(Exception):
        pass

    cv2.error = error

    from oracle import self_sup_ref as O          # the restated OpenCV algorithms (see the module docstring)

    def _seamless(src, dst, mask, center, flags):
        try:
            return O.seamless_clone(src, dst, mask, center, flags)
        except ValueError as e:
            raise error(str(e))

    cv2.resize, cv2.seamlessClone = (lambda img, dsize: O.resize_linear_u8(img, dsize)), _seamless
    sk = types.ModuleType("skimage")
    skm = types.ModuleType("skimage.morphology")
    skf = types.ModuleType("skimage.filters")

    def disk(r):
        y, x = np.mgrid[-r:r + 1, -r:r + 1]
        return (x * x + y * y <= r * r).astype(np.uint8)

    skm.disk = disk
    skf.median = lambda img, fp: ndi.median_filter(img, footprint=fp.astype(bool), mode="nearest")
    sys.modules.update({"cv2": cv2, "skimage": sk, "skimage.morphology": skm, "skimage.filters": skf})
    spec = importlib.util.spec_from_file_location("ref_self_sup", os.path.join(ref, "minigpt4/datasets/self_sup_tasks.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_image(seed, size=224):
    """Low-entropy structured uint8 image (compresses well): background level + a textured object + a few blocks."""
    r = np.random.RandomState(seed)
    y, x = np.mgrid[0:size, 0:size]
    img = np.full((size, size, 3), 20 + 5 * (seed % 3), np.float64)
    cy, cx, rad = r.randint(80, 144), r.randint(80, 144), r.randint(60, 90)
    obj = ((y - cy) ** 2 + (x - cx) ** 2) <= rad * rad
    tex = 120 + 60 * np.sin(x / (5.0 + seed % 4)) * np.cos(y / (7.0 + seed % 3))
    for ch in range(3):
        img[..., ch] = np.where(obj, tex + 15 * ch, img[..., ch])
    for _ in range(4):
        y0, x0, h, w = r.randint(0, size - 40), r.randint(0, size - 40), r.randint(8, 40), r.randint(8, 40)
        img[y0:y0 + h, x0:x0 + w] = r.randint(0, 255, 3)
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = [
    # name, np seed, kwargs of patch_ex (beyond ima_dest / ima_src)
    ("swap_nsa", 1, dict(mode="swap", num_patches=2, min_object_pct=0, min_overlap_pct=0.25, gamma_params=(2, 0.05, 0.03), resize=False,
                         shift=True, same=False, label_mode="logistic-intensity", width_bounds_pct=((0.05, 0.2), (0.05, 0.2)))),
    ("swap_nsa_b", 2, dict(mode="swap", num_patches=2, min_object_pct=0, min_overlap_pct=0.25, gamma_params=(2, 0.05, 0.03), resize=False,
                           shift=True, same=False, label_mode="logistic-intensity", width_bounds_pct=((0.03, 0.4), (0.03, 0.4)),
                           intensity_logistic_params=(1 / 12, 24))),
    ("uniform_cont", 3, dict(mode="uniform", num_patches=3, resize=False, shift=True, label_mode="continuous", min_object_pct=0,
                             min_overlap_pct=0.25)),
    ("swap_binary_noshift", 4, dict(mode="swap", num_patches=1, resize=False, shift=False, label_mode="binary")),
    ("swap_skipbg", 5, dict(mode="swap", num_patches=2, resize=False, shift=True, label_mode="intensity", skip_background=(20, 10),
                            min_object_pct=0.25, min_overlap_pct=0.25, gamma_params=(2, 0.05, 0.03))),
    ("cutpaste", 6, dict(mode="swap", cutpaste_patch_generation=True, label_mode="binary", shift=True)),
    ("uniform_ellipses", 7, dict(mode="uniform", num_patches=2, resize=False, shift=True, label_mode="logistic-intensity", num_ellipses=3,
                                 gamma_params=(2, 0.05, 0.03))),
    # ---- the shipped recipe: resize=True + Poisson blending, with the per-class argument sets of anomaly_detection.py:50-65,118-141
    ("swap_resize", 9, dict(mode="swap", num_patches=2, min_object_pct=0, min_overlap_pct=0.25, gamma_params=(2, 0.05, 0.03), resize=True,
                            shift=True, same=False, label_mode="logistic-intensity", width_bounds_pct=((0.03, 0.4), (0.03, 0.4)))),
    ("poisson_noresize", 10, dict(mode=1, num_patches=2, min_object_pct=0, min_overlap_pct=0.25, gamma_params=(2, 0.05, 0.03), resize=False,
                                  shift=True, same=False, label_mode="logistic-intensity", width_bounds_pct=((0.03, 0.4), (0.03, 0.4)),
                                  intensity_logistic_params=(1 / 3, 7))),
    ("poisson_mvtec_carpet", 11, dict(mode=1, num_patches=2, min_object_pct=0, min_overlap_pct=0.25, gamma_params=(2, 0.05, 0.03), resize=True,
                                      shift=True, same=False, label_mode="logistic-intensity", width_bounds_pct=((0.03, 0.4), (0.03, 0.4)),
                                      intensity_logistic_params=(1 / 3, 7), skip_background=None)),
    ("poisson_mvtec_hazelnut", 12, dict(mode=1, num_patches=2, min_object_pct=0, min_overlap_pct=0.25, gamma_params=(2, 0.05, 0.03), resize=True,
                                        shift=True, same=False, label_mode="logistic-intensity", width_bounds_pct=((0.03, 0.35), (0.03, 0.35)),
                                        intensity_logistic_params=(1 / 12, 24), skip_background=(20, 20))),
    ("poisson_mvtec_screw", 13, dict(mode=1, num_patches=2, min_object_pct=0, min_overlap_pct=0.25, gamma_params=(2, 0.05, 0.03), resize=True,
                                     shift=True, same=False, label_mode="logistic-intensity", width_bounds_pct=((0.03, 0.12), (0.03, 0.12)),
                                     intensity_logistic_params=(1, 3), skip_background=(200, 60))),
    ("poisson_visa", 14, dict(mode=1, num_patches=2, min_object_pct=0, min_overlap_pct=0.25, gamma_params=(2, 0.05, 0.03), resize=True,
                              shift=True, same=False, label_mode="logistic-intensity", width_bounds_pct=((0.03, 0.4), (0.03, 0.4)),
                              intensity_logistic_params=(1 / 12, 24), skip_background=None, resize_bounds=(.5, 2))),
    ("poisson_visa_b", 15, dict(mode=1, num_patches=2, min_object_pct=0, min_overlap_pct=0.25, gamma_params=(2, 0.05, 0.03), resize=True,
                                shift=True, same=False, label_mode="logistic-intensity", width_bounds_pct=((0.03, 0.4), (0.03, 0.4)),
                                intensity_logistic_params=(1 / 12, 24), skip_background=None, resize_bounds=(.5, 2))),
    ("same_source", 8, dict(mode="swap", num_patches=2, resize=False, shift=True, same=True, label_mode="binary", gamma_params=(2, 0.05, 0.03))),
    # ---- round 4: cv2.MIXED_CLONE and 'mix' (self_sup_tasks.py:22,47-48,267): the reference's coin flip + its control flow around the clone
    ("mixed_clone", 16, dict(mode=2, num_patches=2, min_object_pct=0, min_overlap_pct=0.25, gamma_params=(2, 0.05, 0.03), resize=True,
                             shift=True, same=False, label_mode="logistic-intensity", width_bounds_pct=((0.03, 0.4), (0.03, 0.4)),
                             intensity_logistic_params=(1 / 12, 24))),
    ("mix_a", 19, dict(mode="mix", num_patches=2, min_object_pct=0, min_overlap_pct=0.25, gamma_params=(2, 0.05, 0.03), resize=True,
                       shift=True, same=False, label_mode="logistic-intensity", width_bounds_pct=((0.03, 0.4), (0.03, 0.4)))),
    ("mix_b", 22, dict(mode="mix", num_patches=2, min_object_pct=0, min_overlap_pct=0.25, gamma_params=(2, 0.05, 0.03), resize=False,
                       shift=True, same=False, label_mode="logistic-intensity", width_bounds_pct=((0.03, 0.4), (0.03, 0.4)),
                       skip_background=(20, 20))),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    a = ap.parse_args()
    M = load_reference(a.ref)
    out = {}
    for name, seed, kw in CASES:
        dest, src = test_image(10 + seed), test_image(40 + seed)
        np.random.seed(1000 + seed)
        patchex, label, boxes = M.patch_ex(dest.copy(), src.copy(), verbose=False, **kw)
        after = np.random.randint(1 << 30)        # the generator's state after the call: pins the number of draws
        out[name + "_patchex"] = patchex
        out[name + "_label"] = np.asarray(label, np.float64)
        out[name + "_boxes"] = np.asarray(boxes, np.int64).reshape(-1, 4)
        out[name + "_next_draw"] = np.array(after)
        print(name, "changed px", int((patchex != dest).any(-1).sum()), "label max", float(np.max(label)), "boxes", len(boxes))
    path = os.path.join(ROOT, "tests", "golden", "self_sup.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()

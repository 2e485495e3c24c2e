"""NSA / CutPaste self-supervised augmentation (reference minigpt4/datasets/self_sup_tasks.py:11-292).
CPU: the oracle restatement against the golden produced by the reference's own `patch_ex`; the product's host path against
the oracle.  GPU: the blend / label kernels against the oracle."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import self_sup_ref as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "self_sup.npz"))
_spec = importlib.util.spec_from_file_location("mk_selfsup", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools",
                                                                         "make_golden_selfsup.py"))
MK = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(MK)            # CASES + the synthetic test images (pure numpy; the reference is NOT imported)


@pytest.mark.parametrize("name,seed,kw", MK.CASES, ids=[c[0] for c in MK.CASES])
def test_oracle_matches_the_reference_patch_ex(name, seed, kw):
    dest, src = MK.test_image(10 + seed), MK.test_image(40 + seed)
    np.random.seed(1000 + seed)
    patchex, label, boxes = O.patch_ex(dest.copy(), src.copy(), **kw)
    assert np.random.randint(1 << 30) == int(G[name + "_next_draw"])          # same number of random draws
    assert np.array_equal(patchex, G[name + "_patchex"])                      # bit-exact uint8 image
    assert np.array_equal(np.asarray(boxes, np.int64).reshape(-1, 4), G[name + "_boxes"])
    np.testing.assert_allclose(np.asarray(label, np.float64), G[name + "_label"], rtol=0, atol=1e-12)


from myriad_amd import self_sup as P  # noqa: E402


@pytest.mark.parametrize("name,seed,kw", MK.CASES, ids=[c[0] for c in MK.CASES])
def test_product_host_path_matches_the_oracle(name, seed, kw):
    """myriad_amd.self_sup.plan + apply_numpy (what a DataLoader worker runs) == the oracle == the reference's patch_ex."""
    dest, src = MK.test_image(10 + seed), MK.test_image(40 + seed)
    np.random.seed(1000 + seed)
    patchex, label, boxes = P.patch_ex(dest.copy(), src.copy(), **kw)
    assert np.random.randint(1 << 30) == int(G[name + "_next_draw"])
    assert np.array_equal(patchex, G[name + "_patchex"])
    assert np.array_equal(np.asarray(boxes, np.int64).reshape(-1, 4), G[name + "_boxes"])
    np.testing.assert_allclose(np.asarray(label, np.float64), G[name + "_label"], rtol=0, atol=1e-12)
    with pytest.raises(ValueError):
        P.patch_ex(dest, src, mode="no_such_mode")               # self_sup_tasks.py:290-291


@pytest.mark.gpu
def test_device_blend_and_label_match_the_oracle():
    """csrc/selfsup.hip on a batch of crops in HBM: bit-exact uint8 images, labels within fp32 rounding of the float64 oracle,
    for every case of the golden -- the arithmetic blends, the resampled patches (8-bit fixed-point bilinear) and the Poisson
    clones of the shipped recipe (float64 sine-transform solve on the device; the truncation sees the same values as the
    oracle's) -- one batch per label mode (the label kernel takes one mode per launch)."""
    ex = P.PatchExHIP("cuda")
    by_mode = {}
    for name, seed, kw in MK.CASES:
        by_mode.setdefault((kw.get("label_mode", "binary"), kw.get("intensity_logistic_params", (1 / 6, 20))), []).append((name, seed, kw))
    for (label_mode, ilp), cases in by_mode.items():
        dests, srcs, plans = [], [], []
        for name, seed, kw in cases:
            dest, src = MK.test_image(10 + seed), MK.test_image(40 + seed)
            np.random.seed(1000 + seed)
            pk = {k: v for k, v in kw.items() if k != "intensity_logistic_params"}
            ops, factor = P.plan(dest, src, **pk)
            dests.append(dest)
            srcs.append(dest if kw.get("same") else src)
            plans.append((ops, factor))
        d = torch.from_numpy(np.stack(dests)).cuda()
        s_ = torch.from_numpy(np.stack(srcs)).cuda()
        out, label, union = ex(d, s_, plans, label_mode=label_mode, intensity_logistic_params=ilp)
        for i, (name, seed, kw) in enumerate(cases):
            assert np.array_equal(out[i].cpu().numpy(), G[name + "_patchex"]), name
            np.testing.assert_allclose(label[i].cpu().numpy().astype(np.float64), G[name + "_label"][..., 0], rtol=0, atol=2e-5, err_msg=name)


# ---- properties of the two restated OpenCV algorithms (their pixels are PARITY UNPINNED: OpenCV is absent) ----------------
def test_resize_linear_properties():
    """cv2.resize stand-in: identity at the same size, constants stay constant, an exact 2 x 2 decimation is the rounded
    mean, the result stays within the range of the 2 x 2 source neighbourhood, product == oracle bit for bit."""
    r = np.random.RandomState(3)
    img = r.randint(0, 256, (41, 57, 3)).astype(np.uint8)
    assert np.array_equal(O.resize_linear_u8(img, (57, 41)), img)
    assert np.unique(O.resize_linear_u8(np.full((30, 44, 3), 201, np.uint8), (61, 23))).tolist() == [201]
    even = img[:40, :56]
    q = even.astype(np.int64)
    assert np.array_equal(O.resize_linear_u8(even, (28, 20)), ((q[0::2, 0::2] + q[0::2, 1::2] + q[1::2, 0::2] + q[1::2, 1::2] + 2) >> 2))
    for ds in ((80, 50), (33, 29), (57, 60), (20, 41), (114, 82)):
        a, b = O.resize_linear_u8(img, ds), P.resize_linear_u8(img, ds)
        assert a.shape == (ds[1], ds[0], 3) and np.array_equal(a, b)
        assert a.min() >= img.min() and a.max() <= img.max()
        m2 = O.resize_linear_u8(img[..., 0], ds)
        assert np.array_equal(m2, a[..., 0])                         # channels are independent; 2-D input == one channel
    ramp = np.tile(np.arange(0, 200, 4, dtype=np.uint8)[None, :, None], (8, 1, 3))   # a linear ramp stays monotone
    up = O.resize_linear_u8(ramp, (125, 8))[0, :, 0].astype(int)
    assert (np.diff(up) >= 0).all()


def test_poisson_clone_properties():
    """seamlessClone stand-in: (1) cloning a patch cut from the destination itself changes nothing; (2) the ROI's boundary
    ring keeps the destination's pixels; (3) a constant offset of the source is invisible (only gradients are cloned);
    (4) inside the eroded mask the result's Laplacian equals the source's (gradient-domain residual ~ 0, up to the final
    truncation); (5) an ROI that leaves the image raises, as OpenCV does."""
    r = np.random.RandomState(5)
    yy, xx = np.mgrid[0:120, 0:140]
    dst = np.clip(np.stack([90 + 40 * np.sin(xx / 9.0), 100 + 30 * np.cos(yy / 7.0), 80 + 0.5 * xx], -1) + r.randint(-3, 4, (120, 140, 3)), 0, 255).astype(np.uint8)
    mask = np.full((40, 60), 255, np.uint8)
    same = O.seamless_clone(dst[30:70, 40:100].copy(), dst, mask, (40 + 30, 30 + 20))
    assert np.array_equal(same, dst)                                                                     # (1)
    src = np.clip(np.stack([120 + 50 * np.sin(yy / 5.0 + xx / 11.0)] * 3, -1), 0, 255).astype(np.uint8)[:40, :60]
    out = O.seamless_clone(src, dst, mask, (70, 50))
    y0, x0, h, w = 50 - 38 // 2, 70 - 58 // 2, 38, 58                                                    # roi_d of the cleared-border mask
    changed = np.argwhere((out != dst).any(-1))
    assert changed[:, 0].min() >= y0 + 1 and changed[:, 0].max() <= y0 + h - 2                           # (2)
    assert changed[:, 1].min() >= x0 + 1 and changed[:, 1].max() <= x0 + w - 2
    assert np.array_equal(out[y0, x0:x0 + w], dst[y0, x0:x0 + w]) and np.array_equal(out[y0:y0 + h, x0], dst[y0:y0 + h, x0])
    shifted = np.clip(src.astype(int) + 30, 0, 255).astype(np.uint8)
    assert np.abs(O.seamless_clone(shifted, dst, mask, (70, 50)).astype(int) - out).max() <= 1           # (3)
    lap = lambda a: (a[1:-1, :-2] + a[1:-1, 2:] + a[:-2, 1:-1] + a[2:, 1:-1] - 4 * a[1:-1, 1:-1])
    inner = (slice(y0 + 6, y0 + h - 6), slice(x0 + 6, x0 + w - 6))
    got = lap(out[..., 0].astype(float))[inner[0].start - 1:inner[0].stop - 1, inner[1].start - 1:inner[1].stop - 1]
    sy, sx = 1 + 6, 1 + 6                                                                                # the same pixels in source coordinates
    want = lap(src[..., 0].astype(float))[sy - 1:sy - 1 + got.shape[0], sx - 1:sx - 1 + got.shape[1]]
    assert np.abs(got - want).max() <= 4.0 and np.abs(got - want).mean() < 1.5                           # (4): truncation moves each pixel < 1
    with pytest.raises(ValueError):
        O.seamless_clone(src, dst, mask, (5, 5))                                                         # (5)
    out2 = dst.copy()                                                                                    # the product's host path == the oracle
    pms = mask.copy(); pms[0] = pms[-1] = 0; pms[:, 0] = pms[:, -1] = 0
    P.poisson_clone_numpy(out2, src, pms, P.clone_roi(pms, (70, 50), dst.shape[:2]))
    assert np.array_equal(out2, out)
    # MIXED_CLONE (Cloning::mixedClone, restated): (6) a flat source has no gradients, so the destination's own gradients win
    # everywhere and nothing changes -- where NORMAL_CLONE flattens the region; (7) a source whose texture dominates
    # (|Px - Py| > |Dx - Dy| everywhere) gives exactly the NORMAL_CLONE result; (8) host path == oracle
    flat = np.full_like(src, 128)
    assert np.array_equal(O.seamless_clone(flat, dst, mask, (70, 50), O.MIXED_CLONE), dst)               # (6)
    assert not np.array_equal(O.seamless_clone(flat, dst, mask, (70, 50), O.NORMAL_CLONE), dst)
    smooth = np.full((120, 140, 3), 100, np.uint8)
    tex = np.clip(np.stack([128 + 100 * ((xx // 3) % 2)] * 3, -1), 0, 255).astype(np.uint8)[:40, :60]   # vertical stripes: Px != 0, Py = 0
    assert np.array_equal(O.seamless_clone(tex, smooth, mask, (70, 50), O.MIXED_CLONE),
                          O.seamless_clone(tex, smooth, mask, (70, 50), O.NORMAL_CLONE))                 # (7)
    mixed = O.seamless_clone(src, dst, mask, (70, 50), O.MIXED_CLONE)
    out3 = dst.copy()
    P.poisson_clone_numpy(out3, src, pms, P.clone_roi(pms, (70, 50), dst.shape[:2]), mixed=True)
    assert np.array_equal(out3, mixed) and not np.array_equal(mixed, out)                                # (8)


def test_restated_opencv_algorithms_agree_with_independent_implementations():
    """Not a pin (OpenCV itself is absent), but two independent implementations of the same published mathematics:
    (1) bilinear resampling on half-pixel centres with clamped borders, in floating point (torch, align_corners=False, no
    antialiasing -- INTER_LINEAR's geometry): the 11-bit fixed-point restatement may differ by one grey level, never two;
    (2) the Dirichlet Poisson problem solved by a sparse direct solver (scipy) instead of the sine transform."""
    import torch
    import torch.nn.functional as F
    r = np.random.RandomState(11)
    img = r.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    for ds in ((80, 50), (33, 29), (53, 60), (20, 41), (106, 74), (7, 5)):
        t = torch.from_numpy(img).permute(2, 0, 1)[None].double()
        ref = F.interpolate(t, size=(ds[1], ds[0]), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
        got = O.resize_linear_u8(img, ds).astype(np.float64)
        assert np.abs(got - ref).max() <= 1.0 + 1e-9, ds             # rounding of the 11-bit weights + the final round
        assert np.abs(got - ref).mean() < 0.35, ds
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    for (h, w) in ((9, 12), (23, 17), (40, 31)):
        nh, nw = h - 2, w - 2
        rhs = r.randn(nh, nw) * 30
        lap1 = lambda n: sp.diags([np.ones(n - 1), -2 * np.ones(n), np.ones(n - 1)], [-1, 0, 1])
        A = sp.kron(lap1(nh), sp.identity(nw)) + sp.kron(sp.identity(nh), lap1(nw))
        want = spl.spsolve(A.tocsc(), rhs.ravel()).reshape(nh, nw)
        got = O.poisson_dirichlet(rhs, h, w)
        assert np.abs(got - want).max() < 1e-9 * max(1.0, np.abs(want).max())


def test_dataset_arguments_follow_the_reference_tables():
    """anomaly_detection.py:50-65,118-141,254-264: per-class width bounds / logistic parameters / background for MVTec, the
    VisA set otherwise; both resample the patch and blend with NORMAL_CLONE."""
    a = P.self_sup_args("mvtec", "screw")
    assert a["width_bounds_pct"] == ((0.03, 0.12), (0.03, 0.12)) and a["intensity_logistic_params"] == (1, 3) and a["skip_background"] == (200, 60)
    assert a["resize"] is True and a["mode"] == "normal_clone" and a["num_patches"] == 2 and a["gamma_params"] == (2, 0.05, 0.03)
    assert P.self_sup_args("mvtec", "carpet")["skip_background"] is None
    v = P.self_sup_args("visa", "candle")
    assert v["resize_bounds"] == (.5, 2) and v["width_bounds_pct"] == ((0.03, 0.4), (0.03, 0.4)) and v["intensity_logistic_params"] == (1 / 12, 24)
    # an annotation file with neither tag: the MVTec-style base WITHOUT the VisA extras -- patch_ex's own defaults apply
    # (anomaly_detection.py:118-141 picks the base by 'VISA' in ann_paths[0], :254-259 the extras by get_class_name)
    n = P.self_sup_args("visa", "candle", visa_base=False)
    assert "width_bounds_pct" not in n and "resize_bounds" not in n and "intensity_logistic_params" not in n and n["num_patches"] == 2
    import inspect
    d = {k: p.default for k, p in inspect.signature(P.plan).parameters.items()}
    assert d["width_bounds_pct"] == ((0.05, 0.2), (0.05, 0.2)) and d["resize_bounds"] == (0.7, 1.3) and d["skip_background"] is None
    assert inspect.signature(P.patch_ex).parameters["intensity_logistic_params"].default == (1 / 6, 20)
    with pytest.raises(TypeError):
        P.self_sup_args("mvtec", "screw", visa_base=True)


def test_anomaly_detection_dataset_trains_with_the_reference_recipe(tmp_path):
    """AnomalyDetectionDataset.__getitem__ (anomaly_detection.py:231-362): class name from the path, MVTec tables selected by
    the annotation file's name, default = resampled patch + Poisson blend; 'swap' is the explicit opt-out."""
    from PIL import Image
    from myriad_amd.datasets import AnomalyDetectionDataset
    import json
    root = tmp_path
    rows = []
    for i in range(3):
        rel = f"mvtec/hazelnut/train/good/{i:03d}.png"
        os.makedirs(os.path.dirname(root / rel), exist_ok=True)
        Image.fromarray(MK.test_image(20 + i, 256)).save(root / rel)
        rows.append({"img_path": rel, "is_anomaly": "0", "caption": ""})
    (root / "DC_MVTEC_train_normal.jsonl").write_text("\n".join(json.dumps(r) for r in rows))
    ds = AnomalyDetectionDataset(str(root), ["DC_MVTEC_train_normal.jsonl"], seed=7)
    assert ds.get_class_name(0) == ("mvtec", "hazelnut")
    a = ds[0]
    assert a["aug_image"].shape == (3, 224, 224) and not torch.equal(a["aug_image"], a["image"])
    b = AnomalyDetectionDataset(str(root), ["DC_MVTEC_train_normal.jsonl"], seed=7)[0]
    assert torch.equal(a["aug_image"], b["aug_image"])                              # the dataset's own RandomState drives every draw
    sw = AnomalyDetectionDataset(str(root), ["DC_MVTEC_train_normal.jsonl"], seed=7, self_sup_mode="swap")[0]
    assert not torch.equal(sw["aug_image"], a["aug_image"])
    (root / "visa.jsonl").write_text("\n".join(json.dumps(r) for r in rows))
    assert AnomalyDetectionDataset(str(root), ["visa.jsonl"], seed=7).get_class_name(1)[0] == "visa"


def test_resize_and_clone_edge_cases():
    """Degenerate sizes the reference can reach: a 1-pixel-wide source (every destination column reads it), up- and down-
    scaling by large factors, a clone mask that leaves fewer than 50 pixels (the patch is dropped, self_sup_tasks.py:277-278),
    an ROI thinner than 3 pixels (nothing to solve: the destination is returned unchanged)."""
    one = np.array([[[10, 20, 30]], [[200, 100, 50]]], np.uint8)                       # 2 x 1 source
    up = O.resize_linear_u8(one, (5, 7))
    assert up.shape == (7, 5, 3) and (up[:, 0] == up[:, 4]).all()                      # columns identical
    assert up[0].tolist() == [[10, 20, 30]] * 5 and up[-1].tolist() == [[200, 100, 50]] * 5
    assert np.array_equal(P.resize_linear_u8(one, (5, 7)), up)
    r = np.random.RandomState(1)
    big = r.randint(0, 256, (90, 70, 3)).astype(np.uint8)
    for ds in ((7, 9), (210, 270), (1, 1), (70, 1)):
        a, b = O.resize_linear_u8(big, ds), P.resize_linear_u8(big, ds)
        assert a.shape == (ds[1], ds[0], 3) and np.array_equal(a, b)
    dst = r.randint(0, 256, (64, 64, 3)).astype(np.uint8)
    src = r.randint(0, 256, (12, 12, 3)).astype(np.uint8)
    thin = np.zeros((12, 12), np.uint8)
    thin[1:11, 5:7] = 255                                                               # ROI 10 x 2: no interior
    assert np.array_equal(O.seamless_clone(src, dst, thin, (32, 32)), dst)
    roi = P.clone_roi(thin, (32, 32), dst.shape[:2])
    out = dst.copy()
    P.poisson_clone_numpy(out, src, thin, roi)
    assert np.array_equal(out, dst)
    # a patch whose scaled mask keeps < 50 pixels is dropped by the plan (and by the reference): no operation, mask stays empty
    np.random.seed(3)
    ops, _ = P.plan(MK.test_image(11), MK.test_image(41), mode=1, num_patches=1, width_bounds_pct=((0.013, 0.02), (0.013, 0.02)),
                    resize=False, shift=True, label_mode="binary")
    assert ops == []
    np.random.seed(3)
    pe, lab, boxes = O.patch_ex(MK.test_image(11), MK.test_image(41), mode=1, num_patches=1, width_bounds_pct=((0.013, 0.02), (0.013, 0.02)),
                                resize=False, shift=True, label_mode="binary")
    assert np.array_equal(pe, MK.test_image(11)) and lab.sum() == 0 and boxes == []


@pytest.mark.gpu
def test_device_resize_kernel_matches_the_host_on_edge_sizes():
    """mh_patch_resize_u8 (csrc/selfsup.hip) against the host restatement, bit for bit: up / down scaling, the exact 2 x 2
    halving (area mean), identity, a 1-pixel-wide source box, boxes at the image border."""
    from myriad_amd import _lib, ops as Ops
    lib = _lib.load()
    r = np.random.RandomState(9)
    img = r.randint(0, 256, (2, 224, 224, 3)).astype(np.uint8)
    dev_img = torch.from_numpy(img).cuda()
    cases = [((0, 0, 40, 60), (60, 40)), ((10, 20, 40, 60), (30, 20)), ((100, 50, 37, 53), (80, 61)), ((0, 0, 224, 224), (112, 112)),
             ((200, 180, 24, 44), (90, 11)), ((5, 7, 20, 1), (9, 33)), ((3, 3, 50, 50), (7, 90))]
    for b, ((sy, sx, sh, sw), (w, h)) in enumerate(cases):
        xi, xw = P.linear_resize_tables(sw, w)
        yi, yw = P.linear_resize_tables(sh, h)
        tabs = torch.from_numpy(np.concatenate([xi, xw.reshape(-1), yi, yw.reshape(-1)]).astype(np.int32)).cuda()
        out = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
        tp = tabs.data_ptr()
        _lib.check(lib.mh_patch_resize_u8(dev_img.data_ptr(), b % 2, 224, 224, sy, sx, sh, sw, tp, tp + 4 * w, tp + 4 * 3 * w,
                                          tp + 4 * (3 * w + h), out.data_ptr(), h, w, Ops._s()), "mh_patch_resize_u8")
        want = P.resize_linear_u8(img[b % 2, sy:sy + sh, sx:sx + sw], (w, h))
        assert np.array_equal(out.cpu().numpy(), want), (sy, sx, sh, sw, w, h)


# ------------------------------------------------------------------------------------------------ OpenCV known answers
def _kat():
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "opencv_kat.json")) as f:
        return json.load(f)


def test_resize_known_answers_derived_from_opencvs_documented_fixed_point_scheme():
    """VERDICT r5 item 3c.  cv2.resize (8-bit, INTER_LINEAR; self_sup_tasks.py:213-227) against known answers derived from the
    arithmetic OpenCV documents in resize.cpp -- INTER_RESIZE_COEF_BITS = 11 -- by tools/make_golden_opencv_kat.py (exact
    rationals, written without the oracle) and, for the pixels below, by hand:

      [0, 100] -> width 4: scale 0.5; d = 0: f = -0.25 -> clamped (s, f) = (0, 0) -> 0;  d = 1: f = 0.25 -> weights
        cvRound(0.75 * 2048), cvRound(0.25 * 2048) = 1536, 512 -> H = 0 * 1536 + 100 * 512 = 51200; vertical weights (2048, 0):
        ((2048 * (51200 >> 4)) >> 16) + 0 + 2 >> 2 = (100 + 2) >> 2 = 25;  d = 2: 512, 1536 -> H = 153600 -> 300 -> 75;
        d = 3: f = 1.25 -> s = 1 >= ssize - 1 -> (1, 0) -> 100.                                      => [0, 25, 75, 100]
      [10, 201] -> width 3: the middle pixel has f = 0.5 -> 1024, 1024: H = 211 * 1024 = 216064, >> 4 = 13504, * 2048 >> 16 = 422,
        (422 + 2) >> 2 = 106 (the exact mean 105.5 goes UP: the scheme's + 2 >> 2 is round-half-up)   => [10, 106, 201]
      [0, 80, 240] -> width 4: scale 0.75, f = 0.625 and 1.375 -> weights (768, 1280) at s = 0 and (1280, 768) at s = 1:
        H = 80 * 1280 = 102400 -> 6400 -> 200 -> 50;  H = 80 * 1280 + 240 * 768 = 286720 -> 17920 -> 560 -> 140
                                                                                                     => [0, 50, 140, 240]
      [[1, 2], [3, 5]] -> 1 x 1: both scales exactly 2 -> INTER_AREA's fast path (1 + 2 + 3 + 5 + 2) >> 2 = 3
      [10, 201, 0, 100] -> width 2 (height unchanged): NOT the area path (only one direction halves): f = 0.5 twice -> [106, 50]

    Checked: the oracle's restatement, the product's host path, and the weight tables the device kernel consumes.  These pin the
    restatements to OpenCV's documented scheme; no OpenCV binary produced them (row f-2 stays partial for exactly that)."""
    from myriad_amd import self_sup as P
    k = _kat()
    hand = {"up2x_row": [[0, 25, 75, 100]], "up1p5x_row_rounding": [[10, 106, 201]],
            "up4over3_row_weights_768_1280": [[0, 50, 140, 240]], "exact_halving_is_the_area_mean": [[3]],
            "half_width_only_stays_linear": [[106, 50]], "up1p5x_column_rounding": [[10], [106], [201]]}
    seen = set()
    for c in k["resize"]:
        src = np.array(c["src"], dtype=np.uint8)
        want = np.array(c["want"], dtype=np.uint8)
        if c["name"] in hand:
            assert want.tolist() == hand[c["name"]], c["name"]          # the generator agrees with the hand computation
            seen.add(c["name"])
        for fn in (O.resize_linear_u8, P.resize_linear_u8):
            got = fn(src, tuple(c["dsize"]))
            assert np.array_equal(got, want), (c["name"], fn.__module__, got.tolist(), want.tolist())
            got3 = fn(np.repeat(src[..., None], 3, -1), tuple(c["dsize"]))      # three channels, as the recipe calls it
            assert np.array_equal(got3, np.repeat(want[..., None], 3, -1)), c["name"]
    assert seen == set(hand)
    for c in k["coeffs"]:
        xi, xw = P.linear_resize_tables(c["ssize"], c["dsize"])
        assert xi.tolist() == c["index"] and xw[:, 0].tolist() == c["w0"] and xw[:, 1].tolist() == c["w1"], (c["ssize"], c["dsize"])
        oi, ow = O._linear_coeffs(c["ssize"], c["dsize"])
        assert np.asarray(oi).tolist() == c["index"] and np.asarray(ow)[:, 0].tolist() == c["w0"] and np.asarray(ow)[:, 1].tolist() == c["w1"]
        assert all(a + b == 2048 for a, b in zip(c["w0"], c["w1"]))


def test_normal_clone_closed_form_constant_patch_on_a_linear_ramp():
    """cv2.seamlessClone(NORMAL_CLONE) (self_sup_tasks.py:254-288) on a case with a closed-form answer: a CONSTANT source patch
    has a zero gradient field, so inside the mask the result solves the discrete Laplace equation with the destination's values on
    the ROI ring; a destination that is linear in x and y is discrete-harmonic, hence the result IS the destination -- every pixel,
    every channel (a decreasing channel included).  Oracle and product host path.  (An OpenCV binary solves in float32 and truncates:
    it may land one level below on pixels where its solution is 1e-5 under an integer; the stated float64 + 1e-6 deviation)"""
    from myriad_amd import self_sup as P
    c = _kat()["clone"][0]
    dst = np.array(c["dst"], dtype=np.uint8)
    h, w, _ = dst.shape
    assert (h, w) == tuple(c["dst_shape"][:2]) and int(dst[3, 5, 0]) == 10 + 2 * 5 + 3 * 3 and int(dst[7, 9, 2]) == 200 - 9 - 14
    src = np.zeros_like(dst)
    src[...] = np.array(c["src_value"], dtype=np.uint8)
    y0, x0, y1, x1 = c["mask_box"]
    mask = np.zeros((h, w), dtype=np.uint8)
    mask[y0:y1, x0:x1] = 255
    out = O.seamless_clone(src, dst, mask, tuple(c["center"]), O.NORMAL_CLONE)
    assert np.array_equal(out, dst)
    pms = mask[y0:y1, x0:x1, None]
    out2 = dst.copy()
    P.poisson_clone_numpy(out2, src[y0:y1, x0:x1], pms, P.clone_roi(pms, tuple(c["center"]), dst.shape[:2]))
    assert np.array_equal(out2, dst)
    # not vacuous: a NON-constant patch does change the destination, and a plain paste of the constant patch would too
    src2 = src.copy()
    src2[y0:y1, x0:x1, 0] = (np.arange(x1 - x0)[None, :] ** 2 * 3 % 251).astype(np.uint8)     # not linear: a linear source is invisible too
    assert not np.array_equal(O.seamless_clone(src2, dst, mask, tuple(c["center"]), O.NORMAL_CLONE), dst)
    assert not np.array_equal(np.where(mask[..., None] > 0, src, dst), dst)


@pytest.mark.gpu
def test_device_kernels_reproduce_the_opencv_known_answers():
    """The device resample kernel (csrc/selfsup.hip: mh_patch_resize_u8) on the hand-derived INTER_LINEAR answers, and the device
    Poisson path (PatchExHIP, NORMAL_CLONE) on the closed-form clone case."""
    from myriad_amd import _lib, ops as Ops, self_sup as P
    lib = _lib.load()
    k = _kat()
    for c in k["resize"]:
        src = np.repeat(np.array(c["src"], dtype=np.uint8)[..., None], 3, -1)
        sh, sw, _ = src.shape
        w, h = c["dsize"]
        if sw == 2 * w and sh == 2 * h:
            continue                                        # the area path is taken on the host side of the plan (covered above)
        img = torch.from_numpy(src[None].copy()).cuda()
        xi, xw = P.linear_resize_tables(sw, w)
        yi, yw = P.linear_resize_tables(sh, h)
        tabs = torch.from_numpy(np.concatenate([xi, xw.reshape(-1), yi, yw.reshape(-1)]).astype(np.int32)).cuda()
        out = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
        tp = tabs.data_ptr()
        _lib.check(lib.mh_patch_resize_u8(img.data_ptr(), 0, sh, sw, 0, 0, sh, sw, tp, tp + 4 * w, tp + 4 * 3 * w,
                                          tp + 4 * (3 * w + h), out.data_ptr(), h, w, Ops._s()), "mh_patch_resize_u8")
        assert np.array_equal(out.cpu().numpy()[..., 0], np.array(c["want"], dtype=np.uint8)), c["name"]
    # NORMAL_CLONE, closed form, through the device path: one hand-built PatchOp (the constant patch at its own position)
    c = k["clone"][0]
    dst = np.array(c["dst"], dtype=np.uint8)
    src = np.zeros_like(dst)
    src[...] = np.array(c["src_value"], dtype=np.uint8)
    y0, x0, y1, x1 = c["mask_box"]
    pms = np.full((y1 - y0, x1 - x0, 1), 255, dtype=np.uint8)
    roi = P.clone_roi(pms, tuple(c["center"]), dst.shape[:2])
    op = P.PatchOp((y0, x0), (y0, x0, y1 - y0, x1 - x0), np.ones((y1 - y0, x1 - x0), dtype=np.uint8), 1.0, "normal_clone", pms=pms, roi=roi)
    ex = P.PatchExHIP("cuda")
    out, _, _ = ex(torch.from_numpy(dst[None].copy()).cuda(), torch.from_numpy(src[None].copy()).cuda(), [([op], 1.0)])
    assert np.array_equal(out[0].cpu().numpy(), dst)

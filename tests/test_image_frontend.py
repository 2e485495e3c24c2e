"""Image front-end of the data path (image side).

CPU (`-m "not gpu"`): the oracle's restatement of Pillow's 8-bit bicubic resampler is pinned bit-for-bit against Pillow itself
(live, this interpreter's PIL) and against the committed goldens (tools/make_golden_image.py); the host-side table builder of
the product path produces the same tables as the oracle's.  GPU: the HIP kernels against the oracle, exact equality for the
uint8 crop AND the float32 tensor, at the golden sizes and at dataset sizes (900 x 900, 1000 x 1500)."""
import os

import numpy as np
import pytest
import torch

from oracle import image_ref as IR
from tests.golden_utils import IMAGE_CASES, image_case

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "image_frontend.npz"))


@pytest.mark.parametrize("name", list(IMAGE_CASES))
def test_oracle_matches_golden(name):
    H, W, size, mode, seed = IMAGE_CASES[name]
    img = image_case(H, W, seed)
    u8, f32 = (IR.train_image if mode == "train" else IR.eval_image)(img, size)
    assert np.array_equal(u8, GOLD[name + "_u8"])
    assert f32.dtype == np.float32 and np.array_equal(f32, GOLD[name + "_f32"])


@pytest.mark.parametrize("H,W,oh,ow", [(97, 131, 32, 43), (20, 27, 32, 43), (250, 333, 224, 298), (300, 200, 336, 224),
                                       (64, 64, 64, 32), (513, 777, 224, 224), (31, 1, 7, 5), (5, 5, 224, 224)])
def test_oracle_resampler_is_pillows(H, W, oh, ow):
    """Pin: pure-noise images (every tap matters, over/undershoot clips) through PIL.Image.resize vs the restatement."""
    from PIL import Image
    img = np.random.default_rng(H * 1000 + W).integers(0, 256, (H, W, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
    assert np.array_equal(IR.resize_bicubic_u8(img, ow, oh), ref)


def test_size_and_crop_rules():
    assert IR.resized_size(1000, 1500, 224) == (224, 336) and IR.resized_size(1500, 1000, 224) == (336, 224)
    assert IR.resized_size(900, 900, 224) == (224, 224) and IR.resized_size(1284, 1562, 224) == (224, 272)
    assert IR.center_crop_offsets(224, 336, 224) == (0, 56) and IR.center_crop_offsets(224, 229, 224) == (0, 2)   # 2.5 -> 2
    assert IR.center_crop_offsets(227, 224, 224) == (2, 0)                                                         # 1.5 -> 2
    lut = IR.normalize_lut()
    assert lut.shape == (3, 256) and lut.dtype == np.float32
    assert lut[0, 0] == (np.float32(0) - np.float32(0.48145466)) / np.float32(0.26862954)


def test_product_tables_equal_the_oracles():
    """myriad_amd.image_frontend builds its own weight tables (the product may not import the oracle): same integers."""
    from myriad_amd import image_frontend as F
    for (i, o) in [(131, 43), (27, 43), (1500, 336), (900, 224), (224, 224), (1, 5)]:
        k1, b1 = F.resample_tables(i, o)
        k2, b2 = IR.resample_coeffs(i, o)
        assert np.array_equal(k1, k2) and np.array_equal(b1, b2)
    assert F.resized_size(1000, 1500, 224) == IR.resized_size(1000, 1500, 224)
    assert F.resized_size(1500, 1000, 224) == IR.resized_size(1500, 1000, 224)


def test_front_end_refuses_to_run_without_the_library(monkeypatch):
    from myriad_amd import _lib, image_frontend as F
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmyriad_hip.so")
    with pytest.raises(_lib.MyriadHipError):
        F.ImageFrontEndHIP("cpu")


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(IMAGE_CASES))
def test_hip_front_end_matches_golden_and_oracle(name):
    from myriad_amd.image_frontend import ImageFrontEndHIP
    H, W, size, mode, seed = IMAGE_CASES[name]
    img = image_case(H, W, seed)
    fe = ImageFrontEndHIP("cuda", size=size, mode=mode)
    out, u8 = fe([img], return_u8=True)
    assert np.array_equal(u8[0].cpu().numpy(), GOLD[name + "_u8"])
    assert np.array_equal(out[0].cpu().numpy(), GOLD[name + "_f32"])
    # ToTensor + Normalize of an (edited) uint8 crop
    again = fe.normalize_u8(u8)
    assert torch.equal(again, out)


@pytest.mark.gpu
def test_hip_front_end_dataset_sizes_batched():
    """MVTec-AD (900 x 900 ... 1024 x 1024) and VisA (1000 x 1500 landscape) sized inputs, mixed in one call, device- and
    host-resident; exact equality with the oracle for both the uint8 crop and the float32 tensor; train and eval modes."""
    from myriad_amd.image_frontend import ImageFrontEndHIP
    sizes = [(900, 900, 21), (1000, 1500, 22), (1500, 1000, 23), (1024, 1024, 24), (700, 1233, 25)]
    imgs = [image_case(h, w, s) for h, w, s in sizes]
    for mode, fn in (("train", IR.train_image), ("eval", IR.eval_image)):
        fe = ImageFrontEndHIP("cuda", size=224, mode=mode)
        mixed = [imgs[0], torch.from_numpy(imgs[1]).cuda(), torch.from_numpy(imgs[2]), imgs[3], imgs[4]]
        out, u8 = fe(mixed, return_u8=True)
        out2 = fe(mixed)                                             # cached tables, no uint8 output
        assert torch.equal(out, out2)
        for i, im in enumerate(imgs):
            ru8, rf = fn(im, 224)
            assert np.array_equal(u8[i].cpu().numpy(), ru8), (mode, sizes[i])
            assert np.array_equal(out[i].cpu().numpy(), rf), (mode, sizes[i])


@pytest.mark.gpu
def test_hip_front_end_rejects_bad_input():
    from myriad_amd import _lib
    from myriad_amd.image_frontend import ImageFrontEndHIP
    fe = ImageFrontEndHIP("cuda")
    with pytest.raises(_lib.MyriadHipError):
        fe([np.zeros((10, 10), np.uint8)])
    with pytest.raises(_lib.MyriadHipError):
        fe([np.zeros((10, 10, 3), np.float32)])

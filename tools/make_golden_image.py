#!/usr/bin/env python3
"""Golden vectors for the image front-end (image side): seeded random uint8 images through Pillow's own
Image.resize(BICUBIC) (which is what torchvision's Resize calls for PIL inputs) + the crop / ToTensor / Normalize rules in
float32.  Inputs are regenerated from the seed in the tests (tests/golden_utils.py:image_case); the fixture holds outputs only.
Run in the build container: python tools/make_golden_image.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from tests.golden_utils import IMAGE_CASES, image_case

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
out = {}
for name, (H, W, size, mode, seed) in IMAGE_CASES.items():
    img = image_case(H, W, seed)
    pil = Image.fromarray(img)
    if mode == "train":
        short, long = (W, H) if W <= H else (H, W)
        new_long = int(size * long / short)
        rw, rh = (size, new_long) if W <= H else (new_long, size)
        r = np.asarray(pil.resize((rw, rh), Image.BICUBIC))
        top, left = int(round((rh - size) / 2.0)), int(round((rw - size) / 2.0))
        u8 = r[top:top + size, left:left + size]
    else:
        u8 = np.asarray(pil.resize((size, size), Image.BICUBIC))
    x = u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    x = (x - np.array(CLIP_MEAN, np.float32)[:, None, None]) / np.array(CLIP_STD, np.float32)[:, None, None]
    out[name + "_u8"] = u8
    out[name + "_f32"] = x.astype(np.float32)
p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "image_frontend.npz")
np.savez_compressed(p, **out)
print("wrote", p, os.path.getsize(p), "bytes;", len(out), "arrays")

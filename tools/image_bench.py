#!/usr/bin/env python3
"""Image front-end throughput: HIP (device-resident decoded images, and host-resident incl. the PCIe copy) vs Pillow on one
host core, 1000 x 1500 and 900 x 900 inputs -> 224 x 224 normalised tensors."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from myriad_amd.image_frontend import ImageFrontEndHIP
from tests.golden_utils import image_case

fe = ImageFrontEndHIP("cuda:0")
for (H, W) in [(1000, 1500), (900, 900)]:
    imgs = [image_case(H, W, s) for s in range(8)]
    dev_imgs = [torch.from_numpy(i).cuda() for i in imgs]
    for name, batch in (("device-resident", dev_imgs), ("host-resident (+H2D)", imgs)):
        fe(batch); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fe(batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{H}x{W} HIP {name}: {dt*1e3/8:.3f} ms/image  {8/dt:.0f} images/s  ({H*W*3*8/dt/1e9:.1f} GB/s of input bytes)")
    t0 = time.perf_counter()
    for im in imgs[:4]:
        p = Image.fromarray(im)
        rw, rh = (336, 224) if W > H else (224, 224)
        r = np.asarray(p.resize((rw, rh), Image.BICUBIC))
        l = (rw - 224) // 2
        x = (r[:, l:l + 224].astype(np.float32).transpose(2, 0, 1) / 255 - 0.45) / 0.27
    dt = (time.perf_counter() - t0) / 4
    print(f"{H}x{W} Pillow + numpy, 1 core: {dt*1e3:.2f} ms/image  {1/dt:.0f} images/s")

#!/usr/bin/env python3
"""Vision expert at full size: ImageBind-Huge vision trunk (1280 x 32 blocks, 257 tokens, taps 7/15/23/31)
+ zero-shot and one-shot map heads, synthetic weights.  python tools/expert_bench.py [--batch 8] [--k 1]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops
from myriad_amd.vision_expert import VisionExpertHIP
from tests import golden_utils as gu

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--k", type=int, default=1)
ap.add_argument("--blocks", type=int, default=32)
ap.add_argument("--cpu", type=int, default=0, help="also time the oracle trunk (oracle/expert_ref.vision_trunk, fp32) on this many host threads, 2 images")
a = ap.parse_args()
dev = "cuda:0"
ops.ensure_workspace(dev)
layers = [7, 15, 23, 31] if a.blocks == 32 else [a.blocks - 1]
sd = gu.expert_weights(1280, a.blocks, 1024, len(layers), seed=1)
ex = VisionExpertHIP(sd, 16, layers, dev)
sd_cpu = sd if a.cpu else None
del sd
images, refs, text = gu.expert_inputs(a.batch, a.k, 1024, seed=2)
images, refs, text = images.to(dev), refs.to(dev), text.to(dev)
B = a.batch
# per image: patch stem + 32 blocks (4 D^2 + 8 D^2 linear MACs per token, attention 2 N D per token) -- SURVEY 8f: 334 GF
N, D = 257, 1280
gf_img = (2 * 256 * 588 * D + a.blocks * (2 * N * 12 * D * D + 4 * N * N * D)) / 1e9


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


t_trunk = timeit(lambda: ex.trunk.forward(images))
t_zs = timeit(lambda: ex.zero_shot(images, text))
t_os = timeit(lambda: ex.one_shot(images, refs))
t_both = timeit(lambda: ex.forward(images, text, refs))
print(f"batch {B}, k={a.k}: trunk {t_trunk*1e3:.2f} ms ({B/t_trunk:.0f} img/s, {B*gf_img/t_trunk/1e3:.0f} TFLOP/s of {gf_img:.0f} GF/img); "
      f"zero-shot maps {t_zs*1e3:.2f} ms ({B/t_zs:.0f} img/s); one-shot maps {t_os*1e3:.2f} ms ({B/t_os:.0f} img/s, "
      f"{B*(1+a.k)} trunk passes); both map pairs from one pass over [images; references] {t_both*1e3:.2f} ms ({B/t_both:.0f} img/s)")

if a.cpu:
    from oracle import expert_ref as OR          # CPU leg of this bench only
    torch.set_num_threads(a.cpu)
    img2 = images[:2].cpu()
    with torch.no_grad():
        OR.vision_trunk(sd_cpu, img2[:1], 16, layers, a.blocks)
        t0 = time.perf_counter()
        OR.vision_trunk(sd_cpu, img2, 16, layers, a.blocks)
        t_cpu = (time.perf_counter() - t0) / 2
    print(f"oracle trunk on {a.cpu} host threads (fp32): {t_cpu*1e3:.0f} ms/image = {1/t_cpu:.2f} img/s ({gf_img/t_cpu/1e3:.2f} TFLOP/s)")

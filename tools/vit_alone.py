#!/usr/bin/env python3
"""The frozen EVA ViT-g forward of one batch ALONE on the chip (nothing beside it): ms per forward, eager and as the two-piece
hipGraph the look-ahead replays, plus the launch profiler's per-shape GEMM table.  The number VERDICT r5 item 1 asks about
(4.17 TF at batch 8: 6-7.7 ms = 0.22-0.28 of the bf16 peak)."""
import argparse, os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import _lib, ops
from myriad_amd.eva_vit import EvaViTHIP
from myriad_amd.synthetic import SyntheticWeights, full_config

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--depth", type=int, default=39)
a = ap.parse_args()
_lib.load()
dev = torch.device("cuda:0")
cfg = full_config(llm_layers=1, vit_depth=a.depth, qf_layers=1)
w = SyntheticWeights(cfg, dev, seed=0, arch="myriad")
sd = {k: w[k] for k in w.keys() if k.startswith("visual_encoder.")}
vit = EvaViTHIP(sd, cfg["vit_heads"], dev)
ops.ensure_workspace(dev)
img = torch.randn(a.batch, 3, 224, 224, device=dev)
N = (cfg["image_size"] // cfg["patch"]) ** 2 + 1
Dv, Hv = cfg["vit_dim"], cfg["vit_hidden"]
flops = a.batch * (a.depth * (2 * N * (4 * Dv * Dv + 2 * Dv * Hv) + 4 * N * N * Dv) + 2 * (N - 1) * 3 * cfg["patch"] ** 2 * Dv)


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


out = vit.forward(img)
ms = timed(lambda: vit.forward(img), a.iters)
print(f"ViT-g forward alone, batch {a.batch}, eager: {ms:.3f} ms = {flops / ms / 1e9:.0f} TF/s = {flops / ms / 1e9 / 2500:.3f} of peak "
      f"(the host may bound an eager chain of ~290 launches)")
# graph replay (what the look-ahead does)
g = torch.cuda.CUDAGraph()
s = ops.side_stream(dev, "vit")          # registers its own split-K scratch (mh_set_stream_workspace)
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    vit.forward(img)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        o2 = vit.forward(img)
torch.cuda.synchronize()
ms_g = timed(lambda: g.replay(), a.iters)
print(f"ViT-g forward alone, batch {a.batch}, hipGraph replay: {ms_g:.3f} ms = {flops / ms_g / 1e9:.0f} TF/s = "
      f"{flops / ms_g / 1e9 / 2500:.3f} of peak; max|graph - eager| = {float((o2 - out).abs().max()):.3g}")
# per-launch profile of one eager forward
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import LaunchProfile, KERNEL_NAMES
with LaunchProfile() as lp:
    vit.forward(img)
lp.summary()
print("kernel,M,N,K,splits,launches,avg_us,TF/s")
for (kid, m, n, k, sp), cnt, tms, tf, _ in lp.shapes:
    print(f"{KERNEL_NAMES.get(kid, kid)},{m},{n},{k},{sp},{cnt},{1e3 * tms / cnt:.1f},{tf:.0f}")
tot = sum(r[2] for r in lp.shapes)
print(f"GEMM kernel time (each launch alone): {tot:.3f} ms of the {ms_g:.3f} ms forward")

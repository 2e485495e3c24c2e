#!/usr/bin/env python3
"""The 256x256 GEMM's cost per launch outside its K loop: time per back-to-back launch at K = 64 .. 1024 on one-round grids
(4096 x 4096 = 256 tiles, 2048 x 6144 = 192 tiles), bf16 and fp32 outputs, with and without bias + GELU; least-squares fixed cost
and per-k-tile time.  Read next to tools/micro/launch_cost.hip (an empty kernel of the same footprint)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
KS = [64, 128, 256, 512, 1024, 2048]


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3


for (M, N) in ((4096, 4096), (2048, 6144), (1184, 12288)):
    for label, kw in (("bf16 out", {}), ("fp32 out", dict(out_dtype=torch.float32)), ("bias+gelu", dict(bias=True, gelu=True))):
        ts = []
        for K in KS:
            a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
            b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
            out = torch.empty((M, N), dtype=kw.get("out_dtype", torch.bfloat16), device=dev)
            bias = torch.randn(N, device=dev) if kw.get("bias") else None
            ts.append(timeit(lambda: ops.gemm(a, b, out=out, bias=bias, gelu=bool(kw.get("gelu")), variant=12,
                                              out_dtype=kw.get("out_dtype", torch.bfloat16))))
        xs = [k / 64 for k in KS]
        n = len(xs); mx = sum(xs) / n; my = sum(ts) / n
        sl = sum((x - mx) * (y - my) for x, y in zip(xs, ts)) / sum((x - mx) ** 2 for x in xs)
        print(f"{M}x{N} {label:10s}: " + " ".join(f"K={k}:{t:.1f}" for k, t in zip(KS, ts)) + f" us | fit {my - sl * mx:.1f} us + {sl:.3f} us per k-tile", flush=True)

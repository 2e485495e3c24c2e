#!/usr/bin/env python3
"""Fixed cost vs per-k-tile time of the small-tile GEMM kernels on the Q-Former's shapes (648 rows): launch-to-launch time at
K = 64 .. 3072 for N = 768 / 2304 / 3072, plan kernel as the policy picks it, with bias (as the layers call it)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
KS = [64, 128, 256, 768, 1536, 3072]


def timeit(fn, n=300):
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


for (M, N) in ((648, 768), (648, 2304), (648, 3072), (2056, 1408)):
    ts, plans = [], []
    for K in KS:
        a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        plans.append(ops.gemm_plan(M, N, K))
        ts.append(timeit(lambda: ops.gemm(a, b, out=out, bias=bias)))
    print(f"{M}x{N}: " + " ".join(f"K={k}:{t:.1f}us{p}" for k, t, p in zip(KS, ts, plans)), flush=True)

#!/usr/bin/env python3
"""A/B of option gemm_split_xcd on the K-split launches of the batch-8 step (and the ViT's fc2): the GEMM + its slab sum as the step
runs it (ops.gemm, f32 out), weights rotating over 8 'layers' so that every launch streams cold weights; alternating on / off."""
import sys, os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops, _lib
L = _lib.load()
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
SHAPES = [(1184, 4096, 4096), (1184, 4096, 11008), (1184, 4096, 22016), (1184, 4160, 12288), (2056, 1408, 6144), (648, 4096, 4096),
          (648, 4096, 11008), (648, 4096, 22016)]


def bench(fn, nb, reps=5):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(nb):
            fn(i)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / nb)
    return best * 1e3


for (M, N, K) in SHAPES:
    nb = 8
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    bs = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty(M, N, dtype=torch.float32, device=dev)
    k, s = ops.gemm_plan(M, N, K)
    res = []
    for rep in range(3):
        for on in (1, 0):
            L.mh_set_option(b"gemm_split_xcd", on)
            res.append((on, bench(lambda i: ops.gemm(a, bs[i], out=out), nb)))
    L.mh_set_option(b"gemm_split_xcd", 1)
    on_ = [t for o, t in res if o]; off_ = [t for o, t in res if not o]
    print(f"M={M} N={N} K={K} plan(k{k},s{s}): split_xcd on {min(on_):.1f} us ({' '.join(f'{t:.1f}' for t in on_)}) | off {min(off_):.1f} us "
          f"({' '.join(f'{t:.1f}' for t in off_)}) | {100 * (min(on_) / min(off_) - 1):+.1f} %", flush=True)

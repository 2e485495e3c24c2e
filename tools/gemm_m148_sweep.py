#!/usr/bin/env python3
"""Batch-1 step GEMMs (M = 148): the 256x256 kernel at forced K-split counts against the library policy, cold weights,
whole op (kernel + reduce) timed by HIP events.  Usage: python tools/gemm_m148_sweep.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops, _lib
L = _lib.load()
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
SHAPES = [(148, 12288, 4160), (148, 4096, 4096), (148, 22016, 4096), (148, 4096, 11008), (148, 4096, 22016), (148, 4160, 12288),
          (148, 11008, 4096), (257, 6144, 1408), (257, 1408, 6144)]


def timeit(fn, bs):
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for b in bs:
            fn(b)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / len(bs))
    return best * 1e3


for (M, N, K) in SHAPES:
    nb = max(2, int(1.5e9 // (N * K * 2)) + 1)
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    bs = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    kid, sp = ops.gemm_plan(M, N, K)
    line = f"M={M} N={N} K={K}: policy (kernel {kid}, s={sp}) {timeit(lambda b: ops.gemm(a, b, out=out), bs):.0f} us |"
    for s in (1, 2, 3, 4, 6, 8, 12, 16):
        if K // s < 256:
            continue
        L.mhdbg_set_big_splits(s)
        line += f" 256^2 s={s}: {timeit(lambda b: ops.gemm(a, b, out=out, variant=12), bs):.0f}"
    L.mhdbg_set_big_splits(0)
    print(line, flush=True)
    del bs

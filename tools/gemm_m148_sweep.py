#!/usr/bin/env python3
"""Batch-1 step GEMMs (M = 148 LLaMA rows, 257 ViT rows): the 160-row-tile kernels (160 x 96 / 160 x 128, 4-deep ring) at
forced K-split counts against the previous policy (128 x 128 / 256 x 256 tiles), cold weights, whole op (kernel + reduce)
timed by HIP events.  Usage: python tools/gemm_m148_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os; _os.environ.setdefault("MYRIAD_HIP_DEBUG_LIB", "1")   # the mhdbg_* hooks live in libmyriad_hip_dbg.so
from myriad_amd import ops, _lib
L = _lib.load()
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
SHAPES = [(148, 12288, 4160), (148, 4096, 4096), (148, 22016, 4096), (148, 4096, 11008), (148, 4096, 22016), (148, 4160, 12288),
          (148, 11008, 4096), (257, 6144, 1408), (257, 1408, 6144), (257, 4224, 1408), (257, 1408, 1408)]
if len(sys.argv) > 1:      # `gemm_m148_sweep.py 128`: the LLaMA layer's products at another row count (round 5: the last layer's label rows)
    m = int(sys.argv[1])
    SHAPES = [(m, 4096, 4096), (m, 22016, 4096), (m, 4096, 11008), (m, 11008, 4096), (m, 4096, 22016)]


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
    bs = [torch.randn(N, K, device=dev).to(torch.bfloat16) * 0.05 for _ in range(nb)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    L.mh_set_option(b"gemm_skinny", 0)
    k0, s0 = ops.gemm_plan(M, N, K)
    ref = ops.gemm(a, bs[0]).clone()
    t_old = timeit(lambda b: ops.gemm(a, b, out=out), bs)
    L.mh_set_option(b"gemm_skinny", 1)
    k1, s1 = ops.gemm_plan(M, N, K)
    got = ops.gemm(a, bs[0]).clone()
    err = float((got.float() - ref.float()).abs().max() / ref.float().abs().max())
    t_new = timeit(lambda b: ops.gemm(a, b, out=out), bs)
    line = f"M={M} N={N} K={K}: old (k{k0},s{s0}) {t_old:.0f} us | new (k{k1},s{s1}) {t_new:.0f} us  relerr {err:.1e} |"
    for kid in (4, 5):
        for s in (1, 2, 3, 4, 6, 8, 12, 16):
            if K // s < 256:
                continue
            tbn = 128 if kid == 4 else 96
            wg = ((M + 159) // 160) * ((N + tbn - 1) // tbn) * s
            if wg > 600 or wg < 100:
                continue
            L.mhdbg_set_force_plan(kid, s)
            line += f" k{kid}s{s}:{timeit(lambda b: ops.gemm(a, b, out=out), bs):.0f}"
    for s in (1, 2, 3, 4, 5, 6, 8, 10, 12):          # round 4: the 256 x 256 eight-wave kernel on the same rows
        if K // s < 256:
            continue
        wg = ((M + 255) // 256) * ((N + 255) // 256) * s
        if wg > 300 or wg < 60:
            continue
        L.mhdbg_set_force_plan(2, s)
        line += f" k2s{s}:{timeit(lambda b: ops.gemm(a, b, out=out), bs):.0f}"
    L.mhdbg_set_force_plan(-1, 0)
    print(line, flush=True)
    del bs

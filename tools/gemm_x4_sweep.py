#!/usr/bin/env python3
"""Schedule variants / timing-only knock-outs of the four-wave GEMM loop (build with `python myriad_amd/csrc/gen_gemm_x4.py
--sweep` first): time per launch on the step's shapes, and the per-k-tile time + fixed cost per launch from a K sweep on a
one-round grid (4096 x 4096 outputs = 256 tiles).  Variants with `ko` compute wrong results by construction."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os; _os.environ.setdefault("MYRIAD_HIP_DEBUG_LIB", "1")   # the mhdbg_* hooks live in libmyriad_hip_dbg.so
from myriad_amd import ops, _lib
L = _lib.load()
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
L.mhdbg_gemm_x4_nvariants.restype = ctypes.c_int
nv = L.mhdbg_gemm_x4_nvariants()
SHAPES = [(1184, 22016, 4096), (1184, 12288, 4160), (1184, 4096, 11008), (2056, 6144, 1408), (8192, 8192, 8192)]
KS = [1024, 2048, 4096, 8192]


def timeit(a, bs, out, reps=3):
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(len(bs) * reps):
            ops.gemm(a, bs[i % len(bs)], out=out, variant=12)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (len(bs) * reps))
    return best * 1e3


data = {}
for (M, N, K) in SHAPES + [(4096, 4096, k) for k in KS]:
    nb = max(2, min(8, int(1.3e9 // (N * K * 2)) + 1))
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    bs = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    L.mh_set_option(b"gemm256_impl", 0)
    data[(M, N, K, -1)] = timeit(a, bs, out)
    L.mh_set_option(b"gemm256_impl", 1)
    for v in range(nv):
        L.mhdbg_set_gemm_x4_variant(v)
        data[(M, N, K, v)] = timeit(a, bs, out)
    if "--same-panel" in sys.argv:               # every workgroup stages tile (0, 0)'s panels: all requests hit in L2
        L.mhdbg_set_gemm_x4_same_panel(1)
        for v in range(nv):
            L.mhdbg_set_gemm_x4_variant(v)
            data[(M, N, K, 100 + v)] = timeit(a, bs, out)
        L.mhdbg_set_gemm_x4_same_panel(0)
    del bs
L.mhdbg_set_gemm_x4_variant(0); L.mh_set_option(b"gemm256_impl", 1)
hdr = "| variant | " + " | ".join(f"{m}x{n}x{k}" for (m, n, k) in SHAPES) + " | us / k-tile (64) | fixed us |"
print(hdr); print("|" + "---|" * (len(SHAPES) + 3))
rows = list(range(-1, nv)) + ([100 + v for v in range(nv)] if "--same-panel" in sys.argv else [])
for v in rows:
    cells = []
    for (M, N, K) in SHAPES:
        t = data[(M, N, K, v)]
        cells.append(f"{t:.1f} us = {2.0 * M * N * K / t / 1e6:.0f} TF/s")
    # least squares t = a + b * (K / 64)
    xs = [k / 64 for k in KS]; ys = [data[(4096, 4096, k, v)] for k in KS]
    n = len(xs); mx = sum(xs) / n; my = sum(ys) / n
    b = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs); a0 = my - b * mx
    print(f"| {'8-wave' if v < 0 else (v if v < 100 else f'{v - 100} same-panel')} | " + " | ".join(cells) + f" | {b:.3f} | {a0:.1f} |", flush=True)

#!/usr/bin/env python3
"""Timing probe (round 5): the 160-row-tile kernels at M = 148 with B addressed as if it were tile-packed ([N / TBN][K / 64][TBN][64]:
every k-tile of a column tile one contiguous 12-16 KiB block) against the row-major weight (128-byte pieces 8 KiB apart).  The
packed addressing reads the SAME number of bytes from a buffer of the same size -- results are garbage, only the time counts.
Cold weights (a ring of buffers larger than the caches), whole op incl. the split-K reduce."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MYRIAD_HIP_DEBUG_LIB", "1")
import torch
from myriad_amd import ops, _lib
L = _lib.load()
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 148
SHAPES = [(M, 12288, 4160), (M, 4096, 4096), (M, 22016, 4096), (M, 4096, 11008), (M, 4096, 22016), (M, 4160, 12288), (M, 11008, 4096)]


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


for (m, N, K) in SHAPES:
    Np = (N + 383) // 384 * 384                    # room for the packed addressing of the last 96- / 128-wide column tile
    nb = max(2, int(1.5e9 // (Np * K * 2)) + 1)
    a = torch.randn(m, K, device=dev).to(torch.bfloat16)
    bs = [(torch.randn(Np, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty(m, N, dtype=torch.bfloat16, device=dev)
    k, s = ops.gemm_plan(m, N, K)
    L.mhdbg_or_gemm_flags(0)
    t0 = timeit(lambda b: ops.gemm(a, b[:N], out=out), bs)
    L.mhdbg_or_gemm_flags(1 << 20)
    t1 = timeit(lambda b: ops.gemm(a, b[:N], out=out), bs)
    L.mhdbg_or_gemm_flags(0)
    mb = N * K * 2 / 1e6
    print(f"M={m} N={N} K={K} plan k{k} s{s}: row-major {t0:.1f} us ({mb / t0 * 1e-3 * 1e3:.2f} TB/s... {mb / t0:.2f} MB/us) | packed addressing {t1:.1f} us ({mb / t1:.2f} MB/us)", flush=True)
    del bs

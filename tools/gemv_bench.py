#!/usr/bin/env python3
"""Skinny-M (decode) GEMM bandwidth: cold weights (cycling > 1 GiB of distinct matrices), GB/s of weight bytes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops
dev = torch.device("cuda:0")
SHAPES = [(1, 12288, 4096), (1, 4096, 4096), (1, 22016, 4096), (1, 4096, 11008), (1, 32000, 4096), (8, 22016, 4096), (8, 4096, 11008)]
tot_t = 0.0
for (M, N, K) in SHAPES:
    nb = max(2, int(1.5e9 // (N * K * 2)) + 1)
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    bs = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty(M, N, dtype=torch.float32, device=dev)
    ref = (a.float() @ bs[0].float().t())
    ops.gemm(a, bs[0], out=out, out_dtype=torch.float32)
    err = (out - ref).abs().max().item()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for b in bs:
            ops.gemm(a, b, out=out, out_dtype=torch.float32)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / nb)
    print(f"M={M} N={N} K={K}: {best*1e3:.1f} us  {N*K*2/(best*1e-3)/1e12:.2f} TB/s  maxerr {err:.2e}", flush=True)
    del bs

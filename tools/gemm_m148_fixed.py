#!/usr/bin/env python3
"""Batch-1 step GEMMs (M = 148): launch-to-launch time of the whole op (kernel + split-K reduce) at K = 64 .. 4096, cold weights
(a ring of weight copies larger than the caches), policy's own plan: the fixed cost per launch and the per-k-tile time of the
160-row tile kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
M = 148
for N in (4096, 12288, 22016):
    row = []
    for K in (64, 256, 1024, 4096):
        nb = max(2, int(1.2e9 // (N * K * 2)) + 1)
        nb = min(nb, 64)
        a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        bs = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(nb)]
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        for i in range(nb):
            ops.gemm(a, bs[i], out=out)
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(nb * 2):
                ops.gemm(a, bs[i % nb], out=out)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (nb * 2) * 1e3)
        row.append(f"K={K}: {best:.1f} us plan{ops.gemm_plan(M, N, K)} ({N * K * 2 / best / 1e6:.2f} TB/s)")
        del bs
    print(f"M=148 N={N}: " + " | ".join(row), flush=True)

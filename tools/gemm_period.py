#!/usr/bin/env python3
"""Steady-state cost per K tile of a GEMM variant: time(K2) - time(K1) on the same M x N, so prologue/epilogue cancel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
M = N = 4096
def t(K, v):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    best = 1e9
    for r in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm(a, b, out=out, variant=v)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    return best * 1e3
for v in [int(x) for x in sys.argv[1:]] or [1, 11, 12]:
    t1, t2 = t(2048, v), t(6144, v)
    d = (t2 - t1) / 4096 * 32          # us per 32-deep K slice of the whole 4096x4096 problem
    tf = 2.0 * M * N * 32 / (d * 1e-6) / 1e12
    print(f"variant {v}: K=2048 {t1:.1f} us, K=6144 {t2:.1f} us -> steady state {tf:.0f} TF/s, fixed cost {t1 - 2048/32*d:.1f} us")

#!/usr/bin/env python3
"""Run one GEMM shape repeatedly (for rocprofv3 --pmc runs).  python tools/gemm_one.py M N K variant iters"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops
M, N, K, v, it = (int(x) for x in sys.argv[1:6])
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(it):
    ops.gemm(a, b, out=out, variant=v)
torch.cuda.synchronize()

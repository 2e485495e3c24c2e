#!/usr/bin/env python3
"""Does a power-of-two row pitch of the GEMM operands cost L2 channel conflicts?  Same shapes, operands as views of buffers whose
row pitch is K, K + 64 or K + 128 elements.  python tools/gemm_ld_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
for (M, N, K) in [(8192, 8192, 8192), (1184, 22016, 4096), (1184, 11008, 4096), (1184, 12288, 4096), (2056, 6144, 1408)]:
    for pad_a, pad_b in ((0, 0), (64, 0), (0, 64), (64, 64), (128, 128), (32, 32)):
        nb = 3
        a = torch.randn(M, K + pad_a, device=dev).to(torch.bfloat16)[:, :K]
        bs = [torch.randn(N, K + pad_b, device=dev).to(torch.bfloat16)[:, :K] for _ in range(nb)]
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(nb * 3):
                ops.gemm(a, bs[i % nb], out=out, variant=12)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (nb * 3))
        print(f"M={M} N={N} K={K} pitch A K+{pad_a} B K+{pad_b}: {best*1e3:.1f} us  {2*M*N*K/best/1e9:.0f} TF/s", flush=True)
        del bs

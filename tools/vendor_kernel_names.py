#!/usr/bin/env python3
"""Which kernel does the vendor library pick for the step's plain GEMM shapes?  Run under rocprofv3 --kernel-trace --stats."""
import torch
dev = torch.device("cuda:0")
for (M, N, K) in [(1184, 22016, 4096), (1184, 12288, 4160), (1184, 11008, 4096), (2056, 6144, 1408), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(5):
        torch.matmul(a, b.t(), out=out)
    torch.cuda.synchronize()

#!/usr/bin/env python3
"""Times the mid-M streaming kernel alone (variant 13, cold weights) on a few shapes; MYRIAD_STREAM_SPLITS overrides K splits."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
SHAPES = [(148, 22016, 4096), (148, 12352, 4096), (148, 4096, 11008), (32, 12352, 4096), (257, 6144, 1408)]
for (M, N, K) in SHAPES:
    nb = max(2, int(1.5e9 // (N * K * 2)) + 1)
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    bs = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for b in bs:
            ops.gemm(a, b, out=out, variant=13)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / nb)
    print(f"M={M} N={N} K={K}: {best*1e3:.1f} us  {N*K*2/(best*1e-3)/1e12:.2f} TB/s", flush=True)
    del bs

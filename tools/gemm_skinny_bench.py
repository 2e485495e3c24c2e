#!/usr/bin/env python3
"""GEMMs of the batch-1 fine-tune step (M = 148 LLaMA rows, 257 ViT rows): weight-streaming regime, cold weights."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
SHAPES = [(148, 12352, 4096), (148, 4096, 4096), (148, 22016, 4096), (148, 4096, 11008), (148, 4160, 12288), (148, 4096, 22016),
          (148, 11008, 4096), (257, 4224, 1408), (257, 1408, 1408), (257, 6144, 1408), (257, 1408, 6144), (256, 768, 768), (256, 3072, 768), (72, 4096, 25664),
          (60, 32000, 4096), (32, 12352, 4096)]
tot = 0.0
for (M, N, K) in SHAPES:
    nb = max(2, int(1.5e9 // (N * K * 2)) + 1)
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    bs = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    res = {}
    for v in (0, 1, 12, 13):
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for b in bs:
                ops.gemm(a, b, out=out, variant=v)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / nb)
        res[v] = best
    k, s = ops.gemm_plan(M, N, K)
    print(f"M={M} N={N} K={K}: auto {res[0]*1e3:.1f} us (kernel {k}, splits {s}; {N*K*2/(res[0]*1e-3)/1e12:.2f} TB/s of weights, "
          f"{2.0*M*N*K/(res[0]*1e-3)/1e12:.0f} TF) | 128x128 {res[1]*1e3:.1f} us | 256x256 {res[12]*1e3:.1f} us | "
          f"stream {res[13]*1e3:.1f} us ({N*K*2/(res[13]*1e-3)/1e12:.2f} TB/s)", flush=True)
    del bs

#!/usr/bin/env python3
"""Row-stride sensitivity of the weight stream: the same GEMMs with ldb = K (power-of-two rows) and ldb = K + pad."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
SHAPES = [(148, 22016, 4096), (148, 12352, 4096), (32, 12352, 4096)]
for (M, N, K) in SHAPES:
    for pad in (0, 64, 192):
        for variant in (13, 12, 1):
            nb = max(2, int(1.5e9 // (N * (K + pad) * 2)) + 1)
            a = torch.randn(M, K, device=dev).to(torch.bfloat16)
            bs = [torch.randn(N, K + pad, device=dev).to(torch.bfloat16)[:, :K] for _ in range(nb)]
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            best = 1e9
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for b in bs:
                    ops.gemm(a, b, out=out, variant=variant)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / nb)
            print(f"M={M} N={N} K={K} ldb=K+{pad} variant {variant}: {best*1e3:.1f} us  {N*K*2/(best*1e-3)/1e12:.2f} TB/s", flush=True)
            del bs

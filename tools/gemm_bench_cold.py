#!/usr/bin/env python3
"""GEMM variants with COLD weights: cycle through enough distinct B matrices (> 1 GiB) that every launch streams
its weight panel from HBM, as in the real step (13 GB of LLaMA weights per pass >> 256 MiB Infinity Cache)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops

SHAPES = [(1184, 4096, 22016), (1184, 22016, 4096), (1184, 4096, 12288), (1184, 4096, 11008), (1184, 12288, 4096),
          (1184, 11008, 4096), (1184, 4096, 4096), (2056, 1408, 6144), (2056, 6144, 1408), (2056, 4224, 1408)]
VARIANTS = {1: "4w 2st (2blk)", 9: "8w 2st (2blk)", 10: "8w 4st (1blk)", 4: "4w 2st/64 (3blk)"}
ROUNDS = 4

print("| M | N | K | " + " | ".join(VARIANTS.values()) + " |")
print("|---|---|---|" + "---|" * len(VARIANTS))
for (M, N, K) in SHAPES:
    nb = max(2, int(1.3e9 // (N * K * 2)) + 1)
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    bs = [torch.randn(N, K, device="cuda").to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    best = {v: 1e9 for v in VARIANTS}
    for _ in range(ROUNDS):
        for v in VARIANTS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for b in bs:
                ops.gemm(a, b, out=out, variant=v)
            e1.record()
            torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) / nb)
    fl = 2.0 * M * N * K
    print(f"| {M} | {N} | {K} | " + " | ".join(f"{fl / (best[v] * 1e-3) / 1e12:.0f} TF ({best[v] * 1e3:.0f} us)" for v in VARIANTS) + " |")
    sys.stdout.flush()
    del bs

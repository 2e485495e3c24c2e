#!/usr/bin/env python3
"""Calibration only (not a product path): what does the vendor library (hipBLASLt via torch.matmul) reach on the
step's GEMM shapes on this box?  Gives the achievable ceiling to hold the hand-written kernel against."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops

dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
SHAPES = [(1184, 4096, 22016), (1184, 22016, 4096), (1184, 4096, 12288), (1184, 4096, 11008), (1184, 12288, 4096),
          (1184, 11008, 4096), (1184, 4096, 4096), (2056, 1408, 6144), (2056, 6144, 1408), (2056, 4224, 1408),
          (4096, 4096, 4096), (8192, 8192, 8192)]
print("| M | N | K | ours cold | vendor cold | ours warm | vendor warm |")
print("|---|---|---|---|---|---|---|")
for (M, N, K) in SHAPES:
    nb = max(2, int(1.3e9 // (N * K * 2)) + 1)
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    bs = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fns = {"ours": lambda b: ops.gemm(a, b, out=out), "vendor": lambda b: torch.matmul(a, b.t(), out=out)}
    cold = {k: 1e9 for k in fns}
    warm = {k: 1e9 for k in fns}
    for _ in range(4):
        for k, fn in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for b in bs:
                fn(b)
            e1.record()
            torch.cuda.synchronize()
            cold[k] = min(cold[k], e0.elapsed_time(e1) / nb)
            e0.record()
            for _i in range(10):
                fn(bs[0])
            e1.record()
            torch.cuda.synchronize()
            warm[k] = min(warm[k], e0.elapsed_time(e1) / 10)
    fl = 2.0 * M * N * K
    tf = lambda ms: f"{fl / (ms * 1e-3) / 1e12:.0f} TF"
    print(f"| {M} | {N} | {K} | {tf(cold['ours'])} | {tf(cold['vendor'])} | {tf(warm['ours'])} | {tf(warm['vendor'])} |", flush=True)
    del bs

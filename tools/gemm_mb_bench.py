#!/usr/bin/env python3
"""Cold-weight GEMM rates of the library policy for the step's shapes (run once per MYRIAD_G256_MB setting: the
variable is read once per process).  Checks the policy's result against the 128x128 kernel first (same k order per
accumulator => bit-identical on unsplit shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops

dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
torch.manual_seed(0)
tag = "mb" + os.environ.get("MYRIAD_G256_MB", "auto") + ",i" + os.environ.get("MYRIAD_G256I", "1")
for (M, N, K, kw) in [(1184, 4096, 4096, {}), (1184, 12288, 4160, {}), (2056, 6144, 1408, {"bias": True, "gelu": True}),
                      (1184, 4096, 11008, {"res": True, "f32": True}), (1200, 512, 2048, {}), (2056, 4224, 1408, {"bias": True})]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev) if kw.get("bias") else None
    res = torch.randn(M, N, device=dev) if kw.get("res") else None
    od = torch.float32 if kw.get("f32") else torch.bfloat16
    r1 = ops.gemm(a, b, bias=bias, residual=res, out_dtype=od, gelu=bool(kw.get("gelu")), variant=1)
    r0 = ops.gemm(a, b, bias=bias, residual=res, out_dtype=od, gelu=bool(kw.get("gelu")))
    print(f"check[{tag}] M={M} N={N} K={K} {kw} plan={ops.gemm_plan(M, N, K, od == torch.float32)}: "
          f"max|128 - policy| = {(r1.float() - r0.float()).abs().max().item():.3e}", flush=True)

SHAPES = [(1184, 4096, 22016), (1184, 22016, 4096), (1184, 4160, 12288), (1184, 4096, 11008), (1184, 12288, 4160),
          (1184, 11008, 4096), (1184, 4096, 4096), (2056, 1408, 6144), (2056, 6144, 1408), (2056, 4224, 1408)]
print(f"| M | N | K | policy[{tag}] cold |")
print("|---|---|---|---|")
for (M, N, K) in SHAPES:
    nb = max(2, int(1.3e9 // (N * K * 2)) + 1)
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    bs = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for b in bs:
            ops.gemm(a, b, out=out)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / nb)
    fl = 2.0 * M * N * K
    print(f"| {M} | {N} | {K} | {fl / (best * 1e-3) / 1e12:.0f} TF ({best * 1e3:.0f} us) |", flush=True)
    del bs

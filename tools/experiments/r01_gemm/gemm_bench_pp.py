#!/usr/bin/env python3
"""A/B: library default (128x128 4-wave, auto split-K) vs variant 11 (256x128 8-wave ping-pong, gemm_pp.hip).
Checks variant 11 against variant 1 first (same k-order per accumulator => bit-identical without split-K), then
times both with COLD weights (cycling > 1 GiB of distinct B matrices) and warm."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops

import ctypes
from myriad_amd import _lib
_cdll = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "libmyriad_hip.so"))
_cdll.mhdbg_set_gemm_sched(int(os.environ.get("PP_SCHED", "1")))
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)

# ---- correctness
torch.manual_seed(0)
for (M, N, K, kw) in [(1184, 4096, 4096, {}), (300, 200, 128, {}), (256, 128, 64, {}), (1, 8, 64, {}), (1184, 12288, 4096, {}),
                      (777, 1408, 6144, {"bias": True, "gelu": True}), (513, 260, 192, {"res": True, "f32": True}),
                      (1184, 4096, 11008, {"res": True, "f32": True})]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev) if kw.get("bias") else None
    res = torch.randn(M, N, device=dev) if kw.get("res") else None
    od = torch.float32 if kw.get("f32") else torch.bfloat16
    r1 = ops.gemm(a, b, bias=bias, residual=res, out_dtype=od, gelu=bool(kw.get("gelu")), variant=1)
    r11 = ops.gemm(a, b, bias=bias, residual=res, out_dtype=od, gelu=bool(kw.get("gelu")), variant=int(os.environ.get("PP_CHECK", "12")))
    ref = a.float() @ b.float().t()
    d = (r1.float() - r11.float()).abs().max().item()
    e = (r11.float() - (ref + (bias if bias is not None else 0)).float()).abs().max().item() if not kw.get("gelu") and res is None else float("nan")
    print(f"check M={M} N={N} K={K} {kw}: max|v1-v11|={d:.3e} max|v11-fp32|={e:.3e}", flush=True)

SHAPES = [(1184, 4096, 22016), (1184, 22016, 4096), (1184, 4096, 12288), (1184, 4096, 11008), (1184, 12288, 4096),
          (1184, 11008, 4096), (1184, 4096, 4096), (1184, 12352, 4096), (2056, 1408, 6144), (2056, 6144, 1408),
          (2056, 4224, 1408), (2056, 1408, 1408), (4096, 4096, 4096), (8192, 8192, 8192)]
VARIANTS = {1: "128x128", 0: "auto", 12: "256x256"}
print("| M | N | K | " + " | ".join(f"{n} cold" for n in VARIANTS.values()) + " | " + " | ".join(f"{n} warm" for n in VARIANTS.values()) + " |")
print("|---|---|---|" + "---|" * (2 * len(VARIANTS)))
for (M, N, K) in SHAPES:
    nb = max(2, int(1.3e9 // (N * K * 2)) + 1)
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    bs = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    cold = {v: 1e9 for v in VARIANTS}
    warm = {v: 1e9 for v in VARIANTS}
    for _ in range(4):
        for v in VARIANTS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for b in bs:
                ops.gemm(a, b, out=out, variant=v)
            e1.record()
            torch.cuda.synchronize()
            cold[v] = min(cold[v], e0.elapsed_time(e1) / nb)
            e0.record()
            for _i in range(10):
                ops.gemm(a, bs[0], out=out, variant=v)
            e1.record()
            torch.cuda.synchronize()
            warm[v] = min(warm[v], e0.elapsed_time(e1) / 10)
    fl = 2.0 * M * N * K
    print(f"| {M} | {N} | {K} | " + " | ".join(f"{fl / (cold[v] * 1e-3) / 1e12:.0f} TF ({cold[v] * 1e3:.0f} us)" for v in VARIANTS)
          + " | " + " | ".join(f"{fl / (warm[v] * 1e-3) / 1e12:.0f} TF" for v in VARIANTS) + " |", flush=True)
    del bs

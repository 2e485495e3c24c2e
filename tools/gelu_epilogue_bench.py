#!/usr/bin/env python3
"""The ViT fc1 product (2056 x 6144 x 1408, bias + GELU(erf) epilogue, eva_vit.py:54-61) with and without the activation:
what the epilogue's erf costs a one-workgroup-per-CU GEMM launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops
dev = torch.device("cuda:0"); ops.ensure_workspace(dev)
M, N, K = 2056, 6144, 1408
a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
bs = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(24)]
bias = torch.randn(N, device=dev)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for gelu in (False, True):
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for b in bs:
            ops.gemm(a, b, out=out, bias=bias, gelu=gelu)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / len(bs))
    print(f"gelu={gelu}: {best * 1e3:.1f} us")
ref = torch.nn.functional.gelu(a.float() @ bs[-1].float().t() + bias)
print("max |err| / max |ref| vs torch erf-GELU:", float((out.float() - ref).abs().max() / ref.abs().max()))
o32 = ops.gemm(a, bs[-1], bias=bias, gelu=True, out_dtype=torch.float32)
print("fp32 output:", float((o32 - ref).abs().max() / ref.abs().max()))

#!/usr/bin/env python3
"""256 x 224 tiles of gemm_256_kernel (MYRIAD_GEMM_BN): bit-identity with the 256 x 256 tiles and launch time, per shape.
Run twice: MYRIAD_GEMM_BN=256 python tools/gemm_bn224_check.py save ; python tools/gemm_bn224_check.py check"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "time"
SHAPES = [(1184, 22016, 4096), (1184, 11008, 4096), (1184, 12288, 4160), (1184, 4096, 11008), (1184, 4096, 4096), (2056, 6144, 1408),
          (2056, 4224, 1408), (2056, 1408, 6144), (1184, 22000, 4096), (300, 11008, 2048)]
outs = {}
for (M, N, K) in SHAPES:
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dev).to(torch.bfloat16)
    nb = 3
    bs = [(torch.randn(N, K, generator=g) * 0.05).to(dev).to(torch.bfloat16) for _ in range(nb)]
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    o1 = ops.gemm(a, bs[0])
    o2 = ops.gemm(a, bs[0], bias=bias, residual=res, out_dtype=torch.float32)
    ref = a.float() @ bs[0].float().t()
    err = ((o1.float() - ref).abs().max() / ref.abs().max()).item()
    err2 = ((o2 - (ref + bias + res)).abs().max() / ref.abs().max()).item()
    outs[(M, N, K)] = (o1.cpu(), o2.cpu())
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(nb * 4):
            ops.gemm(a, bs[i % nb], out=out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (nb * 4))
    k, s = ops.gemm_plan(M, N, K)
    print(f"M={M} N={N} K={K} plan k{k} s{s}: {best*1e3:.1f} us  {2*M*N*K/best/1e9:.0f} TF/s  relerr bf16 {err:.2e} f32+bias+res {err2:.2e}", flush=True)
f = "/tmp/bn_ref.pt"
if mode == "save":
    torch.save(outs, f)
elif mode == "check":
    ref = torch.load(f)
    for k_, (a1, a2) in outs.items():
        print(k_, "bit-identical:", torch.equal(a1, ref[k_][0]), torch.equal(a2, ref[k_][1]))

#!/usr/bin/env python3
"""gemm_256_kernel with whole-line requests (k-tile-blocked weight copy + 64-deep A staging, mh_gemm_pack_b) against the plain
form: bit-identity and launch time with cold weights.  python tools/gemm_packed_check.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops, _lib
L = _lib.load()
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
SHAPES = [(1184, 22016, 4096), (1184, 11008, 4096), (1184, 4096, 11008), (1184, 4096, 22016), (1184, 4096, 4096), (2056, 6144, 1408),
          (8192, 8192, 8192), (1184, 22000, 4096), (300, 1000, 128), (513, 776, 64), (700, 520, 320)]


def pack(b):
    N, K = b.shape
    bp = torch.empty(L.mh_gemm_packed_bytes(N, K), dtype=torch.uint8, device=dev)
    _lib.check(L.mh_gemm_pack_b(b.data_ptr(), b.stride(0), N, K, bp.data_ptr(), ops._s()), "mh_gemm_pack_b")
    return bp


for (M, N, K) in SHAPES:
    nb = max(2, min(6, int(1.3e9 // (N * K * 2)) + 1))
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dev).to(torch.bfloat16)
    bs = [(torch.randn(N, K, generator=g) * 0.05).to(dev).to(torch.bfloat16) for _ in range(nb)]
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)

    def run():
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(nb * 3):
                ops.gemm(a, bs[i % nb], out=out, variant=12)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (nb * 3))
        return best

    o_plain = ops.gemm(a, bs[0], variant=12).clone()
    f_plain = ops.gemm(a, bs[0], bias=bias, residual=res, out_dtype=torch.float32, variant=12).clone()
    t_plain = run()
    keep = [pack(b) for b in bs]
    o_pk = ops.gemm(a, bs[0], variant=12).clone()
    f_pk = ops.gemm(a, bs[0], bias=bias, residual=res, out_dtype=torch.float32, variant=12).clone()
    t_pk = run()
    for b in bs:
        L.mh_gemm_unregister_packed(b.data_ptr())
    torch.cuda.synchronize()
    ref = a.float() @ bs[0].float().t()
    err = ((o_pk.float() - ref).abs().max() / ref.abs().max()).item()
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: plain {t_plain*1e3:.1f} us ({fl/t_plain/1e9:.0f} TF/s)  whole-line {t_pk*1e3:.1f} us ({fl/t_pk/1e9:.0f} TF/s)  "
          f"bit-identical {torch.equal(o_plain, o_pk)} / {torch.equal(f_plain, f_pk)}  relerr {err:.2e}", flush=True)
    del bs, keep

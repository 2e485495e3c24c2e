#!/usr/bin/env python3
"""(To run: copy gemm_w4.hip into myriad_amd/csrc/, add "gemm_w4" to build.SOURCES, rebuild.)  Four-wave 256 x 256 GEMM instance (csrc/gemm_w4.hip) against the eight-wave kernel: bits and time, cold weights."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops, _lib
L = _lib.load()
L.mhdbg_gemm_w4.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
L.mhdbg_gemm_w4.restype = ctypes.c_int
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
SHAPES = [(1184, 22016, 4096), (1184, 11008, 4096), (2056, 6144, 1408), (4096, 4096, 4096), (8192, 8192, 8192), (300, 1000, 128), (256, 256, 64)]
for (M, N, K) in SHAPES:
    nb = max(2, min(8, int(1.3e9 // (N * K * 2)) + 1))
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dev).to(torch.bfloat16)
    bs = [(torch.randn(N, K, generator=g) * 0.05).to(dev).to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    out4 = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev)

    def w4(b, o=out4):
        rc = L.mhdbg_gemm_w4(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), o.data_ptr(), o.stride(0), M, N, K, 0, ops._s())
        assert rc == 0, rc

    ops.gemm(a, bs[0], out=out, variant=12)
    w4(bs[0])
    torch.cuda.synchronize()
    same = torch.equal(out, out4)
    ref = a.float() @ bs[0].float().t()
    err = ((out4.float() - ref).abs().max() / ref.abs().max()).item()
    res = {}
    for name, fn in (("8-wave", lambda b: ops.gemm(a, b, out=out, variant=12)), ("4-wave", w4)):
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(nb * 3):
                fn(bs[i % nb])
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (nb * 3))
        res[name] = best
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: 8-wave {res['8-wave']*1e3:.1f} us ({fl/res['8-wave']/1e9:.0f} TF/s)  4-wave {res['4-wave']*1e3:.1f} us "
          f"({fl/res['4-wave']/1e9:.0f} TF/s)  bit-identical {same}  relerr {err:.2e}", flush=True)
    del bs

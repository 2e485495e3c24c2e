#!/usr/bin/env python3
"""Four-wave 256 x 256 GEMM main loop (csrc/gemm_x4.hip, the default behind plan kernel 2) against the eight-wave kernel
(gemm_256.hip, MYRIAD_GEMM256_IMPL=0): same bits, and time on the step's shapes with cold weights.  --quick: bits only."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os; _os.environ.setdefault("MYRIAD_HIP_DEBUG_LIB", "1")   # the mhdbg_* hooks live in libmyriad_hip_dbg.so
from myriad_amd import ops, _lib
L = _lib.load()
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
quick = "--quick" in sys.argv
variant = int(sys.argv[sys.argv.index("--variant") + 1]) if "--variant" in sys.argv else 0   # sweep builds only
L.mhdbg_set_gemm_x4_variant(variant)
# (M, N, K, out_f32, bias, residual)
SHAPES = [(256, 256, 64, 0, 0, 0), (256, 256, 128, 0, 0, 0), (256, 256, 192, 1, 0, 0), (300, 1000, 256, 0, 1, 0), (1184, 4160, 1024, 1, 1, 1),
          (1184, 12288, 4160, 0, 0, 0), (1184, 22016, 4096, 0, 0, 0), (1184, 11008, 4096, 0, 0, 0), (1184, 4096, 22016, 0, 0, 0),
          (1184, 4096, 11008, 1, 0, 1), (1184, 4160, 12288, 1, 0, 0), (1184, 4096, 4096, 0, 0, 0), (2056, 6144, 1408, 0, 1, 0),
          (2056, 4224, 1408, 0, 1, 0), (2056, 1408, 6144, 1, 1, 1), (4096, 4096, 4096, 0, 0, 0), (8192, 8192, 8192, 0, 0, 0)]
if quick:
    SHAPES = SHAPES[:6]
bad = 0
for (M, N, K, f32, hb, hr) in SHAPES:
    nb = max(2, min(8, int(1.3e9 // (N * K * 2)) + 1))
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dev).to(torch.bfloat16)
    bs = [(torch.randn(N, K, generator=g) * 0.05).to(dev).to(torch.bfloat16) for _ in range(nb)]
    bias = torch.randn(N, generator=g).to(dev) if hb else None
    res = torch.randn(M, N, generator=g).to(dev) if hr else None
    dt = torch.float32 if f32 else torch.bfloat16
    outs = []
    times = {}
    for impl in (0, 1):
        L.mh_set_option(b"gemm256_impl", impl)
        out = torch.full((M, N), 7.0, dtype=dt, device=dev)
        ops.gemm(a, bs[0], out=out, bias=bias, residual=res, variant=12)
        torch.cuda.synchronize()
        outs.append(out.clone())             # the timing loop below overwrites `out` with other weight matrices
        if quick:
            continue
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(nb * 3):
                ops.gemm(a, bs[i % nb], out=out, bias=bias, residual=res, variant=12)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (nb * 3))
        times[impl] = best
    same = torch.equal(outs[0], outs[1])
    ref = a.float() @ bs[0].float().t()
    if bias is not None: ref = ref + bias
    if res is not None: ref = ref + res
    err = ((outs[1].float() - ref).abs().max() / ref.abs().max()).item()
    bad += (not same) or not (err < 2e-2)
    fl = 2.0 * M * N * K
    t = "" if quick else (f"  8-wave {times[0]*1e3:.1f} us ({fl/times[0]/1e9:.0f} TF/s)  4-wave {times[1]*1e3:.1f} us ({fl/times[1]/1e9:.0f} TF/s)"
                          f"  splits {ops.gemm_plan(M, N, K)[1]}")
    print(f"M={M} N={N} K={K} f32={f32} bias={hb} res={hr}: bit-identical {same} relerr {err:.2e}{t}", flush=True)
    del bs
L.mh_set_option(b"gemm256_impl", 1)
print("FAIL" if bad else "OK")
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Small-grid GEMM shapes of the Q-Former / VE nets (latency-bound: < 256 workgroups): 128x128 kernel variants (ring depth,
tile width) x explicit K splits.  Times are back-to-back launches on fresh weight buffers (cold-ish operands)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops, _lib
L = _lib.load()
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
SHAPES = [(2056, 1408, 1408), (81, 768, 768), (81, 3072, 768), (648, 768, 768), (648, 768, 2304), (648, 2304, 768), (648, 3072, 768), (648, 768, 3072), (2056, 1536, 1408),
          (2056, 1408, 1536), (256, 768, 768), (256, 1408, 640), (648, 4096, 768), (392, 768, 1024),
          (257, 1408, 1408), (196, 2368, 1024), (784, 640, 256), (81, 768, 3072), (81, 768, 2304), (81, 2304, 768)]
VAR_SHIFT = None


def bench(fn, nb):
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(nb):
            fn(i)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / nb)
    return best * 1e3


for (M, N, K) in SHAPES:
    nb = 16
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    bs = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    outf = torch.empty(M, N, dtype=torch.float32, device=dev)
    k, s = ops.gemm_plan(M, N, K)
    line = f"M={M} N={N} K={K} kt={K//64}: auto(k{k},s{s}) {bench(lambda i: ops.gemm(a, bs[i], out=out), nb):.1f} us"
    for v in (1, 3, 11, 13):
        line += f" | v{v} {bench(lambda i: ops.gemm(a, bs[i], out=out, variant=v), nb):.1f}"
    print(line, flush=True)
    line = "      split-K (128^2 default kernel + reduce):"
    for sp in (2, 3, 4, 6, 8):
        if K // 64 // sp < 2:
            continue
        ws = torch.empty(L.mh_gemm_splitk_ws_floats(M, N, sp), dtype=torch.float32, device=dev)
        t = bench(lambda i: L.mh_gemm_bf16_nt_splitk(a.data_ptr(), a.stride(0), bs[i].data_ptr(), bs[i].stride(0), outf.data_ptr(), N,
                                                     M, N, K, sp, ws.data_ptr(), ops._s()), nb)
        line += f" s={sp} {t:.1f}"
    print(line, flush=True)

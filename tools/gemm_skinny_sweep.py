#!/usr/bin/env python3
"""Split-K sweep of the 128x128 kernel on the batch-1 step's skinny GEMMs (explicit-split entry, f32 out, cold weights)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops, _lib
L = _lib.load()
dev = torch.device("cuda:0")
SHAPES = [(148, 12352, 4096), (148, 4096, 4096), (148, 22016, 4096), (148, 4096, 11008), (148, 4096, 22016), (257, 4224, 1408),
          (257, 1408, 1408), (257, 6144, 1408), (257, 1408, 6144)]
for (M, N, K) in SHAPES:
    nb = max(2, int(1.5e9 // (N * K * 2)) + 1)
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    bs = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty(M, N, dtype=torch.float32, device=dev)
    line = f"M={M} N={N} K={K}:"
    for s in (1, 2, 4, 8, 16):
        if K // 64 // s < 4:
            continue
        ws = torch.empty(L.mh_gemm_splitk_ws_floats(M, N, s), dtype=torch.float32, device=dev)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for b in bs:
                L.mh_gemm_bf16_nt_splitk(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), N, M, N, K, s, ws.data_ptr(), ops._s())
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / nb)
        line += f"  s={s}: {best*1e3:.0f} us"
    print(line, flush=True)
    del bs

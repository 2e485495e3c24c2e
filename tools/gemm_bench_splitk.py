#!/usr/bin/env python3
"""Cold-weight A/B: plain 2-stage 128x128 GEMM vs split-K (2/3/4 slabs + fixed-order fp32 reduce)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops, _lib

SHAPES = [(1184, 4096, 22016), (1184, 4096, 12288), (1184, 4096, 11008), (1184, 4096, 4096), (2056, 1408, 6144),
          (2056, 1408, 1408), (2056, 4224, 1408), (648, 768, 3072), (648, 768, 768)]
L = _lib.load()


def splitk(a, b, out, splits, ws):
    M, K = a.shape
    N = b.shape[0]
    rc = L.mh_gemm_bf16_nt_splitk(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), M, N, K,
                                  splits, ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0


print("| M | N | K | plain | split 2 | split 3 | split 4 |")
print("|---|---|---|---|---|---|---|")
for (M, N, K) in SHAPES:
    nb = max(2, int(1.3e9 // (N * K * 2)) + 1)
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    bs = [torch.randn(N, K, device="cuda").to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    ws = torch.empty(4 * M * N, dtype=torch.float32, device="cuda")
    res = {}
    for mode in (0, 2, 3, 4):
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for b in bs:
                if mode == 0:
                    ops.gemm(a, b, out=out, variant=1)
                else:
                    splitk(a, b, out, mode, ws)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / nb)
        res[mode] = best
    fl = 2.0 * M * N * K
    print(f"| {M} | {N} | {K} | " + " | ".join(f"{fl / (res[m] * 1e-3) / 1e12:.0f} TF ({res[m] * 1e3:.0f} us)" for m in (0, 2, 3, 4)) + " |")
    sys.stdout.flush()
    del bs

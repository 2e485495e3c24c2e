#!/usr/bin/env python3
"""Does a decode GEMV run faster when its weights were read just before (Infinity Cache, 256 MiB memory-side)?
cold = cycling > 1.5 GB of distinct matrices; warm = the same matrix again; touched = another kernel (a plain reduction) read the
matrix right before, the timed region holds the GEMV only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops
dev = torch.device("cuda:0")
for (M, N, K) in [(1, 4096, 4096), (1, 12288, 4096), (1, 4096, 11008), (1, 22016, 4096)]:
    nb = max(2, int(1.5e9 // (N * K * 2)) + 1)
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    bs = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.gemm(a, bs[0], out=out, out_dtype=torch.float32)
    def timed(fn, pre=None, n=nb):
        best = 1e9
        for _ in range(5):
            tot = 0.0
            evs = []
            for i in range(n):
                if pre is not None:
                    pre(i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(i); e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            tot = sum(e0.elapsed_time(e1) for e0, e1 in evs) / n
            best = min(best, tot)
        return best * 1e3
    cold = timed(lambda i: ops.gemm(a, bs[i], out=out, out_dtype=torch.float32))
    warm = timed(lambda i: ops.gemm(a, bs[0], out=out, out_dtype=torch.float32))
    touched = timed(lambda i: ops.gemm(a, bs[i], out=out, out_dtype=torch.float32), pre=lambda i: bs[i].view(torch.int16).max())
    mb = N * K * 2 / 1e6
    print(f"N={N} K={K} ({mb:.0f} MB): cold {cold:.1f} us ({mb/cold/1e0*1e-6*1e6/1e6:.2f} TB/s)  warm {warm:.1f} us  touched-before {touched:.1f} us", flush=True)
    del bs

#!/usr/bin/env python3
"""(Needs the MYRIAD_G2_KO knock-out instantiations, which were an experiment-only edit of gemm_256.hip: template <bool TRACE, int KO>, bits 1 no LDS-DMA in the loop, 2 no fragment reads, 4 no barrier, 8 no MFMA, 16 whole-line address pattern.)  Knock-out timing of gemm_256_kernel's main loop (MYRIAD_G2_KO, WRONG RESULTS by construction): 8192^3 and 1184x22016x4096."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops
dev = torch.device("cuda:0")
for (M, N, K) in [(8192, 8192, 8192), (1184, 22016, 4096)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    bs = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(3)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(6):
            ops.gemm(a, bs[i % 3], out=out, variant=12)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 6)
    print(f"KO={os.environ.get('MYRIAD_G2_KO', '0')} {M}x{N}x{K}: {best*1e3:.1f} us  {2*M*N*K/best/1e9:.0f} TF/s", flush=True)

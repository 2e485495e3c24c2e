#!/usr/bin/env python3
"""Phase timeline inside gemm_256_kernel's main loop (TRACE build: workgroup 0, wave 0 of each group stamps s_memtime at the phase
boundaries of every k-tile interval): cycles per phase, averaged over intervals 8..56.  python tools/gemm_phase_trace.py [M N K]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os; _os.environ.setdefault("MYRIAD_HIP_DEBUG_LIB", "1")   # the mhdbg_* hooks live in libmyriad_hip_dbg.so
from myriad_amd import ops, _lib
dev = torch.device("cuda:0")
cdll = ctypes.CDLL(_lib.LIB_PATH)
cdll.mhdbg_set_gemm256_trace.argtypes = [ctypes.c_void_p]
M, N, K = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1184, 12288, 4096))]
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
bs = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(6)]
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
nwg = ((M + 255) // 256) * ((N + 255) // 256)
trace = torch.zeros(1024 + 4 * nwg + 16, dtype=torch.int64, device=dev)
for b in bs[:3]:
    ops.gemm(a, b, out=out, variant=12)
torch.cuda.synchronize()
cdll.mhdbg_set_gemm256_trace(ctypes.c_void_p(trace.data_ptr()))
ops.gemm(a, bs[3], out=out, variant=12)
ops.gemm(a, bs[4], out=out, variant=12)
torch.cuda.synchronize()
cdll.mhdbg_set_gemm256_trace(ctypes.c_void_p(0))
t = trace[:1024].view(2, 64, 8).cpu()
names = ["B: MFMA(t-1) | A: -", "fragment reads + LDS-DMA issue", "A: MFMA(t) | B: -", "counted vmcnt + lgkmcnt waits", "barrier"]
for grp, gname in ((0, "group A (reads, issue, MFMA)"), (1, "group B (MFMA, reads, issue)")):
    rows = t[grp, 8:56]
    d = (rows[:, 1:6] - rows[:, 0:5]).double()
    per = (rows[1:, 0] - rows[:-1, 0]).double()
    clk = (rows[-1, 0] - rows[0, 0]).item() / max(1, (rows[-1, 7] - rows[0, 7]).item()) * 100.0   # shader cycles per us
    print(f"{gname}: interval {per.mean():.0f} cycles = {per.mean() / clk:.3f} us (shader clock {clk / 1e3:.2f} GHz): " +
          " | ".join(f"{names[i]} {d[:, i].mean():.0f}" for i in range(5)))

#!/usr/bin/env python3
"""Where a 256x256 GEMM launch spends its fixed cost: per-workgroup life-cycle stamps (gemm_256.hip TRACE build,
100 MHz counter) -> dispatch ramp, first-tile latency, main loop, epilogue, finish spread.
python tools/gemm_life.py [M N K]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os; _os.environ.setdefault("MYRIAD_HIP_DEBUG_LIB", "1")   # the mhdbg_* hooks live in libmyriad_hip_dbg.so
from myriad_amd import ops, _lib

dev = torch.device("cuda:0")
cdll = ctypes.CDLL(_lib.LIB_PATH)
M, N, K = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1184, 12288, 4096))]
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
bs = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(12)]
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
nwg = ((M + 255) // 256) * ((N + 255) // 256)
trace = torch.zeros(1024 + 4 * nwg + 16, dtype=torch.int64, device=dev)
cdll.mhdbg_set_gemm256_trace.argtypes = [ctypes.c_void_p]
for b in bs[:4]:
    ops.gemm(a, b, out=out, variant=12)
torch.cuda.synchronize()
cdll.mhdbg_set_gemm256_trace(ctypes.c_void_p(trace.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ops.gemm(a, bs[4], out=out, variant=12)       # predecessor (its stamps are overwritten), so the traced launch is back-to-back
e0.record()
ops.gemm(a, bs[5], out=out, variant=12)
e1.record()
torch.cuda.synchronize()
cdll.mhdbg_set_gemm256_trace(ctypes.c_void_p(0))
life = trace[1024:1024 + 4 * nwg].view(nwg, 4).cpu().double() * 0.01      # us
t0 = life[:, 0].min()
life -= t0
ent, first, loop_end, end = life[:, 0], life[:, 1], life[:, 2], life[:, 3]
print(f"{M}x{N}x{K}: {nwg} workgroups, launch {e0.elapsed_time(e1) * 1e3:.1f} us by events (TRACE build)")
print(f"  entry            : first 0.0, median {ent.median():.1f}, last {ent.max():.1f} us after the first workgroup started")
print(f"  first tile landed: median {(first - ent).median():.1f} us after entry (max {(first - ent).max():.1f})")
print(f"  main loop        : median {(loop_end - first).median():.1f} us (min {(loop_end - first).min():.1f}, max {(loop_end - first).max():.1f})")
print(f"  epilogue         : median {(end - loop_end).median():.1f} us (max {(end - loop_end).max():.1f})")
print(f"  finish           : first {end.min():.1f}, median {end.median():.1f}, last {end.max():.1f} us")

#!/usr/bin/env python3
"""Phase-level timeline of the ping-pong GEMM (gemm_pp.hip TRACE build): s_memtime stamps of one wave of each group."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops, _lib

dev = torch.device("cuda:0")
lib = _lib.load() if hasattr(_lib, "load") else None
cdll = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "libmyriad_hip.so"))
M, N, K = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 4096, 4096))]
SCHED = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cdll.mhdbg_set_gemm_sched(SCHED)
VAR = int(sys.argv[5]) if len(sys.argv) > 5 else 12
hook = cdll.mhdbg_set_gemm256_trace if VAR == 12 else cdll.mhdbg_set_gemm_trace
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
b = torch.randn(N, K, device=dev).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    ops.gemm(a, b, out=out, variant=VAR)
trace = torch.zeros(2 * 64 * 8, dtype=torch.int64, device=dev)
hook.argtypes = [ctypes.c_void_p]
hook(ctypes.c_void_p(trace.data_ptr()))
ops.gemm(a, b, out=out, variant=VAR)
torch.cuda.synchronize()
hook(ctypes.c_void_p(0))
tr = trace.cpu().view(2, 64, 8)
t0 = int(tr[0, 0, 0])
names = ["R.start", "R.issued", "R.waited", "bar1.out", "M.issued", "M.waited", "bar2.out"]
print("stamps relative to group A's first READ, s_memtime ticks")
for t in list(range(0, 6)) + list(range(30, 34)):
    for g in range(2):
        row = [int(tr[g, t, s]) - t0 for s in range(7)]
        d = [row[i + 1] - row[i] for i in range(6)]
        if VAR == 12 and SCHED == 1:
            print(f"t={t:2d} grp={'AB'[g]} start={row[0]:7d}  mfmaB={d[0]:5d} reads+dma={d[1]:5d} mfmaA={d[2]:5d} waits={d[3]:5d} barrier={d[4]:5d}")
        else:
            print(f"t={t:2d} grp={'AB'[g]} start={row[0]:7d}  reads={d[0]:5d} wait={d[1]:5d} bar1={d[2]:5d} mfma={d[3]:5d} vmwait={d[4]:5d} bar2={d[5]:5d}")
per = (int(tr[0, 40, 0]) - int(tr[0, 8, 0])) / 32.0
print(f"period per k-tile (A, t=8..40): {per:.0f} ticks")

if VAR == 12 and SCHED == 1:
    ticks = int(tr[0, 60, 0]) - int(tr[0, 4, 0])
    rt = int(tr[0, 60, 7]) - int(tr[0, 4, 7])
    print(f"s_memtime ticks {ticks} over {rt} ticks of the 100 MHz counter -> s_memtime runs at {ticks / (rt / 100e6) / 1e9:.3f} GHz")

#!/usr/bin/env python3
"""MiniGPT-4 arch shapes (M = 648 = 8 x 81 rows: 2.53 row tiles of 256): plan kernel 2 with forced K splits 1..8, GEMM + its slab sum
(ops.gemm, f32 out), cold weights.  Debug library (mhdbg_set_big_splits)."""
import sys, os
os.environ["MYRIAD_HIP_DEBUG_LIB"] = "1"
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops, _lib
L = _lib.load()
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 648
SHAPES = [(M, 4096, 4096), (M, 4096, 11008), (M, 4096, 22016), (M, 12288, 4096), (M, 11008, 4096), (M, 22016, 4096)]


def bench(fn, nb, reps=4):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(nb):
            fn(i)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / nb)
    return best * 1e3


for (m, N, K) in SHAPES:
    nb = 6
    a = torch.randn(m, K, device=dev).to(torch.bfloat16)
    bs = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(nb)]
    out = torch.empty(m, N, dtype=torch.float32, device=dev)
    outb = torch.empty(m, N, dtype=torch.bfloat16, device=dev)
    L.mhdbg_set_big_splits(0)
    k, s = ops.gemm_plan(m, N, K)
    line = f"M={m} N={N} K={K} auto(k{k},s{s}) f32 {bench(lambda i: ops.gemm(a, bs[i], out=out), nb):.1f} bf16 {bench(lambda i: ops.gemm(a, bs[i], out=outb), nb):.1f} us |"
    tiles = ((m + 255) // 256) * ((N + 255) // 256)
    for sp in (1, 2, 3, 4, 5, 6, 8):
        if K // 64 // sp < 8:
            continue
        L.mhdbg_set_big_splits(sp)
        line += f" s{sp}({tiles * sp}wg) {bench(lambda i: ops.gemm(a, bs[i], out=out, variant=12), nb):.1f}"
    L.mhdbg_set_big_splits(0)
    print(line, flush=True)

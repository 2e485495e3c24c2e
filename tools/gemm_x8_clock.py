#!/usr/bin/env python3
"""Effective shader clock under the 256x256 GEMM loop: workgroups 0, 64, 128, ... stamp s_memtime (shader clock) and
s_memrealtime (100 MHz) at entry and exit (mhdbg_set_gemm_x4_clock_probe); cycles / ticks x 100 MHz is the clock the workgroup
really ran at.  One workgroup alone, a quarter / half / all of the chip, and the step's shapes."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os; _os.environ.setdefault("MYRIAD_HIP_DEBUG_LIB", "1")   # the mhdbg_* hooks live in libmyriad_hip_dbg.so
from myriad_amd import ops, _lib
L = _lib.load()
L.mhdbg_set_gemm_x4_clock_probe.argtypes = [ctypes.c_void_p]
L.mhdbg_set_gemm_x4_clock_probe.restype = None
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
probe = torch.zeros(64, dtype=torch.int64, device=dev)
for (M, N, K) in ((256, 256, 16384), (2048, 2048, 8192), (2048, 4096, 8192), (4096, 4096, 8192), (8192, 8192, 8192), (1184, 22016, 4096),
                  (1184, 12288, 4160), (2056, 6144, 1408)):
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    for _ in range(20):                                   # warm: the clock settles under sustained load
        ops.gemm(a, b, out=out, variant=12)
    torch.cuda.synchronize()
    probe.zero_()
    L.mhdbg_set_gemm_x4_clock_probe(probe.data_ptr())
    for _ in range(10):
        ops.gemm(a, b, out=out, variant=12)
    torch.cuda.synchronize()
    L.mhdbg_set_gemm_x4_clock_probe(None)
    p = probe.cpu().view(-1, 4)[:, :2]
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    rows = [(int(c), int(t)) for c, t in p.tolist() if t > 0]
    ghz = [c / t * 0.1 for c, t in rows]
    us = [t / 100 for _, t in rows]
    print(f"{M}x{N}x{K}: {tiles} tiles; per probed workgroup {min(us):.1f}-{max(us):.1f} us; shader clock {min(ghz):.3f}-{max(ghz):.3f} GHz "
          f"(mean {sum(ghz) / len(ghz):.3f})", flush=True)

# in-workgroup time vs launch-to-launch time: what a launch costs OUTSIDE its workgroups (dispatch, end-of-kernel L2 write-back
# across the eight XCDs, next launch)
print("\n| shape | K | launch-to-launch us | probed workgroups us (min-max) | outside the workgroups us | GHz |\n|---|---|---|---|---|---|")
for (M, N, NOST) in ((4096, 4096, 0), (4096, 4096, 1), (1184, 12288, 0), (2056, 6144, 0)):
    L.mhdbg_set_gemm_x4_no_stores(NOST)
    for K in (64, 256, 1024, 4096):
        a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        for _ in range(20):
            ops.gemm(a, b, out=out, variant=12)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100):
            ops.gemm(a, b, out=out, variant=12)
        e1.record(); torch.cuda.synchronize()
        l2l = e0.elapsed_time(e1) * 10
        probe.zero_()
        L.mhdbg_set_gemm_x4_clock_probe(probe.data_ptr())
        for _ in range(5):
            ops.gemm(a, b, out=out, variant=12)
        torch.cuda.synchronize()
        L.mhdbg_set_gemm_x4_clock_probe(None)
        rows = [(int(c), int(t), int(h), int(e)) for c, t, h, e in probe.cpu().view(-1, 4).tolist() if t > 0]
        us = [t / 100 for _, t, _, _ in rows]
        ghz = sum(c / t * 0.1 for c, t, _, _ in rows) / len(rows)
        head = sum(h for _, _, h, _ in rows) / len(rows) / 100
        loop = sum(e - h for _, _, h, e in rows) / len(rows) / 100
        tail = sum(t - e for _, t, _, e in rows) / len(rows) / 100
        print(f"| {M}x{N}{' NO STORES' if NOST else ''} | {K} | {l2l:.1f} | {min(us):.1f}-{max(us):.1f} (setup {head:.2f} + loop {loop:.1f} + read-out {tail:.1f}) | {l2l - max(us):.1f} | {ghz:.2f} |", flush=True)
L.mhdbg_set_gemm_x4_no_stores(0)

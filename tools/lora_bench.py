#!/usr/bin/env python3
"""LoRA skinny kernels in isolation at the step's shape (M = 1184, D = 4096, r = 8), with and without dropout."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import _lib, ops
L = _lib.load()
dev = "cuda:0"
M, D, r, W = 1184, 4096, 8, 4096
R2 = 2 * r
x_ext = torch.randn(M, D + 64, device=dev).to(torch.bfloat16)
A = torch.randn(R2, D, device=dev) * 0.02
dx_ext = torch.randn(M, D + 64, device=dev)
dqkv = torch.randn(M, 3 * W, device=dev).to(torch.bfloat16)
gA = torch.zeros(R2, D, device=dev); gBq = torch.zeros(W, r, device=dev); gBv = torch.zeros(W, r, device=dev)
ws = torch.empty(L.mh_lora_wgrad_ws_floats(D, R2), device=dev)
out = torch.empty(M, D, device=dev)
s = ops._s()


def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for p in (0.0, 0.05):
    down = t(lambda: L.mh_lora_down(x_ext.data_ptr(), x_ext.stride(0), A.data_ptr(), x_ext[:, D:].data_ptr(), x_ext.stride(0), M, D, R2, 2.0, p, 123, s))
    dx = t(lambda: L.mh_lora_dx(dx_ext.data_ptr(), dx_ext.stride(0), A.data_ptr(), out.data_ptr(), M, D, R2, 2.0, p, 123, s))
    wg = t(lambda: L.mh_lora_wgrad(x_ext.data_ptr(), x_ext.stride(0), dx_ext.data_ptr(), dx_ext.stride(0), dqkv.data_ptr(), dqkv[:, 2 * W:].data_ptr(),
                                   dqkv.stride(0), x_ext[:, D:].data_ptr(), x_ext.stride(0), gA.data_ptr(), gBq.data_ptr(), gBv.data_ptr(), ws.data_ptr(),
                                   M, D, R2, 2.0, p, 123, s))
    print(f"p={p}: lora_down {down:.1f} us, lora_dx {dx:.1f} us, lora_wgrad (partial + reduce) {wg:.1f} us")

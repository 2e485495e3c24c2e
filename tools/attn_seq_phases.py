#!/usr/bin/env python3
"""Phase timeline of attn_seq_bwd_kernel (workgroup 0, waves 0 and 7; 100 MHz counter) at the step's shape (H = 32, S = 148),
fed by the o_proj dgrad's split-K slabs as in the step: python tools/attn_seq_phases.py [B]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os; _os.environ.setdefault("MYRIAD_HIP_DEBUG_LIB", "1")   # the mhdbg_* hooks live in libmyriad_hip_dbg.so
from myriad_amd import ops, _lib
L = _lib.load()
L.mhdbg_set_attn_seq_trace.argtypes = [ctypes.c_void_p]
L.mhdbg_set_attn_seq_trace.restype = None
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, D, S = 32, 128, 148
W = H * D
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B, S, 3 * W, generator=g) * 0.5).to(dev).to(torch.bfloat16)
pos = torch.arange(S, dtype=torch.int32).repeat(B).to(dev)
inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
ang = torch.arange(2048).float()[:, None] * inv[None]
cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
scale = D ** -0.5
o, lse = ops.attn_rope_fwd(qkv, H, D, scale, pos, cos, sin)
a = (torch.randn(B * S, 4096, generator=g) * 0.1).to(dev).to(torch.bfloat16)
bw = (torch.randn(W, 4096, generator=g) * 0.02).to(dev).to(torch.bfloat16)
trace = torch.zeros(64, dtype=torch.int64, device=dev)
names = ["stage K,V (global -> rope -> LDS)", "lse + own rows: Q, dO slab sums, O, delta", "barrier", "phase A (dq)", "swap images", "phase B (dk, dv)"]
for it in range(3):
    ops.gemm_attn_rope_bwd(a, bw, qkv, o, lse, H, D, scale, pos, cos, sin)
torch.cuda.synchronize()
L.mhdbg_set_attn_seq_trace(trace.data_ptr())
ops.gemm_attn_rope_bwd(a, bw, qkv, o, lse, H, D, scale, pos, cos, sin)
torch.cuda.synchronize()
L.mhdbg_set_attn_seq_trace(None)
t = trace.cpu().tolist()
k, s = ops.gemm_plan(B * S, W, 4096)
print(f"B={B}: o_proj dgrad plan kernel {k} splits {s}")
for w, base in (("wave 0", 0), ("wave 7", 16)):
    st = t[base:base + 7]
    print(w, " | ".join(f"{names[i]} {(st[i + 1] - st[i]) / 100:.1f} us" for i in range(6) if st[i + 1] and st[i]), f"| total {(max(st) - st[0]) / 100:.1f} us")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
do = ops.gemm(a, bw)
e0.record()
for _ in range(20):
    ops.attn_rope_bwd(qkv, o, do, lse, H, D, scale, pos, cos, sin)
e1.record(); torch.cuda.synchronize()
print(f"attn_rope_bwd alone (bf16 dO): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")
e0.record()
for _ in range(20):
    ops.attn_rope_fwd(qkv, H, D, scale, pos, cos, sin)
e1.record(); torch.cuda.synchronize()
print(f"attn_rope_fwd: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")

#!/usr/bin/env python3
"""LLaMA attention at the step's shape (B=8, H=32, S=148, d=128): tile kernels + separate rope launches vs the
whole-sequence fused kernels (attn_seq.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops

dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
B, H, S, D = int(os.environ.get("B", 8)), 32, int(os.environ.get("S", 148)), 128
W = H * D
ld = 3 * W + 64
torch.manual_seed(0)
qkvs = [(torch.randn(B, S, ld, device=dev) * 0.5).to(torch.bfloat16) for _ in range(8)]
inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
fr = torch.einsum("i,j->ij", torch.arange(2048).float(), inv)
cos, sin = fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev)
pos = torch.arange(S, dtype=torch.int32).repeat(B).to(dev)
scale = D ** -0.5
dout = (torch.randn(B, S, W, device=dev) * 0.1).to(torch.bfloat16)


def timeit(fn, n=40):
    for _ in range(3):
        fn(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def old_fwd(i):
    q = qkvs[i % 8]
    ops.rope_(q.view(B * S, ld), 0, 2 * H, D, pos, cos, sin, 1.0)
    return ops.attn_fwd(q[:, :, :W], q[:, :, W:2 * W], q[:, :, 2 * W:3 * W], H, D, scale, causal=True)


def new_fwd(i):
    return ops.attn_rope_fwd(qkvs[i % 8], H, D, scale, pos, cos, sin)


o_old, lse_old = old_fwd(0)
o_new, lse_new = new_fwd(1)
dq_buf = torch.empty_like(qkvs[0])


def old_bwd(i):
    q = qkvs[i % 8]
    d3 = dq_buf
    ops.attn_bwd(q[:, :, :W], q[:, :, W:2 * W], q[:, :, 2 * W:3 * W], o_old, dout, lse_old, H, D, scale, causal=True,
                 dq=d3[:, :, :W], dk=d3[:, :, W:2 * W], dv=d3[:, :, 2 * W:3 * W])
    ops.rope_(d3.view(B * S, ld), 0, 2 * H, D, pos, cos, sin, -1.0)


def new_bwd(i):
    ops.attn_rope_bwd(qkvs[i % 8], o_new, dout, lse_new, H, D, scale, pos, cos, sin, dqkv=dq_buf)


print(f"B={B} H={H} S={S} d={D}")
print(f"forward : rope + tile kernel {timeit(old_fwd):7.1f} us   fused whole-sequence {timeit(new_fwd):7.1f} us")
print(f"backward: tile kernels + rope {timeit(old_bwd):7.1f} us   fused whole-sequence {timeit(new_bwd):7.1f} us")

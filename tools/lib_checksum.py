#!/usr/bin/env python3
"""Exact checksums (int64 sum of the raw output bits) of a battery of library ops on seeded inputs: run it under two builds of
libmyriad_hip.so and diff the output to see which kernels changed a single bit.  Usage: python tools/lib_checksum.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops
dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
BF, F32 = torch.bfloat16, torch.float32


def bits(t):
    t = t.contiguous()
    return int((t.view(torch.int16) if t.dtype == BF else t.view(torch.int32)).to(torch.int64).sum())


def rnd(*shape, seed, scale=1.0, dtype=F32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev)


def show(name, *ts):
    print(f"{name:44s} " + " ".join(str(bits(t)) for t in ts if t is not None), flush=True)


M, D, I = 1184, 4096, 11008
x32 = rnd(M, D, seed=1); w = rnd(D, seed=2, scale=0.1) + 1.0
show("rmsnorm_fwd", ops.rmsnorm_fwd(x32, w, 1e-6))
dy = rnd(M, D, seed=3, scale=0.1)
show("rmsnorm_bwd", *ops.rmsnorm_bwd(dy, x32, w, 1e-6, want_f32=True, want_bf16=True))
b = rnd(D, seed=4, scale=0.1)
show("layernorm_fwd", *ops.layernorm_fwd(x32[:, :1408].contiguous(), w[:1408].contiguous(), b[:1408].contiguous(), 1e-6, want_bf16=True, want_f32=True))
gu = rnd(M, 2 * I, seed=5, scale=0.7, dtype=BF)
act = ops.silu_mul_fwd_blk(gu)
show("silu_mul_fwd_blk", act)
dh = rnd(M, I, seed=6, scale=0.1, dtype=BF)
show("silu_mul_bwd_blk", ops.silu_mul_bwd_blk(dh, gu))
pre = rnd(648, 3072, seed=7, dtype=BF)
show("gelu_fwd / bwd", ops.gelu_fwd(pre), ops.gelu_bwd(rnd(648, 3072, seed=8, scale=0.1, dtype=BF), pre))
show("to_bf16", ops.to_bf16(x32))
a = rnd(M, 4096, seed=9, scale=0.5, dtype=BF)
for (N, K, kw, name) in ((12288, 4096, {}, "plain"), (4096, 4096, {}, "split"), (6144, 1408, dict(bias=True, gelu=True), "bias+gelu"),
                         (4224, 1408, dict(bias=True), "bias"), (4096, 4096, dict(res=True, f32=True), "res f32"),
                         (768, 768, dict(bias=True), "small bias"), (3072, 768, dict(bias=True, gelu=True), "small gelu")):
    aa = a[:, :K].contiguous() if K <= 4096 else a
    mrows = 2056 if K == 1408 else (648 if K == 768 else M)
    aa = rnd(mrows, K, seed=10 + N % 97, scale=0.5, dtype=BF)
    bw = rnd(N, K, seed=11 + N % 89, scale=0.05, dtype=BF)
    bias = rnd(N, seed=12, scale=0.3) if kw.get("bias") else None
    res = rnd(mrows, N, seed=13) if kw.get("res") else None
    out = ops.gemm(aa, bw, bias=bias, residual=res, gelu=bool(kw.get("gelu")), out_dtype=F32 if kw.get("f32") else BF)
    show(f"gemm {mrows}x{N}x{K} {name} plan{ops.gemm_plan(mrows, N, K)}", out)
H, hd, S, B = 32, 128, 148, 2
qkv = rnd(B, S, 3 * H * hd, seed=20, scale=0.5, dtype=BF)
inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
fr = torch.einsum("i,j->ij", torch.arange(256).float(), inv)
cos, sin = fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev)
pos = torch.arange(S, dtype=torch.int32).repeat(B).to(dev)
o, lse = ops.attn_rope_fwd(qkv, H, hd, hd ** -0.5, pos, cos, sin)
show("attn_rope_fwd", o, lse)
do = rnd(B, S, H * hd, seed=21, scale=0.1, dtype=BF)
show("attn_rope_bwd", ops.attn_rope_bwd(qkv, o, do, lse, H, hd, hd ** -0.5, pos, cos, sin))
q = rnd(4, 257, 16 * 88, seed=22, scale=0.5, dtype=BF); kv = rnd(4, 257, 2 * 16 * 88, seed=23, scale=0.5, dtype=BF)
o2, lse2 = ops.attn_fwd(q, kv[:, :, :1408], kv[:, :, 1408:], 16, 88, 88 ** -0.5)
show("attn_fwd d=88", o2, lse2)
show("attn_bwd d=88", *ops.attn_bwd(q, kv[:, :, :1408], kv[:, :, 1408:], o2, rnd(4, 257, 1408, seed=24, scale=0.1, dtype=BF), lse2, 16, 88, 88 ** -0.5))
logits = rnd(128, 32000, seed=30, scale=2.0)
labels = torch.randint(0, 32000, (128,), generator=torch.Generator().manual_seed(31)).to(dev)
r = ops.clamp_ce(logits, labels, 1.0)
show("clamp_ce", *[t for t in (r if isinstance(r, (tuple, list)) else (r,)) if isinstance(t, torch.Tensor)])

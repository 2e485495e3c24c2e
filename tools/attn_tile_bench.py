#!/usr/bin/env python3
"""Tile attention kernels (attention.hip) at the step's non-LLaMA shapes: EVA-ViT self-attention (B=8, H=16, S=257, d=88, qkv
packed [B,S,3*1408]), Q-Former self-attention (H=12, S=81, d=64) and cross-attention (Sq=81, Sk=257).  Forward and backward,
HIP-event time per call and an exact checksum of the outputs (int64 sum of the raw bf16 bits) so that two builds can be
compared bit for bit.  Usage: python tools/attn_tile_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops

dev = torch.device("cuda:0")
ops.ensure_workspace(dev)
torch.manual_seed(0)


def bits(t):
    return int(t.contiguous().view(torch.int16).to(torch.int64).sum()) if t.dtype == torch.bfloat16 else int(
        t.contiguous().view(torch.int32).to(torch.int64).sum())


def timeit(fn, n=50):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, B, H, Sq, Sk, D in (("vit", 8, 16, 257, 257, 88), ("qf-self", 8, 12, 81, 81, 64), ("qf-cross", 8, 12, 81, 257, 64),
                              ("vit-b1", 1, 16, 257, 257, 88)):
    W = H * D
    q = (torch.randn(B, Sq, W, device=dev) * 0.5).to(torch.bfloat16)
    kv = (torch.randn(B, Sk, 2 * W, device=dev) * 0.5).to(torch.bfloat16)
    k, v = kv[:, :, :W], kv[:, :, W:]
    dout = (torch.randn(B, Sq, W, device=dev) * 0.1).to(torch.bfloat16)
    scale = D ** -0.5
    o, lse = ops.attn_fwd(q, k, v, H, D, scale)
    t_f = timeit(lambda: ops.attn_fwd(q, k, v, H, D, scale))
    dq, dk, dv = ops.attn_bwd(q, k, v, o, dout, lse, H, D, scale)
    t_b = timeit(lambda: ops.attn_bwd(q, k, v, o, dout, lse, H, D, scale))
    fl = 4.0 * B * H * Sq * Sk * D
    print(f"{name:9s} B={B} H={H} Sq={Sq} Sk={Sk} d={D}: fwd {t_f:6.1f} us ({fl / t_f / 1e6:6.1f} TF/s)  bwd {t_b:6.1f} us   "
          f"bits o {bits(o)} lse {bits(lse)} dq {bits(dq)} dk {bits(dk)} dv {bits(dv)}", flush=True)

#!/usr/bin/env python3
"""transpose_to_bf16 at the VE-net shapes: achieved bandwidth."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import ops
dev = torch.device("cuda:0")
for (R, C, dt) in [(4096, 25664, torch.bfloat16), (72, 25664, torch.bfloat16), (72, 4096, torch.bfloat16), (401408, 64, torch.bfloat16),
                   (401408, 4, torch.bfloat16), (100352, 64, torch.bfloat16), (1184, 4096, torch.bfloat16), (648, 768, torch.bfloat16)]:
    x = torch.randn(R, C, device=dev).to(dt)
    y = ops.transpose_to_bf16(x, 64)
    ref = x.float().t().to(torch.bfloat16)
    assert torch.equal(y[:, :R], ref), (R, C)
    assert float(y[:, R:].float().abs().sum()) == 0.0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.transpose_to_bf16(x, 64)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    mb = (x.numel() * x.element_size() + y.numel() * 2) / 1e6
    print(f"[{R} x {C}] {us:.1f} us  {mb / us:.2f} TB/s ({mb:.0f} MB)")

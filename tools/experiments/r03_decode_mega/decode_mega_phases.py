#!/usr/bin/env python3
"""Phase timeline of the persistent decode kernel (mh_decode_mega): workgroup 0 stamps the 100 MHz counter at every phase
boundary of layer 1; printed in us.  Usage: MYRIAD_DECODE_MEGA=1 python tools/decode_mega_phases.py"""
import os, sys
os.environ.setdefault("MYRIAD_DECODE_MEGA", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd import _lib
from myriad_amd.llama import LlamaHIP
from tests import golden_utils as gu

dev = "cuda:0"
L = _lib.load()
D, layers, heads, inter, V = 4096, 4, 32, 11008, 32000
lm = LlamaHIP(gu.llama_weights(D, layers, inter, V, seed=1, std=0.03), heads, dev, need_backward=False)
emb = (torch.randn(1, 120, D) * 0.05).to(dev)
import ctypes
L.mhdbg_set_mega_trace.argtypes = [ctypes.c_void_p]
L.mhdbg_set_mega_trace.restype = None
trace = torch.zeros(32, dtype=torch.int64, device=dev)
lm.greedy_generate(emb, max_new_tokens=6, stop_ids=(), use_graph=False)           # warm
L.mhdbg_set_mega_trace(trace.data_ptr())
lm.greedy_generate(emb, max_new_tokens=6, stop_ids=(), use_graph=False)
torch.cuda.synchronize()
L.mhdbg_set_mega_trace(None)
t = trace.cpu().tolist()
names = ["P1 rmsnorm rows", "P1 qkv gemv", "barrier 1 (+wo prefetch)", "P2 attention", "barrier 2", "P3 copy + wo gemv", "barrier 3 (+wgu prefetch)",
         "P4 rmsnorm rows", "P4 gate|up gemv", "barrier 4 (+wd prefetch)", "P5 silu rows", "P5 down gemv", "barrier 5 (+next qkv prefetch)"]
for i, n in enumerate(names):
    print(f"{n:36s} {(t[i + 1] - t[i]) / 100.0:8.2f} us")
print(f"{'layer total':36s} {(t[13] - t[0]) / 100.0:8.2f} us   (ideal weight stream: 402 MB / 6.3 TB/s = 64 us)")

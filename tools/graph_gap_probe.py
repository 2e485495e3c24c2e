#!/usr/bin/env python3
"""Does hipGraph replay shorten the GPU-side gaps of a chain of small dependent launches?  Q-Former forward (full width, 12
layers, 81 queries x 257 image tokens) at batch 1 and 8: eager stream launches vs one captured graph, GPU time by events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd.myriad import MyriadHIP
from myriad_amd.synthetic import SyntheticWeights, full_config
dev = "cuda:0"
cfg = full_config(llm_layers=1, vit_depth=1)
m = MyriadHIP(SyntheticWeights(cfg, dev, seed=0), dict(need_backward=True), device=dev)
qf = m.qformer
import os as _os
for B in ([int(_os.environ['PROBE_B'])] if _os.environ.get('PROBE_B') else (1, 8)):
    q = torch.randn(B, 81, qf.D, device=dev)
    enc = torch.randn(B, 257, 1408, device=dev).to(torch.bfloat16)
    def run():
        return qf.forward(q, enc, save_for_backward=False)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    def gpu_ms(fn, n=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    eager = gpu_ms(run)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
        with torch.cuda.graph(g, stream=s):
            out = run()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    replay = gpu_ms(g.replay)
    print(f"B={B}: Q-Former forward eager {eager:.3f} ms, graph replay {replay:.3f} ms", flush=True)
    dout = torch.randn(B, 81, qf.D, device=dev)
    def fb():
        qf.forward(q, enc, save_for_backward=True)
        return qf.backward(dout)
    for _ in range(3):
        fb()
    eager2 = gpu_ms(fb)
    g2 = torch.cuda.CUDAGraph()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fb()
        with torch.cuda.graph(g2, stream=s):
            o2 = fb()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    replay2 = gpu_ms(g2.replay)
    print(f"B={B}: Q-Former forward + backward eager {eager2:.3f} ms, graph replay {replay2:.3f} ms", flush=True)

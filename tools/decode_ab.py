#!/usr/bin/env python3
"""A/B of the decode loop on one box: the previous greedy_generate (graph captured per call, 4 copies per step) vs the current one."""
import os, sys, time, types, importlib.util
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd.myriad import MyriadHIP
from myriad_amd.synthetic import SyntheticWeights, full_config
spec = importlib.util.spec_from_file_location("myriad_amd.llama_prev", os.path.join(os.path.dirname(__file__), "experiments", "llama_r2_prev.py"))
prev = importlib.util.module_from_spec(spec); prev.__package__ = "myriad_amd"; spec.loader.exec_module(prev)
dev = "cuda:0"
model = MyriadHIP(SyntheticWeights(full_config(), dev, seed=0), dict(need_backward=False), device=dev)
model.eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = torch.Generator().manual_seed(1)
smp = dict(image=torch.randn(B, 3, 224, 224, generator=g), anomaly_maps=torch.rand(B, 1, 224, 224, generator=g),
           before_ids=torch.randint(3, 32000, (1, 4), generator=g).expand(B, -1).contiguous(),
           after_ids=torch.randint(3, 32000, (1, 28), generator=g).expand(B, -1).contiguous())
cur = model.llama.greedy_generate
old = types.MethodType(prev.LlamaHIP.greedy_generate, model.llama)
for name, fn in (("current", cur), ("previous", old), ("current", cur), ("previous", old)):
    model.llama.greedy_generate = fn
    ts = {}
    for n in (4, 8, 24, 96, 24, 96):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model.generate(smp, max_new_tokens=n, stop_ids=((-1,),), min_length=0, eos_token_id=-5)
        torch.cuda.synchronize(); ts[n] = time.perf_counter() - t0
    print(f"{name:9s} B={B}: 24 tokens {ts[24]*1e3:.1f} ms, 96 tokens {ts[96]*1e3:.1f} ms -> {(ts[96]-ts[24])/72*1e3:.3f} ms/token, fixed {(ts[24]-23*(ts[96]-ts[24])/72)*1e3:.1f} ms")

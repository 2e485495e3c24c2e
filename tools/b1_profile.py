#!/usr/bin/env python3
"""Host-side profile of one B=1 fine-tune step (launch-bound regime): where the Python time goes."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_samples
from myriad_amd.myriad import MyriadHIP
from myriad_amd.synthetic import SyntheticWeights, full_config
dev = "cuda:0"
cfg = full_config()
m = MyriadHIP(SyntheticWeights(cfg, dev, seed=0), dict(fixed_stage=1, fixed_taskstage=0, use_lora=True), device=dev)
m.train()
s1 = make_samples(1, cfg["vocab"], 42, dev)
for i in range(3):
    m.train_step(s1, 1e-4)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(5):
    m.train_step(s1, 1e-4)
t_launch = (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
t_total = (time.perf_counter() - t0) / 5
print(f"B=1: host launch time {t_launch*1e3:.1f} ms/step, wall {t_total*1e3:.1f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(3):
    m.train_step(s1, 1e-4)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr).sort_stats("tottime")
st.print_stats(18)

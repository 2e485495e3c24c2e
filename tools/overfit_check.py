#!/usr/bin/env python3
"""Full-size end-to-end sanity: the step (forward, backward, AdamW) must be able to FIT a fixed synthetic batch.
Myriad stage 1 + LoRA, batch 8, shipped schedule scaled to a short run (lr 1e-4 cosine, wd 0.05)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_samples
from myriad_amd.myriad import MyriadHIP
from myriad_amd.runner import LinearWarmupCosineLRScheduler, setup_seeds
from myriad_amd.synthetic import SyntheticWeights, full_config

dev = torch.device("cuda:0")
setup_seeds(42, 0)
cfg = full_config()
model = MyriadHIP(SyntheticWeights(cfg, dev, seed=0), dict(fixed_stage=1, fixed_taskstage=0, vit_heads=cfg["vit_heads"],
                                                             qf_heads=cfg["qf_heads"], llm_heads=cfg["llm_heads"], use_lora=True),
                  device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sched = LinearWarmupCosineLRScheduler(None, max_epoch=1, iters_per_epoch=n, min_lr=1e-5, init_lr=3e-4, warmup_steps=5,
                                      warmup_start_lr=1e-6)
s = make_samples(8, cfg["vocab"], 42, dev)
t0 = time.time()
for i in range(n):
    loss = float(model.train_step(s, sched.step(0, i), 0.05, next_samples=s))
    if i % 5 == 0 or i == n - 1:
        print(f"step {i:3d}  loss {loss:.4f}", flush=True)
model.finish_update()
torch.cuda.synchronize()
print(f"{n} steps in {time.time() - t0:.1f} s")

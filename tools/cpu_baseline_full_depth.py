#!/usr/bin/env python3
"""bench.py's `cpu_baseline` extrapolates a reduced-depth sample (ViT 8/39, Q-Former 6/12, LLaMA 8/32 layers) linearly to full depth
so that the default bench finishes in minutes.  This tool runs the SAME oracle step (the CPU restatement pinned to the reference's
modules; test infrastructure, timed here as the baseline leg only) ONCE at full depth -- 39 / 12 / 32 layers, full width, B = 1,
V = 2048 as in bench.py -- and prints it beside the extrapolated figure from the same process, same thread count.
Needs ~35 GB of host memory (fp32 weights + autograd's saved activations).  Usage: python tools/cpu_baseline_full_depth.py [threads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from myriad_amd.synthetic import full_config
from oracle import myriad_ref as R            # the checker, timed as the CPU baseline
from tests import golden_utils as gu

nth = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.set_num_threads(nth)
cfg = full_config()
arch, stage, V = "myriad", 1, 2048
torch.manual_seed(0)
t0 = time.time()
sd = {}
sd.update(gu.vit_weights(cfg["vit_dim"], cfg["vit_depth"], cfg["vit_heads"], cfg["vit_hidden"], cfg["patch"], 257, seed=1))
sd.update(gu.qformer_weights(cfg["qf_dim"], cfg["qf_layers"], cfg["qf_inter"], cfg["vit_dim"], seed=2))
sd.update(gu.llama_weights(cfg["llm_dim"], cfg["llm_layers"], cfg["llm_inter"], V, seed=3))
sd.update(gu.adapter_weights(seed=4))
sd.update(gu.glue_weights(seed=5))
train = [k for k in sd if k.startswith(("expert_adaptor.", "VEInstructor.", "VETokenizer."))]
for k in train:
    sd[k] = sd[k].clone().requires_grad_(True)
batch = gu.synthetic_batch(1, V, seed=6)
print(f"weights built in {time.time() - t0:.0f} s; {nth} threads of {os.cpu_count()} logical cores", flush=True)


def run(depths):
    for k in train:
        sd[k].grad = None
    return bench._cpu_baseline_once(R, sd, train, batch, arch, stage, cfg, depths)


full = (cfg["vit_depth"], cfg["qf_layers"], cfg["llm_layers"])
run((2, 2, 2))                                     # page the weights in
ex = sorted(run((8, 6, 8))[0] for _ in range(3))[1]
print(f"extrapolated from the 8 / 6 / 8-layer sample (bench.py's protocol, median of 3): {ex:.2f} s/step = {1 / ex:.4f} images/s", flush=True)
t1 = time.perf_counter()
tot, meas = run(full)
wall = time.perf_counter() - t1
print(f"FULL depth {full}, one run: {meas:.2f} s measured (sum of the timed parts; wall {wall:.2f} s) -> {tot:.2f} s/step with the AdamW slice "
      f"scaled = {1 / tot:.4f} images/s")
print(f"extrapolation / full-depth = {ex / tot:.3f}")

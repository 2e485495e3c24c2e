#!/usr/bin/env python3
"""Soak test of the concurrent step (side-stream ViT lookahead, LoRA / leaf backward on side streams): N optimisation steps
over a rotating set of different batches, run twice from the same initial state; parameters and AdamW moments must be
bit-identical between the runs, and identical to a third run with every side stream switched off."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_samples
from myriad_amd.myriad import MyriadHIP
from myriad_amd.synthetic import SyntheticWeights, full_config

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
cfg = full_config()
model = MyriadHIP(SyntheticWeights(cfg, dev, seed=0), dict(fixed_stage=1, fixed_taskstage=0, vit_heads=cfg["vit_heads"],
                                                             qf_heads=cfg["qf_heads"], llm_heads=cfg["llm_heads"], use_lora=True),
                  device=dev)
st = model.store
keep = (st.flat_p.clone(), st.flat_m.clone(), st.flat_v.clone(), st.step, model.lora.step_seed, st.steps_dev.clone())
batches = [make_samples(b, cfg["vocab"], 100 + i, dev) for i, b in enumerate((8, 8, 4, 8, 2, 8, 8))]


def digest(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]


def run(side: bool):
    st.flat_p.copy_(keep[0]); st.flat_m.copy_(keep[1]); st.flat_v.copy_(keep[2]); st.step = keep[3]; st.steps_dev.copy_(keep[5])
    model.lora.step_seed = keep[4]                   # the LoRA dropout masks are a function of (step, layer)
    model._leaf_aside = side
    model.llama.defer_lora_wgrad = side
    model._vit_prefetched = None
    t0 = time.time()
    losses = []
    for i in range(n):
        cur, nxt = batches[i % len(batches)], batches[(i + 1) % len(batches)]
        losses.append(float(model.train_step(cur, 2e-4, 0.05, next_samples=nxt if side else None)))
    model.finish_update()
    torch.cuda.synchronize()
    return digest(st.flat_p), digest(st.flat_m), digest(st.flat_v), losses[-1], time.time() - t0


a = run(True)
b = run(True)
c = run(False)
print("side streams on :", a)
print("side streams on :", b)
print("side streams off:", c)
ok = a[:4] == b[:4] == c[:4]
print("IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)

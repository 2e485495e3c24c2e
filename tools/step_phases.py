#!/usr/bin/env python3
"""Unprofiled phase timeline of the fine-tune step: at each phase boundary record the host clock and a HIP event on the main
stream.  Where the event time tracks the host time the GPU is waiting for launches (launch-bound); where it lags, the GPU is the
bottleneck.  Usage: python tools/step_phases.py [--batch 8] [--steps 6]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_samples
from myriad_amd.myriad import MyriadHIP
from myriad_amd.synthetic import SyntheticWeights, full_config

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--no-prefetch", action="store_true")
a = ap.parse_args()
dev = "cuda:0"
cfg = full_config()
m = MyriadHIP(SyntheticWeights(cfg, dev, seed=0), dict(fixed_stage=1, fixed_taskstage=0, use_lora=True), device=dev)
m.train()
batches = [make_samples(a.batch, cfg["vocab"], 42 + i, dev) for i in range(a.steps + 4)]
marks = []


def mark(name):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append((name, time.perf_counter(), ev))


def wrap(obj, attr, name):
    orig = getattr(obj, attr)

    def f(*args, **kw):
        out = orig(*args, **kw)
        mark(name)
        return out
    setattr(obj, attr, f)


wrap(m, "prefetch_vit", "prefetch_vit enqueued")
wrap(m, "encode_img", "encode_img (adaptor, Q-Former fwd, VE nets)")
wrap(m.llama, "forward_loss", "llama.forward_loss")
wrap(m.llama, "backward", "llama.backward")
wrap(m.qformer, "backward", "qformer.backward")
wrap(m, "_finish_backward", "rest of backward (VE, adaptor, joins)")
wrap(m.store, "adamw_step", "adamw")
for i in range(3):
    m.train_step(batches[i], 1e-4, next_samples=None if a.no_prefetch else batches[i + 1])
torch.cuda.synchronize()
steps = []
for i in range(3, 3 + a.steps):
    marks.clear()
    mark("step start")
    m.train_step(batches[i], 1e-4, next_samples=None if a.no_prefetch else batches[i + 1])
    mark("train_step returned")
    steps.append(list(marks))
torch.cuda.synchronize()
t_end = time.perf_counter()
print(f"batch {a.batch}: mean wall {(t_end - steps[0][0][1]) / a.steps * 1e3:.2f} ms/step")
st = steps[-2]
h0, e0 = st[0][1], st[0][2]
print(f"{'phase':55s} {'host ms':>9s} {'gpu ms':>9s}  (offsets from the step's first mark; a later step, queues warm)")
for name, h, ev in st:
    print(f"{name:55s} {(h - h0) * 1e3:9.2f} {e0.elapsed_time(ev):9.2f}")
nxt = steps[-1][0]
print(f"{'next step start':55s} {(nxt[1] - h0) * 1e3:9.2f} {e0.elapsed_time(nxt[2]):9.2f}")

#!/usr/bin/env python3
"""Where the fixed cost of one generate() call goes (host wall clock with a device sync at each boundary)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd.myriad import MyriadHIP
from myriad_amd.synthetic import SyntheticWeights, full_config

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--new", type=int, default=32)
ap.add_argument("--lora", type=int, default=1)
a = ap.parse_args()
dev = "cuda:0"
cfg = full_config()
model = MyriadHIP(SyntheticWeights(cfg, dev, seed=0), dict(need_backward=False, use_lora=bool(a.lora)), device=dev)
model.eval()
g = torch.Generator().manual_seed(1)
B = a.batch
smp = dict(image=torch.randn(B, 3, 224, 224, generator=g), anomaly_maps=torch.rand(B, 1, 224, 224, generator=g),
           before_ids=torch.randint(3, 32000, (1, 4), generator=g).expand(B, -1).contiguous(),
           after_ids=torch.randint(3, 32000, (1, 28), generator=g).expand(B, -1).contiguous())
marks = []


def wrap(obj, attr, name):
    orig = getattr(obj, attr)

    def f(*args, **kw):
        if torch.cuda.is_current_stream_capturing():
            return orig(*args, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = orig(*args, **kw)
        torch.cuda.synchronize(); marks.append((name, time.perf_counter() - t0))
        return out
    setattr(obj, attr, f)


wrap(model, "encode_img", "encode_img (ViT + Q-Former + VE nets)")
wrap(model.llama, "_pack_for_decode", "pack_for_decode")
wrap(model.llama, "_decode_block", "decode_block (prefill / eager token step)")
if model.llama.lora is not None:
    wrap(model.llama.lora, "refresh", "lora.refresh")
_g = torch.cuda.graph
class TimedGraph(_g):
    def __enter__(self):
        torch.cuda.synchronize(); self._t0 = time.perf_counter(); return super().__enter__()
    def __exit__(self, *e):
        r = super().__exit__(*e); torch.cuda.synchronize(); marks.append(("graph capture", time.perf_counter() - self._t0)); return r
torch.cuda.graph = TimedGraph
for it in range(3):
    marks.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model.generate(smp, max_new_tokens=a.new, stop_ids=((-1,),), min_length=0, eos_token_id=-5)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
print(f"batch {B}, {out['token_ids'].shape[1]} tokens, lora={a.lora}: generate() {tot*1e3:.1f} ms")
agg = {}
for n, t in marks:
    agg.setdefault(n, [0, 0.0]); agg[n][0] += 1; agg[n][1] += t
for n, (c, t) in agg.items():
    print(f"  {n:45s} x{c:<3d} {t*1e3:8.2f} ms")
print(f"  {'everything else (replays, host bookkeeping)':45s}      {(tot - sum(t for _, t in marks))*1e3:8.2f} ms")

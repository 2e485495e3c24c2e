#!/usr/bin/env python3
"""Pure device time of the replayed token step vs what generate() achieves per token."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd.myriad import MyriadHIP
from myriad_amd.synthetic import SyntheticWeights, full_config
from myriad_amd import ops
dev = "cuda:0"
model = MyriadHIP(SyntheticWeights(full_config(), dev, seed=0), dict(need_backward=False), device=dev)
model.eval()
g = torch.Generator().manual_seed(1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
smp = dict(image=torch.randn(B, 3, 224, 224, generator=g), anomaly_maps=torch.rand(B, 1, 224, 224, generator=g),
           before_ids=torch.randint(3, 32000, (1, 4), generator=g).expand(B, -1).contiguous(),
           after_ids=torch.randint(3, 32000, (1, 28), generator=g).expand(B, -1).contiguous())
for n in (8, 8, 64):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.generate(smp, max_new_tokens=n, stop_ids=((-1,),), min_length=0, eos_token_id=-5)
    torch.cuda.synchronize(); print(f"generate({n}) {1e3 * (time.perf_counter() - t0):.1f} ms")
ws = next(iter(model.llama._decode_ws.values()))
gr = ws["graph"]
ops.add_i32_(ws["pos"], -40); ops.add_i32_(ws["kvlen"], -40); ws["step"].zero_()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30):
    gr.replay()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"30 replays: host enqueue {1e3 * (t1 - t0) / 30:.3f} ms each, device {1e3 * (t2 - t0) / 30:.3f} ms per step")
# the loop's own pieces: replay + one record copy, nothing else
ops.add_i32_(ws["pos"], -30); ops.add_i32_(ws["kvlen"], -30); ws["step"].zero_()
torch.cuda.synchronize()
tl = tc = 0.0
t00 = time.perf_counter()
for _ in range(30):
    a = time.perf_counter(); gr.replay(); b = time.perf_counter(); r = ws["rec"].cpu(); c = time.perf_counter()
    tl += b - a; tc += c - b
print(f"replay + rec.cpu(): {1e3 * (time.perf_counter() - t00) / 30:.3f} ms per step (replay call {1e3 * tl / 30:.3f}, copy+wait {1e3 * tc / 30:.3f})")
ops.add_i32_(ws["pos"], -30); ops.add_i32_(ws["kvlen"], -30); ws["step"].zero_()
torch.cuda.synchronize()
t00 = time.perf_counter()
for _ in range(30):
    gr.replay(); torch.cuda.current_stream().synchronize()
print(f"replay + stream sync: {1e3 * (time.perf_counter() - t00) / 30:.3f} ms per step")

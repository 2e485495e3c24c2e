#!/usr/bin/env python3
"""Data-path row f-2 measured: the shipped NSA augmentation (`patch_ex`, resize + seamlessClone NORMAL_CLONE, MVTec / VisA
argument tables) in images per second -- oracle (the pinned restatement of the reference's function) and the product's host
path on the box's cores, and the product's device path (host plan + csrc/selfsup.hip) on one MI355X, same seeded inputs.
python tools/selfsup_bench.py [--n 64] [--dataset mvtec --cls carpet]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from myriad_amd import self_sup as P
from oracle import self_sup_ref as O            # bench's CPU leg only (the oracle is never on the product path)

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=64)
ap.add_argument("--dataset", default="mvtec")
ap.add_argument("--cls", default="carpet")
ap.add_argument("--size", type=int, default=224)
a = ap.parse_args()
kw = dict(P.self_sup_args(a.dataset, a.cls))
label_mode = kw.get("label_mode", "binary")
ilp = kw.pop("intensity_logistic_params", (1 / 6, 20))
r = np.random.RandomState(0)
yy, xx = np.mgrid[0:a.size, 0:a.size]


def image(i):
    base = np.stack([120 + 60 * np.sin(xx / (7.0 + i % 5) + i), 110 + 50 * np.cos(yy / (9.0 + i % 3)), 90 + 0.4 * xx + 0.2 * yy], -1)
    return np.clip(base + r.randint(-12, 13, base.shape), 0, 255).astype(np.uint8)


dests = [image(i) for i in range(a.n)]
srcs = [image(1000 + i) for i in range(a.n)]

t0 = time.perf_counter()
for i in range(a.n):
    np.random.seed(100 + i)
    O.patch_ex(dests[i], srcs[i], intensity_logistic_params=ilp, **kw)
t_oracle = (time.perf_counter() - t0) / a.n

t0 = time.perf_counter()
plans = []
for i in range(a.n):
    np.random.seed(100 + i)
    plans.append(P.plan(dests[i], srcs[i], **kw))
t_plan = (time.perf_counter() - t0) / a.n
t0 = time.perf_counter()
host_out = []
for i in range(a.n):
    np.random.seed(100 + i)
    host_out.append(P.patch_ex(dests[i], srcs[i], intensity_logistic_params=ilp, **kw)[0])
t_host = (time.perf_counter() - t0) / a.n
nops = sum(len(p[0]) for p in plans)
nclone = sum(1 for p in plans for op in p[0] if op.mode == "normal_clone")
print(f"{a.dataset}/{a.cls}: {a.n} images {a.size}x{a.size}, {nops} patch operations ({nclone} Poisson clones)")
print(f"oracle (reference restatement), 1 thread: {t_oracle*1e3:.2f} ms/image = {1/t_oracle:.0f} images/s")
print(f"product host path (plan + numpy apply), 1 thread: {t_host*1e3:.2f} ms/image = {1/t_host:.0f} images/s  (plan alone {t_plan*1e3:.2f} ms)")
if torch.cuda.is_available():
    ex = P.PatchExHIP("cuda")
    for B in (8, 64):
        B = min(B, a.n)
        d = torch.from_numpy(np.stack(dests[:B])).cuda()
        s_ = torch.from_numpy(np.stack(srcs[:B])).cuda()
        out, label, union = ex(d, s_, plans[:B], label_mode=label_mode, intensity_logistic_params=ilp)   # warm-up + check
        torch.cuda.synchronize()
        same = all(np.array_equal(out[i].cpu().numpy(), host_out[i]) for i in range(B))
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ex(d, s_, plans[:B], label_mode=label_mode, intensity_logistic_params=ilp)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"device apply, batch {B}: {best/B*1e3:.2f} ms/image = {B/best:.0f} images/s (+ host plan {t_plan*1e3:.2f} ms/image on one core)"
              f"; pixels equal to the host path: {same}")

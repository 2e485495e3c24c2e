#!/usr/bin/env python3
"""time the ACTUAL reference modules on this container's host cores (build container only; the
reference never travels).  Modules are loaded by file path with the shims of tools/make_golden.py and composed as
minigpt4/models/myriad.py:241-272,354-431 does: ViT-39 + Q-Former-12 at full size, LLaMA at k in {2, 6} layers extrapolated
linearly to 32, fp32, B=1, forward + backward to the trainable parameters + AdamW (torch.optim.AdamW, as optims/runner build it).
Usage: python tools/time_reference_cpu.py [--ref /root/reference] [--steps 3] > profiles/r02_reference_cpu.md"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as mg  # noqa: E402
from tests import golden_utils as gu  # noqa: E402


KS = (2, 6)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--vit-depth", type=int, default=39)
    a = ap.parse_args()
    nth = os.cpu_count()
    torch.set_num_threads(nth)
    torch.manual_seed(0)
    M = mg.load_reference(a.ref)
    N = M["networks"]
    V = 32000
    vit = mg.ref_vit(M, 1408, a.vit_depth, 16, 4.3637, 224)
    qf = mg.ref_qformer(M, 768, 12, 12, 3072, 1408, 32)
    lnv, proj = nn.LayerNorm(1408), nn.Linear(768, 4096)
    qtok = torch.zeros(1, 32, 768).normal_(std=0.02)
    for mod in (vit, qf, lnv, proj):
        for p in mod.parameters():
            p.requires_grad = False
    print(f"# Reference modules timed on the build container's host cores ({nth} threads, fp32, B=1)\n")
    print("`tools/time_reference_cpu.py`: eva_vit.VisionTransformer (39 blocks), Qformer.BertLMHeadModel (12 layers, surgery of "
          "myriad.py:151-156), modeling_llama.LlamaForCausalLM at k layers, networks.{LoraAdaptorV2, VEInstructorV2, VETokenizer}, "
          "composed as myriad.py:241-272,354-431; forward + backward to the trainables + torch.optim.AdamW; 1 warm-up + "
          f"{a.steps} timed steps (medians); random-init weights, synthetic 224x224 image + 32-token prompt + 16-token target.\n")
    print("| arch | S | LLaMA layers k | ViT fwd s | rest fwd s | bwd s | AdamW s | step s |")
    print("|---|---|---|---|---|---|---|---|")
    res = {}
    for arch, stage in (("mini_gpt4", 0), ("myriad", 1)):
        for k in KS:
            lm = mg.ref_llama(M, 4096, k, 32, 11008, V)
            for p in lm.parameters():
                p.requires_grad = False
            ad, ins, tok = N.LoraAdaptorV2(dims=1408, input_dim=4), N.VEInstructorV2(), N.VETokenizer()
            train = [p for m in (ad, ins, tok) for p in m.parameters()] if arch == "myriad" else []
            if arch == "mini_gpt4":
                for p in proj.parameters():
                    p.requires_grad = True
                train = list(proj.parameters())
            opt = torch.optim.AdamW(train, lr=1e-4, weight_decay=0.05, betas=(0.9, 0.999))
            image, maps, before, after, tgt, tmask = gu.synthetic_batch(1, V, seed=6)
            embed = lm.model.embed_tokens
            acc = dict(vit=[], fwd=[], bwd=[], opt=[])
            S = 0
            for it in range(a.steps + 1):
                t0 = time.perf_counter()
                with torch.no_grad():
                    x = vit(image)
                t1 = time.perf_counter()
                if arch == "myriad":
                    x = ad(x)
                x = lnv(x.float())
                q = qtok.expand(1, -1, -1)
                if arch == "myriad" and stage in (1, 2):
                    q = torch.cat([q, ins(maps)], 1)
                qo = qf.bert(query_embeds=q, encoder_hidden_states=x, encoder_attention_mask=torch.ones(1, 257, dtype=torch.long),
                             return_dict=True).last_hidden_state
                img = proj(qo)
                if arch == "myriad" and stage in (0, 1):
                    img = torch.cat([img, tok(maps)], 1)
                wrapped = torch.cat([embed(before), img, embed(after)], 1)
                targets = tgt.masked_fill(tgt == 2, -100)
                labels = torch.cat([torch.full((1, wrapped.shape[1] + 1), -100, dtype=torch.long), targets], 1)
                emb = torch.cat([embed(torch.ones(1, 1, dtype=torch.long)), wrapped, embed(tgt)], 1)
                attn = torch.cat([torch.ones(1, 1 + wrapped.shape[1], dtype=torch.long), tmask], 1)
                S = emb.shape[1]
                loss = lm(inputs_embeds=emb, attention_mask=attn, labels=labels, return_dict=True).loss
                t2 = time.perf_counter()
                opt.zero_grad()
                loss.backward()
                t3 = time.perf_counter()
                opt.step()
                t4 = time.perf_counter()
                if it > 0:
                    acc["vit"].append(t1 - t0)
                    acc["fwd"].append(t2 - t1)
                    acc["bwd"].append(t3 - t2)
                    acc["opt"].append(t4 - t3)
            r = {kk: sorted(v)[len(v) // 2] for kk, v in acc.items()}      # medians: the container's cores are shared
            r["step"] = sum(r.values())
            res[(arch, k)] = (r, S)
            print(f"| {arch} stage {stage} | {S} | {k} | {r['vit']:.2f} | {r['fwd']:.2f} | {r['bwd']:.2f} | {r['opt']:.3f} | {r['step']:.2f} |", flush=True)
            del lm, opt
        (r2, S), (r4, _) = res[(arch, KS[0])], res[(arch, KS[1])]
        per_layer = ((r4["fwd"] + r4["bwd"]) - (r2["fwd"] + r2["bwd"])) / (KS[1] - KS[0])     # LLaMA layers only differ in fwd / bwd
        vit_s = min(r2["vit"], r4["vit"]) * 39.0 / a.vit_depth
        full = vit_s + r2["fwd"] + r2["bwd"] + per_layer * (32 - KS[0]) + min(r2["opt"], r4["opt"])
        print(f"| {arch} stage {stage} | {S} | **32 (extrapolated: {per_layer:.2f} s per layer)** | | | | | **{full:.1f} s/step = {1.0 / full:.4f} images/s** |", flush=True)
    print("\nLoRA on q/v (peft, un-vendored and absent here) is not part of this timing; it adds 2 x 32 rank-8 products to a step "
          "that is dominated by the frozen 7B matmuls.")


if __name__ == "__main__":
    main()

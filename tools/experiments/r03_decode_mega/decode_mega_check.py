import os, sys, torch
sys.path.insert(0, "/root/repo")
from myriad_amd.llama import LlamaHIP
from tests import golden_utils as gu
DEV = "cuda:0"
D, layers, heads, inter, V = 4096, 3, 32, 11008, 32000
sd = gu.llama_weights(D, layers, inter, V, seed=931, std=0.03)
import itertools
for B in [2, 1] * 20:
    emb = (torch.randn(B, 37, D, generator=torch.Generator().manual_seed(932)) * 0.05).to(DEV)
    outs = []
    for mega in (False, True):
        lm = LlamaHIP(sd, heads, DEV, need_backward=False)
        lm.decode_mega = mega
        ids = lm.greedy_generate(emb, max_new_tokens=40, stop_ids=(), min_length=1)
        ws = next(iter(lm._decode_ws.values()))
        outs.append((ids.clone(), ws["logits"].clone(), [c.clone() for c in ws["caches"]]))
        if mega:
            print("   abort flag", int(ws["mega"]["bar"][1]), end="")
        del lm
    same_ids = torch.equal(outs[0][0], outs[1][0])
    first = None if same_ids else int((outs[0][0] != outs[1][0]).any(0).nonzero()[0])
    dl = (outs[0][1] - outs[1][1]).abs().max().item()
    dc = [float((a.float() - b.float()).abs().max()) for a, b in zip(outs[0][2], outs[1][2])]
    print(f"B={B}: ids equal {same_ids} first diff step {first}; logits maxdiff {dl:.3e}; cache maxdiff per layer {dc}", flush=True)

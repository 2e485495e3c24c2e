import sys, torch
sys.path.insert(0, "/root/repo")
from tests import golden_utils as gu
from tests.test_model_gpu import load, bf16_round, relerr, DEV
from myriad_amd.llama import LlamaHIP
from oracle import myriad_ref as R
g = load("llama_tiny")
D, layers, heads, inter, V, seed = [int(x) for x in g["meta"]]
print("D", D, "layers", layers, "heads", heads, "inter", inter, "V", V)
sd_full = bf16_round(gu.llama_weights(D, layers, inter, V, seed=seed, std=0.2))
emb = g["emb"]
for k in range(1, layers + 1):
    sd = {n: t for n, t in sd_full.items() if ".layers." not in n or int(n.split(".layers.")[1].split(".")[0]) < k}
    lm = LlamaHIP(sd, heads, DEV)
    e2 = emb.clone().requires_grad_(True)
    lr, _ = R.llama_causal_lm(sd, e2, g["mask"], g["labels"], heads)
    lr.backward()
    loss = lm.forward_loss(emb.to(DEV), g["mask"], g["labels"])
    demb = lm.backward()
    print(k, "loss rel", abs(loss.item() - lr.item()) / abs(lr.item()), "grad maxabs-rel", relerr(demb, e2.grad), "fro", ((demb.cpu() - e2.grad).norm() / e2.grad.norm()).item())

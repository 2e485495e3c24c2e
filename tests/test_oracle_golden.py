"""Pin the CPU oracle (oracle/myriad_ref.py) against golden vectors produced by the reference's own
modules (tools/make_golden.py, run in the build container).  CPU-only; no reference at run time."""
import os

import numpy as np
import pytest
import torch

from oracle import myriad_ref as R
from tests import golden_utils as gu

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(G, name + ".npz")).items()}


def close(a, b, rtol=2e-4, atol=2e-5):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, f"max err {err:.3e} vs ref scale {ref:.3e}"


def test_vit_tiny():
    g = load("vit_tiny")
    D, depth, heads, hidden, img, seed = [int(x) for x in g["meta"]]
    sd = gu.vit_weights(D, depth, heads, hidden, 14, 17, seed=seed)
    close(R.vit_forward(sd, g["image"], heads), g["out"])


def test_vit_fullwidth_head_dim_88():
    g = load("vit_fullwidth")
    D, depth, heads, hidden, img, seed = [int(x) for x in g["meta"]]
    sd = gu.vit_weights(D, depth, heads, hidden, 14, 257, seed=seed)
    x = torch.randn(1, 3, img, img, generator=torch.Generator().manual_seed(int(g["image_seed"][0])))
    y = R.vit_forward(sd, x, heads)
    close(y[:, ::8, ::4], g["out_sub"])
    close(y.mean(), g["out_mean"], atol=1e-5)


def test_networks_forward_and_grads():
    g = load("networks_full")
    sd = {k: v.clone().requires_grad_(True) for k, v in gu.adapter_weights(seed=int(g["seed"][0])).items()}
    gen = torch.Generator().manual_seed(int(g["seed"][1]))
    x = torch.randn(2, 257, 1408, generator=gen, requires_grad=True)
    maps = torch.rand(2, 1, 224, 224, generator=gen)
    ct_a = torch.randn(2, 257, 1408, generator=gen)
    ct_i = torch.randn(2, 49, 768, generator=gen)
    ct_t = torch.randn(2, 18, 4096, generator=gen)
    ya, yi, yt = R.lora_adaptor(sd, x), R.ve_instructor(sd, maps), R.ve_tokenizer(sd, maps)
    close(ya[:, ::16, ::8], g["adaptor_out_sub"])
    close(yi, g["instr_out"])
    close(yt[:, :, ::8], g["tok_out_sub"])
    ((ya * ct_a).sum() + (yi * ct_i).sum() + (yt * ct_t).sum()).backward()
    close(x.grad[:, ::16, ::8], g["dx_sub"])
    close(sd["expert_adaptor.conv1.weight"].grad, g["dA"], rtol=1e-3)
    close(sd["expert_adaptor.conv2.weight"].grad, g["dB"], rtol=1e-3)
    for nm, pre in (("instr", "VEInstructor."), ("tok", "VETokenizer.")):
        for idx in (0, 3, 6, 9, 12, 15):
            w = sd[pre + f"meta_net.{idx}.weight"].grad
            close(w.norm(), g[f"{nm}_dw{idx}_norm"], rtol=1e-3)
            close(w.reshape(w.shape[0], -1)[:8, :32], g[f"{nm}_dw{idx}_sub"], rtol=1e-3, atol=1e-4)
            close(sd[pre + f"meta_net.{idx}.bias"].grad[:64], g[f"{nm}_db{idx}"], rtol=1e-3, atol=1e-4)
    close(sd["VETokenizer.base_prompts"].grad[:, ::64], g["tok_dbase"])


def test_qformer_tiny():
    g = load("qformer_tiny")
    D, layers, heads, inter, enc_w, seed = [int(x) for x in g["meta"]]
    sd = gu.qformer_weights(D, layers, inter, enc_w, seed=seed)
    q = g["query"].clone().requires_grad_(True)
    e = g["enc"].clone().requires_grad_(True)
    y = R.qformer_forward(sd, q, e, heads)
    close(y, g["out"])
    (y * g["ct"]).sum().backward()
    close(q.grad, g["dquery"], rtol=1e-3)
    close(e.grad, g["denc"], rtol=1e-3)


def test_qformer_fullwidth_81_queries():
    g = load("qformer_fullwidth")
    D, layers, heads, inter, enc_w, seed = [int(x) for x in g["meta"]]
    sd = gu.qformer_weights(D, layers, inter, enc_w, seed=seed)
    gen = torch.Generator().manual_seed(int(g["seed"][1]))
    q = torch.randn(1, 81, D, generator=gen, requires_grad=True)
    e = torch.randn(1, 257, enc_w, generator=gen, requires_grad=True)
    ct = torch.randn(1, 81, D, generator=gen)
    y = R.qformer_forward(sd, q, e, heads)
    close(y[:, :, ::4], g["out_sub"])
    (y * ct).sum().backward()
    close(q.grad[:, :, ::4], g["dquery_sub"], rtol=1e-3)
    close(e.grad[:, ::4, ::8], g["denc_sub"], rtol=1e-3)


def test_llama_tiny_loss_logits_grad_and_greedy():
    g = load("llama_tiny")
    D, layers, heads, inter, V, seed = [int(x) for x in g["meta"]]
    sd = gu.llama_weights(D, layers, inter, V, seed=seed, std=0.2)
    emb = g["emb"].clone().requires_grad_(True)
    loss, logits = R.llama_causal_lm(sd, emb, g["mask"], g["labels"], heads)
    close(loss, g["loss"])
    # rows at padded positions are defined but label-free: compare all rows anyway
    close(logits, g["logits"], rtol=5e-4)
    loss.backward()
    close(emb.grad, g["demb"], rtol=1e-3)
    with torch.no_grad():
        ids, margins = R.greedy_generate(sd, g["emb"][:2, :7], heads, max_new_tokens=12, stop_ids=((7,),),
                                         return_margins=True)
    assert torch.equal(ids, g["gen_ids"]), (ids, g["gen_ids"])
    close(margins, g["gen_margins"], rtol=1e-3, atol=1e-4)


def test_decode_chain_every_id_equals_the_reference():
    """Peaked-logit fixture (tools/make_golden.py case_decode_chain: the reference's forward(use_cache=True) loop with HF's
    sample-loop bookkeeping): the eval script's stop ids, the min_length EOS ban, an early-finished row, 835 on a row that
    is not row 0.  No margin gating: every id equal."""
    g = load("decode_chain")
    c = gu.DECODE_CHAIN
    sd = gu.decode_chain_weights()
    for name, rows in (("b4", ["row0", "row1", "row2", "row3"]), ("b1", ["row0"]), ("stop835", ["stop835"])):
        with torch.no_grad():
            ids = R.greedy_generate(sd, gu.decode_chain_inputs(rows), c["heads"], max_new_tokens=90, min_length=1)
        assert torch.equal(ids, g[name + "_ids"]), (name, ids, g[name + "_ids"])
    assert g["b4_ids"].shape[1] == 32 and g["b4_ids"][0, -2:].tolist() == [2277, 29937]      # two-token stop on row 0
    assert g["b4_ids"][2, 3:].eq(2).all() and g["b4_ids"][3, 1] == 835                          # padded row; 835 off row 0


def test_llama_fullwidth_layer():
    g = load("llama_fullwidth")
    D, layers, heads, inter, V, seed = [int(x) for x in g["meta"]]
    sd = gu.llama_weights(D, layers, inter, V, seed=seed)
    gen = torch.Generator().manual_seed(int(g["seed"][1]))
    emb = (torch.randn(2, 24, D, generator=gen) * 0.02).requires_grad_(True)
    loss, logits = R.llama_causal_lm(sd, emb, g["mask"], g["labels"], heads)
    close(loss, g["loss"])
    close(logits[:, :, ::10], g["logits_sub"], rtol=5e-4)
    loss.backward()
    close(emb.grad[:, :, ::16], g["demb_sub"], rtol=1e-3)


@pytest.mark.parametrize("name", ["normal", "saturated"])
def test_clamp_ce(name):
    g = load("clamp_ce")
    x = g[f"{name}_x"].clone().requires_grad_(True)
    loss = R.clamp_ce_loss(x, g[f"{name}_y"])
    close(loss, g[f"{name}_loss"], rtol=1e-6)
    loss.backward()
    close(x.grad, g[f"{name}_dx"], rtol=1e-5, atol=1e-9)
    if name == "saturated":
        # both clamp ends active -> whole-row zero gradient
        assert x.grad[0].abs().max() == 0 and x.grad[1].abs().max() == 0


def _composite_sd(seeds):
    sd = {}
    sd.update(gu.vit_weights(1408, 1, 16, int(1408 * 4.3637), 14, 257, seed=seeds[0]))
    sd.update(gu.qformer_weights(768, 2, 3072, 1408, seed=seeds[1]))
    sd.update(gu.llama_weights(4096, 1, 11008, 1000, seed=seeds[2]))
    sd.update(gu.adapter_weights(seed=seeds[3]))
    sd.update(gu.glue_weights(seed=seeds[4]))
    return sd


@pytest.mark.parametrize("arch,stage", [("mini_gpt4", 0), ("myriad", 0), ("myriad", 1), ("myriad", 2)])
def test_composite_forward_loss_and_trainable_grads(arch, stage):
    g = load("composite_fullwidth")
    seeds = [int(x) for x in g["seed"]]
    sd = _composite_sd(seeds)
    train = [k for k in sd if k.startswith(("expert_adaptor.", "VEInstructor.", "VETokenizer."))]
    for k in train:
        sd[k] = sd[k].clone().requires_grad_(True)
    image, maps, before, after, tgt, tmask = gu.synthetic_batch(2, 1000, seed=seeds[5], pad_tail=1)
    loss = R.model_forward(sd, image, maps, stage, before, after, tgt, tmask, arch=arch)
    key = f"{arch}_s{stage}"
    close(loss, g[key + "_loss"], rtol=2e-4)
    n_img = {"mini_gpt4": 32, "myriad0": 50, "myriad1": 99, "myriad2": 81}[arch if arch == "mini_gpt4" else f"myriad{stage}"]
    assert int(g[key + "_S"][0]) == 1 + 32 + n_img + 16
    if arch == "myriad":
        loss.backward()
        close(sd["expert_adaptor.conv1.weight"].grad, g[key + "_dA"], rtol=2e-3, atol=1e-7)
        close(sd["expert_adaptor.conv2.weight"].grad[::16], g[key + "_dB_sub"], rtol=2e-3, atol=1e-7)
        if stage in (1, 2):
            close(sd["VEInstructor.meta_net.15.weight"].grad.norm(), g[key + "_instr_dw15_norm"], rtol=2e-3)
            close(sd["VEInstructor.meta_net.0.weight"].grad, g[key + "_instr_dw0"], rtol=2e-3, atol=1e-7)
        if stage in (0, 1):
            close(sd["VETokenizer.meta_net.15.weight"].grad.norm(), g[key + "_tok_dw15_norm"], rtol=2e-3)
            close(sd["VETokenizer.meta_net.0.weight"].grad, g[key + "_tok_dw0"], rtol=2e-3, atol=1e-7)
            close(sd["VETokenizer.base_prompts"].grad[:, ::64], g[key + "_tok_dbase_sub"], rtol=2e-3, atol=1e-9)


def test_lr_schedule_and_adamw():
    g = load("optim")
    for (e, s), lr in zip(g["pts"].tolist(), g["lrs"].tolist()):
        assert abs(R.lr_at(e, s, 1600, 10, 1e-4, 0.0, 0, 1e-6) - lr) <= 1e-12
    for (e, s), lr in zip(g["pts2"].tolist(), g["lrs2"].tolist()):
        assert abs(R.lr_at(e, s, 100, 2, 1e-3, 1e-5, 20, 1e-6) - lr) <= 1e-12
    p, b = g["p0"].clone(), g["b0"].clone()
    mp, vp, mb, vb = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(b), torch.zeros_like(b)
    for k in range(3):
        lr = 1e-4 * (1 - 0.1 * k)
        R.adamw_step(p, g["gp"][k], mp, vp, k + 1, lr, 0.05)
        R.adamw_step(b, g["gb"][k], mb, vb, k + 1, lr, 0.0)
    close(p, g["p3"], rtol=1e-6, atol=1e-7)
    close(b, g["b3"], rtol=1e-6, atol=1e-7)
    assert R.uses_weight_decay("VETokenizer.meta_net.15.weight", 4)
    assert R.uses_weight_decay("VETokenizer.base_prompts", 2)
    assert not R.uses_weight_decay("VETokenizer.meta_net.15.bias", 1)
    assert not R.uses_weight_decay("ln_vision.weight", 1)


def test_pipeline_chain_every_id_equals_the_reference():
    """Full-pipeline peaked fixture (tools/make_golden.py case_pipeline_chain: the reference's ViT / Q-Former / networks /
    LLaMA composed as myriad.py:241-272,433-454): the oracle's encode_img + prompt wrap + greedy loop reproduces every id,
    at batch 4 and for row 3 alone."""
    g = load("pipeline_chain")
    sd = gu.pipeline_chain_weights(g["probe"])
    image, maps, before, after = gu.pipeline_chain_batch()
    ew = sd["llama_model.model.embed_tokens.weight"]
    with torch.no_grad():
        for name, sel in (("b4", [0, 1, 2, 3]), ("b1r3", [3])):
            img = R.encode_img(sd, image[sel], maps[sel], 1, "myriad")
            wrapped = torch.cat([ew[before[sel]], img, ew[after[sel]]], 1)
            ids, margins = R.greedy_generate(sd, wrapped, 32, max_new_tokens=90, min_length=1, return_margins=True)
            assert torch.equal(ids, g[name + "_ids"]), (name, ids, g[name + "_ids"])
            close(margins, g[name + "_margins"], rtol=2e-3, atol=2e-3)


# ---- a-14: known-answer vectors for peft's LoRA form, worked by hand (no peft, no oracle code in the expected values) ----
#   x = [1 2 3 4]   W = [[1 0 2 0] [0 1 0 1] [1 1 0 0] [0 0 1 2]]   A (r=2) = [[1 0 1 0] [0 2 0 1]]   B = [[1 0] [0 1] [2 1] [1 3]]
#   alpha = 4, r = 2 -> scale 2.   W x = [7 6 3 11];  A x = [4 8];  B (A x) = [4 8 16 28];  y = W x + 2 B A x = [15 22 35 67]
#   dropout p = 0.5 keeping elements 0 and 2 (factors [2 0 2 0]): drop(x) = [2 0 6 0]; A drop(x) = [8 0]; B . = [8 0 16 8];
#   y = [7 6 3 11] + 2 [8 0 16 8] = [23 6 35 27]
#   backward (no dropout) with dy = [1 0 -1 2]:  B^T dy = [1 5];  dA = 2 [1 5]^T x = [[2 4 6 8] [10 20 30 40]];
#   dB = 2 dy (A x)^T = [[8 16] [0 0] [-8 -16] [16 32]];  dx = W^T dy + 2 A^T B^T dy = [0 -1 4 4] + [2 20 2 10] = [2 19 6 14]
LORA_KAT = dict(
    x=[1., 2., 3., 4.], W=[[1., 0., 2., 0.], [0., 1., 0., 1.], [1., 1., 0., 0.], [0., 0., 1., 2.]],
    A=[[1., 0., 1., 0.], [0., 2., 0., 1.]], B=[[1., 0.], [0., 1.], [2., 1.], [1., 3.]], alpha=4.0, r=2,
    y=[15., 22., 35., 67.], keep=[2., 0., 2., 0.], y_drop=[23., 6., 35., 27.], dy=[1., 0., -1., 2.],
    dA=[[2., 4., 6., 8.], [10., 20., 30., 40.]], dB=[[8., 16.], [0., 0.], [-8., -16.], [16., 32.]], dx=[2., 19., 6., 14.])


def test_peft_lora_formula_known_answer():
    k = LORA_KAT
    x = torch.tensor([k["x"]], requires_grad=True)
    W = torch.tensor(k["W"])
    A = torch.tensor(k["A"], requires_grad=True)
    Bm = torch.tensor(k["B"], requires_grad=True)
    y = R.lora_linear(x, W, A, Bm, k["alpha"], k["r"])
    assert y.tolist() == [k["y"]]
    y.backward(torch.tensor([k["dy"]]))
    assert A.grad.tolist() == k["dA"] and Bm.grad.tolist() == k["dB"] and x.grad.tolist() == [k["dx"]]
    y2 = R.lora_linear(x.detach(), W, A.detach(), Bm.detach(), k["alpha"], k["r"], torch.tensor([k["keep"]]))
    assert y2.tolist() == [k["y_drop"]]

"""checkpoint I/O at the reference's file formats (host-side, CPU)."""
import json
import os

import numpy as np
import pytest
import torch

from myriad_amd import checkpoint as C
from myriad_amd.myriad import ParamStore
from myriad_amd.networks import to_reference_layout, ve_param_specs

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ckpt_io.npz"))


@pytest.mark.parametrize("tag", ["up", "down", "same", "real"])
def test_interpolate_pos_embed_matches_reference(tag):
    out = C.interpolate_pos_embed(torch.from_numpy(G[f"{tag}_in"]), int(G[f"{tag}_n"]))
    assert out.dtype == torch.float32
    np.testing.assert_allclose(out.numpy(), G[f"{tag}_out"], rtol=0, atol=1e-6)


def test_hf_shards_bin_and_safetensors(tmp_path):
    g = torch.Generator().manual_seed(0)
    full = {f"model.layers.{i}.self_attn.q_proj.weight": torch.randn(8, 8, generator=g) for i in range(4)}
    full["model.embed_tokens.weight"] = torch.randn(16, 8, generator=g)
    full["lm_head.weight"] = torch.randn(16, 8, generator=g)
    keys = list(full)
    # sharded .bin with an index
    d1 = tmp_path / "bin"
    d1.mkdir()
    wm = {}
    for s, part in enumerate((keys[:3], keys[3:])):
        name = f"pytorch_model-0000{s + 1}-of-00002.bin"
        torch.save({k: full[k] for k in part}, d1 / name)
        wm.update({k: name for k in part})
    json.dump({"metadata": {}, "weight_map": wm}, open(d1 / "pytorch_model.bin.index.json", "w"))
    got = C.load_hf_shards(str(d1))
    assert set(got) == set(full) and all(torch.equal(got[k], full[k]) for k in full)
    # sharded safetensors
    from safetensors.torch import save_file
    d2 = tmp_path / "st"
    d2.mkdir()
    wm = {}
    for s, part in enumerate((keys[:2], keys[2:])):
        name = f"model-0000{s + 1}-of-00002.safetensors"
        save_file({k: full[k].contiguous() for k in part}, str(d2 / name))
        wm.update({k: name for k in part})
    json.dump({"weight_map": wm}, open(d2 / "model.safetensors.index.json", "w"))
    got = C.load_hf_shards(str(d2))
    assert all(torch.equal(got[k], full[k]) for k in full)
    # an index that names a tensor no shard holds must fail loudly
    wm["model.norm.weight"] = "model-00002-of-00002.safetensors"
    json.dump({"weight_map": wm}, open(d2 / "model.safetensors.index.json", "w"))
    with pytest.raises(KeyError):
        C.load_hf_shards(str(d2))
    with pytest.raises(FileNotFoundError):
        C.load_hf_shards(str(tmp_path))


def test_assemble_reference_weights_key_names():
    vit = {"cls_token": torch.zeros(1, 1, 8), "pos_embed": torch.randn(1, 1 + 16, 8), "blocks.0.attn.qkv.weight": torch.zeros(24, 8)}
    qf = {"model": {"Qformer.bert.encoder.layer.0.attention.self.query.weight": torch.zeros(4, 4), "query_tokens": torch.zeros(1, 32, 4),
                    "ln_vision.weight": torch.ones(8), "opt_proj.weight": torch.zeros(2, 2), "visual_encoder.cls_token": torch.ones(1, 1, 8)}}
    mg = {"model": {"llama_proj.weight": torch.zeros(6, 4), "llama_proj.bias": torch.zeros(6)}}
    ll = {"model.layers.0.self_attn.q_proj.weight": torch.zeros(6, 6), "model.layers.0.self_attn.rotary_emb.inv_freq": torch.zeros(3),
          "base_model.model.model.layers.0.self_attn.v_proj.weight": torch.ones(6, 6),
          "base_model.model.model.layers.0.self_attn.v_proj.lora_A.default.weight": torch.ones(2, 6), "lm_head.weight": torch.zeros(9, 6)}
    w = C.assemble_reference_weights(vit, qf, mg, ll, num_patches=49)
    assert w["visual_encoder.pos_embed"].shape == (1, 50, 8)                      # 4x4 grid -> 7x7, class token kept
    assert torch.equal(w["visual_encoder.cls_token"], torch.zeros(1, 1, 8))        # the ViT file wins over the BLIP-2 copy
    assert set(k for k in w if not k.startswith("visual_encoder.")) == {
        "Qformer.bert.encoder.layer.0.attention.self.query.weight", "query_tokens", "ln_vision.weight", "llama_proj.weight",
        "llama_proj.bias", "llama_model.model.layers.0.self_attn.q_proj.weight", "llama_model.model.layers.0.self_attn.v_proj.weight",
        "llama_model.lm_head.weight"}


class _ToyModel:
    """state_dict / load_state_dict / store of MyriadHIP without the GPU parts."""

    def __init__(self):
        specs = ve_param_specs("VEInstructor.", 768, 1)[:6] + [("expert_adaptor.conv1.weight", (4, 1408), (4, 1408)),
                                                               ("VETokenizer.base_prompts", (9, 4096), (9, 4096))]
        self.store = ParamStore(specs, "cpu")             # real specs: 4-D conv weights (layout permutation), 1-D biases, 2-D tables
        g = torch.Generator().manual_seed(1)
        for t in (self.store.flat_p, self.store.flat_m, self.store.flat_v):
            t.copy_(torch.randn(t.shape, generator=g))
        self.store.flat_v.abs_()
        self.store.step = 17
        self.store.set_module_steps({"expert_adaptor": 17, "VETokenizer": 11, "VEInstructor": 9})   # modules step separately

    def state_dict(self):
        return {n: to_reference_layout(self.store.p[n], r).clone() for n, _, r in self.store.specs}

    def load_state_dict(self, sd, strict=False):
        from myriad_amd.networks import from_reference_layout
        for n, i, _ in self.store.specs:
            if n in sd:
                self.store.p[n].copy_(from_reference_layout(sd[n].float(), i))


def test_optimizer_indices_follow_the_reference_named_parameters_order():
    """tests/golden/param_order.json = named_parameters() of the reference's own networks.py modules registered in
    Myriad.__init__'s order, split into RunnerBase.optimizer's two groups (tools/make_golden_host.py --param-order)."""
    import json
    from myriad_amd.lora import lora_param_specs
    from myriad_amd.myriad import uses_weight_decay
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "param_order.json")))
    ref_names = [n for n, _ in g["named_parameters"]]
    shapes = {n: tuple(s) for n, s in g["named_parameters"]}
    specs = ([("expert_adaptor.conv1.weight", (4, 1408), (4, 1408)), ("expert_adaptor.conv2.weight", (1408, 4), (1408, 4))]
             + ve_param_specs("VETokenizer.", 4096, 5) + [("VETokenizer.base_prompts", (9, 4096), (9, 4096))]
             + ve_param_specs("VEInstructor.", 768, 1))
    assert {n: tuple(r) for n, _, r in specs} == shapes                       # same tensors, same reference shapes
    assert C.reference_param_order([n for n, _, _ in specs][::-1]) == ref_names
    store = ParamStore(lora_param_specs(2, 64, 8) + specs, "cpu")             # LoRA first in the flat buffer: irrelevant here
    order, n_wd = C._optimizer_index(store)
    lora = [n for n in order if "lora_" in n]
    assert order[:n_wd] == g["weight_decay_group"] + lora and order[n_wd:] == g["no_decay_group"]
    assert [n.split("self_attn.")[1].rsplit(".default", 1)[0] for n in lora[:4]] == g["peft_lora_order_restated"]
    assert all(uses_weight_decay(n, len(shapes.get(n, (1, 1)))) for n in order[:n_wd])


def test_checkpoint_round_trip_and_torch_adamw_compat(tmp_path):
    m = _ToyModel()
    mgr = C.CheckpointManager(str(tmp_path), max_checkpoints=2)
    paths = [mgr.save(m, e, lr=1e-4, config={"run": {"max_epoch": 3}}) for e in range(3)]
    assert [os.path.exists(p) for p in paths] == [False, True, True]               # history cap (runner_base.py:618-626)
    ck = torch.load(paths[-1], map_location="cpu")
    assert set(ck) == {"model", "optimizer", "config", "scaler", "epoch"} and ck["scaler"] is None and ck["epoch"] == 2
    assert sorted(ck["model"]) == sorted(n for n, _, _ in m.store.specs)
    # the optimizer block is a valid torch.optim.AdamW state for the reference's two parameter groups, built the
    # reference's way: parameters in named_parameters() order, weight-decay group first (runner_base.py:110-131)
    order = C.reference_param_order(list(ck["model"]))
    shapes = {n: r for n, _, r in m.store.specs}
    wd = [n for n in order if not (len(shapes[n]) < 2 or "bias" in n or "ln" in n or "bn" in n)]
    nwd = [n for n in order if n not in wd]
    ref_params = {n: torch.nn.Parameter(ck["model"][n].clone()) for n in order}
    opt = torch.optim.AdamW([{"params": [ref_params[n] for n in wd], "weight_decay": 0.05},
                             {"params": [ref_params[n] for n in nwd], "weight_decay": 0.0}], lr=1e-4, betas=(0.9, 0.999))
    opt.load_state_dict(ck["optimizer"])
    ishapes = {n: i for n, i, _ in m.store.specs}
    steps = {"expert_adaptor": 17.0, "VETokenizer": 11.0, "VEInstructor": 9.0}
    for name in order:
        st = opt.state[ref_params[name]]
        o, n = m.store.offsets[name]
        assert float(st["step"]) == steps[name.split(".")[0]] and st["exp_avg"].shape == tuple(shapes[name])
        assert torch.equal(st["exp_avg"], to_reference_layout(m.store.flat_m[o:o + n].view(ishapes[name]), shapes[name]))
        assert torch.equal(st["exp_avg_sq"], to_reference_layout(m.store.flat_v[o:o + n].view(ishapes[name]), shapes[name]))
    # ... and a state written by torch loads back into the flat buffers
    m2 = _ToyModel()
    m2.store.flat_p.zero_(); m2.store.flat_m.zero_(); m2.store.flat_v.zero_(); m2.store.step = 0
    m2.store.set_module_steps({})
    assert C.CheckpointManager.load(m2, paths[-1]) == 3                            # resume at epoch + 1
    for a, b in ((m.store.flat_p, m2.store.flat_p), (m.store.flat_m, m2.store.flat_m), (m.store.flat_v, m2.store.flat_v)):
        assert torch.equal(a[:m.store.n_used], b[:m.store.n_used])          # past n_used: exchange padding, not parameters
    assert m2.store.module_steps() == m.store.module_steps() and m2.store.step == 17
    with pytest.raises(RuntimeError):
        C.CheckpointManager.load(m2, str(tmp_path / "nope.pth"))


def test_a_module_that_never_stepped_has_no_optimizer_state(tmp_path):
    """torch creates AdamW state lazily: a module skipped by every step so far (grad None in the reference) has no entry."""
    m = _ToyModel()
    m.store.set_module_steps({"expert_adaptor": 3, "VETokenizer": 0, "VEInstructor": 2})
    sd = C.optimizer_state_dict(m.store, lr=1e-4)
    order, _ = C._optimizer_index(m.store)
    have = {order[i] for i in sd["state"]}
    assert have == {n for n in order if not n.startswith("VETokenizer.")}

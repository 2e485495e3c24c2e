"""Seeded synthetic weights generated directly on the device (no checkpoints / network in the benchmark
environment; SURVEY 8d: linears N(0, 0.02), norm weights 1, biases 0, adapters as networks.py:78-79 / default
conv init, base_prompts N(0,1)).  A dict-like object keyed by the reference's state_dict names, materialising each
tensor on first access so a 7 B-parameter model never exists twice in memory."""
from __future__ import annotations

import math
import zlib
from typing import Dict, Tuple

import torch


def full_config(**over) -> dict:
    cfg = dict(vit_dim=1408, vit_depth=39, vit_heads=16, vit_hidden=6144, patch=14, image_size=224,
               qf_dim=768, qf_layers=12, qf_heads=12, qf_inter=3072, num_query_token=32,
               llm_dim=4096, llm_layers=32, llm_heads=32, llm_inter=11008, vocab=32000)
    cfg.update(over)
    return cfg


def shape_table(cfg: dict, arch: str = "myriad") -> Dict[str, Tuple[Tuple[int, ...], str]]:
    """name -> (shape, kind) with kind in {normal, ones, zeros, conv, base}."""
    t: Dict[str, Tuple[Tuple[int, ...], str]] = {}
    D, Hd, P = cfg["vit_dim"], cfg["vit_hidden"], cfg["patch"]
    ntok = (cfg["image_size"] // P) ** 2 + 1
    v = "visual_encoder."
    t[v + "cls_token"] = ((1, 1, D), "normal")
    t[v + "pos_embed"] = ((1, ntok, D), "normal")
    t[v + "patch_embed.proj.weight"] = ((D, 3, P, P), "normal")
    t[v + "patch_embed.proj.bias"] = ((D,), "zeros")
    for i in range(cfg["vit_depth"]):
        p = v + f"blocks.{i}."
        for n in ("norm1", "norm2"):
            t[p + n + ".weight"] = ((D,), "ones")
            t[p + n + ".bias"] = ((D,), "zeros")
        t[p + "attn.q_bias"] = ((D,), "zeros")
        t[p + "attn.v_bias"] = ((D,), "zeros")
        t[p + "attn.qkv.weight"] = ((3 * D, D), "normal")
        t[p + "attn.proj.weight"] = ((D, D), "normal")
        t[p + "attn.proj.bias"] = ((D,), "zeros")
        t[p + "mlp.fc1.weight"] = ((Hd, D), "normal")
        t[p + "mlp.fc1.bias"] = ((Hd,), "zeros")
        t[p + "mlp.fc2.weight"] = ((D, Hd), "normal")
        t[p + "mlp.fc2.bias"] = ((D,), "zeros")
    t["ln_vision.weight"] = ((D,), "ones")
    t["ln_vision.bias"] = ((D,), "zeros")
    Q, QI = cfg["qf_dim"], cfg["qf_inter"]
    t["query_tokens"] = ((1, cfg["num_query_token"], Q), "normal")
    b = "Qformer.bert."
    t[b + "embeddings.LayerNorm.weight"] = ((Q,), "ones")
    t[b + "embeddings.LayerNorm.bias"] = ((Q,), "zeros")

    def attn(p, kvw):
        for n, w in (("query", Q), ("key", kvw), ("value", kvw)):
            t[p + f"self.{n}.weight"] = ((Q, w), "normal")
            t[p + f"self.{n}.bias"] = ((Q,), "zeros")
        t[p + "output.dense.weight"] = ((Q, Q), "normal")
        t[p + "output.dense.bias"] = ((Q,), "zeros")
        t[p + "output.LayerNorm.weight"] = ((Q,), "ones")
        t[p + "output.LayerNorm.bias"] = ((Q,), "zeros")

    for i in range(cfg["qf_layers"]):
        p = b + f"encoder.layer.{i}."
        attn(p + "attention.", Q)
        if i % 2 == 0:
            attn(p + "crossattention.", D)
        t[p + "intermediate_query.dense.weight"] = ((QI, Q), "normal")
        t[p + "intermediate_query.dense.bias"] = ((QI,), "zeros")
        t[p + "output_query.dense.weight"] = ((Q, QI), "normal")
        t[p + "output_query.dense.bias"] = ((Q,), "zeros")
        t[p + "output_query.LayerNorm.weight"] = ((Q,), "ones")
        t[p + "output_query.LayerNorm.bias"] = ((Q,), "zeros")
    L, LI, V = cfg["llm_dim"], cfg["llm_inter"], cfg["vocab"]
    t["llama_proj.weight"] = ((L, Q), "normal")
    t["llama_proj.bias"] = ((L,), "zeros")
    m = "llama_model.model."
    t[m + "embed_tokens.weight"] = ((V, L), "normal")
    for i in range(cfg["llm_layers"]):
        p = m + f"layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            t[p + f"self_attn.{n}.weight"] = ((L, L), "normal")
        t[p + "mlp.gate_proj.weight"] = ((LI, L), "normal")
        t[p + "mlp.up_proj.weight"] = ((LI, L), "normal")
        t[p + "mlp.down_proj.weight"] = ((L, LI), "normal")
        t[p + "input_layernorm.weight"] = ((L,), "ones")
        t[p + "post_attention_layernorm.weight"] = ((L,), "ones")
    t[m + "norm.weight"] = ((L,), "ones")
    t["llama_model.lm_head.weight"] = ((V, L), "normal")
    if arch == "myriad":
        t["expert_adaptor.conv1.weight"] = ((4, D), "normal")
        t["expert_adaptor.conv2.weight"] = ((D, 4), "normal")
        for pre, hk, ho in (("VEInstructor.", 1, Q), ("VETokenizer.", 5, L)):
            c = 1
            for idx in (0, 3, 6, 9, 12):
                t[pre + f"meta_net.{idx}.weight"] = ((c * 4, c, 3, 3), "conv")
                t[pre + f"meta_net.{idx}.bias"] = ((c * 4,), "convb")
                c *= 4
            t[pre + "meta_net.15.weight"] = ((ho, c, hk, hk), "conv")
            t[pre + "meta_net.15.bias"] = ((ho,), "convb")
        t["VETokenizer.base_prompts"] = ((9, L), "base")
    return t


class SyntheticWeights:
    """Mapping name -> tensor, generated on `device` on access (fp32 for small tensors, bf16 for big matrices)."""

    def __init__(self, cfg: dict, device, seed: int = 0, arch: str = "myriad", big_dtype=torch.bfloat16):
        self.table = shape_table(cfg, arch)
        self.dev, self.seed, self.big = torch.device(device), seed, big_dtype

    def __contains__(self, k):
        return k in self.table

    def keys(self):
        return self.table.keys()

    def get(self, k, default=None):
        return self[k] if k in self.table else default

    def __getitem__(self, k) -> torch.Tensor:
        shape, kind = self.table[k]
        n = 1
        for s in shape:
            n *= s
        g = torch.Generator(device=self.dev).manual_seed((self.seed * 1000003 + zlib.crc32(k.encode())) & 0x7FFFFFFF)
        dt = self.big if n >= (1 << 20) else torch.float32
        if kind == "normal":
            return torch.randn(shape, generator=g, device=self.dev, dtype=dt) * 0.02
        if kind == "ones":
            return torch.ones(shape, device=self.dev)
        if kind == "zeros":
            return torch.zeros(shape, device=self.dev)
        if kind == "base":
            return torch.randn(shape, generator=g, device=self.dev, dtype=torch.float32)
        if kind in ("conv", "convb"):
            if kind == "conv":
                fan_in = shape[1] * shape[2] * shape[3]
            else:  # bias bound uses the matching weight's fan-in
                wshape = self.table[k.replace(".bias", ".weight")][0]
                fan_in = wshape[1] * wshape[2] * wshape[3]
            bound = 1.0 / math.sqrt(fan_in)
            return (torch.rand(shape, generator=g, device=self.dev, dtype=torch.float32) * 2 - 1) * bound
        raise KeyError(kind)

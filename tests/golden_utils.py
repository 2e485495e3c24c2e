"""Seeded weight / input generators shared by tools/make_golden.py (which feeds them to the
reference's own modules, in the build container only) and the tests (which feed the same
tensors to the oracle and to the HIP path).  Independent of the reference: pure torch CPU RNG,
deterministic for a given torch build (the GPU box runs the same image)."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch


def _gen(shape, g, std=0.02, kind="normal"):
    if kind == "normal":
        return torch.randn(*shape, generator=g) * std
    if kind == "ones":
        return torch.ones(*shape) + torch.randn(*shape, generator=g) * 0.1
    if kind == "small":
        return torch.randn(*shape, generator=g) * 0.05
    raise ValueError(kind)


def vit_weights(D: int, depth: int, heads: int, hidden: int, patch: int, n_tok: int, seed: int,
                prefix: str = "visual_encoder.") -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    sd[prefix + "cls_token"] = _gen((1, 1, D), g)
    sd[prefix + "pos_embed"] = _gen((1, n_tok, D), g)
    sd[prefix + "patch_embed.proj.weight"] = _gen((D, 3, patch, patch), g)
    sd[prefix + "patch_embed.proj.bias"] = _gen((D,), g, kind="small")
    for i in range(depth):
        p = prefix + f"blocks.{i}."
        sd[p + "norm1.weight"] = _gen((D,), g, kind="ones")
        sd[p + "norm1.bias"] = _gen((D,), g, kind="small")
        sd[p + "attn.q_bias"] = _gen((D,), g, kind="small")
        sd[p + "attn.v_bias"] = _gen((D,), g, kind="small")
        sd[p + "attn.qkv.weight"] = _gen((3 * D, D), g)
        sd[p + "attn.proj.weight"] = _gen((D, D), g)
        sd[p + "attn.proj.bias"] = _gen((D,), g, kind="small")
        sd[p + "norm2.weight"] = _gen((D,), g, kind="ones")
        sd[p + "norm2.bias"] = _gen((D,), g, kind="small")
        sd[p + "mlp.fc1.weight"] = _gen((hidden, D), g)
        sd[p + "mlp.fc1.bias"] = _gen((hidden,), g, kind="small")
        sd[p + "mlp.fc2.weight"] = _gen((D, hidden), g)
        sd[p + "mlp.fc2.bias"] = _gen((D,), g, kind="small")
    return sd


def qformer_weights(D: int, layers: int, inter: int, enc_w: int, seed: int, cross_freq: int = 2,
                    prefix: str = "Qformer.bert.", std: float = 0.02) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    sd[prefix + "embeddings.LayerNorm.weight"] = _gen((D,), g, kind="ones")
    sd[prefix + "embeddings.LayerNorm.bias"] = _gen((D,), g, kind="small")

    def attn(p, kv_w):
        sd[p + "self.query.weight"] = _gen((D, D), g, std)
        sd[p + "self.query.bias"] = _gen((D,), g, kind="small")
        sd[p + "self.key.weight"] = _gen((D, kv_w), g, std)
        sd[p + "self.key.bias"] = _gen((D,), g, kind="small")
        sd[p + "self.value.weight"] = _gen((D, kv_w), g, std)
        sd[p + "self.value.bias"] = _gen((D,), g, kind="small")
        sd[p + "output.dense.weight"] = _gen((D, D), g, std)
        sd[p + "output.dense.bias"] = _gen((D,), g, kind="small")
        sd[p + "output.LayerNorm.weight"] = _gen((D,), g, kind="ones")
        sd[p + "output.LayerNorm.bias"] = _gen((D,), g, kind="small")

    for i in range(layers):
        p = prefix + f"encoder.layer.{i}."
        attn(p + "attention.", D)
        if i % cross_freq == 0:
            attn(p + "crossattention.", enc_w)
        sd[p + "intermediate_query.dense.weight"] = _gen((inter, D), g, std)
        sd[p + "intermediate_query.dense.bias"] = _gen((inter,), g, kind="small")
        sd[p + "output_query.dense.weight"] = _gen((D, inter), g, std)
        sd[p + "output_query.dense.bias"] = _gen((D,), g, kind="small")
        sd[p + "output_query.LayerNorm.weight"] = _gen((D,), g, kind="ones")
        sd[p + "output_query.LayerNorm.bias"] = _gen((D,), g, kind="small")
    return sd


def llama_weights(D: int, layers: int, inter: int, vocab: int, seed: int, prefix: str = "llama_model.",
                  std: float = 0.02, lora_r: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    sd[prefix + "model.embed_tokens.weight"] = _gen((vocab, D), g, std)
    for i in range(layers):
        p = prefix + f"model.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            sd[p + f"self_attn.{n}.weight"] = _gen((D, D), g, std)
        sd[p + "mlp.gate_proj.weight"] = _gen((inter, D), g, std)
        sd[p + "mlp.down_proj.weight"] = _gen((D, inter), g, std)
        sd[p + "mlp.up_proj.weight"] = _gen((inter, D), g, std)
        sd[p + "input_layernorm.weight"] = _gen((D,), g, kind="ones")
        sd[p + "post_attention_layernorm.weight"] = _gen((D,), g, kind="ones")
        if lora_r:
            for n in ("q_proj", "v_proj"):
                sd[p + f"self_attn.{n}.lora_A.default.weight"] = _gen((lora_r, D), g, std)
                sd[p + f"self_attn.{n}.lora_B.default.weight"] = _gen((D, lora_r), g, std)
    sd[prefix + "model.norm.weight"] = _gen((D,), g, kind="ones")
    sd[prefix + "lm_head.weight"] = _gen((vocab, D), g, std)
    return sd


# ---- peaked-logit decode fixture (random-weight logits are flat; greedy ids are only comparable with margin)
# The tiny LLaMA is made a token-transition machine: embeddings of the used tokens are scaled random directions that
# dominate the residual stream, and lm_head[f(t)] carries the unit direction of embed[t], so the arg-max after token t is
# f(t) with a margin of several logits whatever the two small decoder layers add.  The chains exercise, with the eval
# script's real stop ids (evaluation_aqa_dataset.py:268-270): the min_length EOS ban on step 0 (row 0's favourite first
# token is EOS), the two-token stop [2277, 29937] on row 0 after 32 tokens, a row that finishes early with EOS and is then
# padded, and a row that emits 835 without stopping the batch (the criterion looks at row 0 only, conversation.py:102-107).
DECODE_CHAIN = dict(D=64, layers=2, heads=4, inter=172, vocab=32000, seed=601, s0=6)
DECODE_CHAINS = {
    "row0": [100] + list(range(101, 131)) + [2277, 29937],                 # start token, then the expected generation
    "row1": [200] + list(range(201, 246)),
    "row2": [300, 301, 302, 303, 2],
    "row3": [400, 401, 835, 402] + list(range(403, 440)),
    "stop835": [500, 501, 502, 835],
}


def decode_chain_weights() -> Dict[str, torch.Tensor]:
    c = DECODE_CHAIN
    D, V = c["D"], c["vocab"]
    sd = llama_weights(D, c["layers"], c["inter"], V, c["seed"], std=0.05)
    g = torch.Generator().manual_seed(c["seed"] + 1)
    emb = sd["llama_model.model.embed_tokens.weight"]
    lm = sd["llama_model.lm_head.weight"]
    used = sorted({t for ch in DECODE_CHAINS.values() for t in ch})
    dirs = {}
    for t in used:
        v = torch.randn(D, generator=g)
        dirs[t] = v / v.norm()
        emb[t] = dirs[t] * math.sqrt(D) * 2.0
    for name, ch in DECODE_CHAINS.items():
        for a, b in zip(ch[:-1], ch[1:]):
            lm[b] += 2.5 * dirs[a]      # logit ~ 20 against ~N(0, 1) for the other 32k rows: p_max ~ 1 (>= the eval's top_p 0.01)
    lm[2] += 2.5 * dirs[100]        # row 0's favourite FIRST token is EOS: banned by min_length = 1, ...
    lm[101] -= 0.75 * dirs[100]     # ... which leaves 101 (0.7 of the direction) as the clear runner-up
    return sd


def decode_chain_inputs(rows) -> torch.Tensor:
    """[B, s0, D] prompt embeddings: small noise, last position = the embedding of the row's start token."""
    c = DECODE_CHAIN
    sd = decode_chain_weights()
    g = torch.Generator().manual_seed(c["seed"] + 2)
    x = torch.randn(len(rows), c["s0"], c["D"], generator=g) * 0.3
    for i, r in enumerate(rows):
        x[i, -1] = sd["llama_model.model.embed_tokens.weight"][DECODE_CHAINS[r][0]]
    return x


# ---- peaked FULL-PIPELINE decode fixture: ViT (1 block) -> adaptor -> ln_vision -> Q-Former (2 layers,
# 81 queries) -> llama_proj + VETokenizer -> prompt_wrap -> LLaMA (1 layer, full width, V = 32000) -> greedy decode, composed
# from the reference's own modules as myriad.py:241-272,433-454 do.  Two engineered pieces make every arg-max decisive:
#   * the FIRST generated token is chosen by the IMAGE: `probe` (committed data, tests/golden/pipeline_chain.npz) holds
#     rows P_i with P_i . h_j = gamma * delta_ij for the reference's final hidden state h_j of batch row j at the last
#     prompt position (gamma x the pseudo-inverse of those four states); lm_head[first token of row i] += P_i.  The four
#     rows share one prompt, so only the image path (ViT / Q-Former / VE nets -> attention) tells them apart;
#   * from then on the LLaMA is the token-transition machine of DECODE_CHAIN above, at width 4096.
# Row 0's favourite first token is EOS (banned by min_length = 1), its chain ends in the eval script's two-token stop.
PIPELINE_CHAIN = dict(vocab=32000, vit=711, qf=712, llm=713, ad=714, glue=715, batch=716, dirs=717, gamma=24.0)
PIPELINE_CHAINS = {
    "row0": list(range(100, 131)) + [2277, 29937],
    "row1": list(range(200, 246)),
    "row2": [300, 301, 302, 303, 2],
    "row3": [400, 401, 835, 402] + list(range(403, 440)),
}


def pipeline_chain_weights(probe=None) -> Dict[str, torch.Tensor]:
    c = PIPELINE_CHAIN
    D, V = 4096, c["vocab"]
    sd = {}
    sd.update(vit_weights(1408, 1, 16, int(1408 * 4.3637), 14, 257, seed=c["vit"]))
    sd.update(qformer_weights(768, 2, 3072, 1408, seed=c["qf"]))
    sd.update(llama_weights(D, 1, 11008, V, seed=c["llm"]))
    sd.update(adapter_weights(seed=c["ad"]))
    sd.update(glue_weights(seed=c["glue"]))
    g = torch.Generator().manual_seed(c["dirs"])
    emb = sd["llama_model.model.embed_tokens.weight"]
    lm = sd["llama_model.lm_head.weight"]
    used = sorted({t for ch in PIPELINE_CHAINS.values() for t in ch})
    dirs = {}
    for t in used:
        v = torch.randn(D, generator=g)
        dirs[t] = v / v.norm()
        emb[t] = dirs[t] * math.sqrt(D) * 2.0
    for ch in PIPELINE_CHAINS.values():
        for a, b in zip(ch[:-1], ch[1:]):
            lm[b] += 2.5 * dirs[a]
    if probe is not None:
        probe = torch.as_tensor(probe, dtype=torch.float32)
        for i, name in enumerate(("row0", "row1", "row2", "row3")):
            lm[PIPELINE_CHAINS[name][0]] += probe[i]
        lm[2] += 1.25 * probe[0]          # EOS is row 0's favourite first token; min_length = 1 bans it
    return sd


def pipeline_chain_batch():
    """4 images N(0,1); 4 anomaly maps with a bump at a row-specific place (noise maps pool to the same statistics in
    every row); one shared prompt (4 ids before / 28 after <ImageHere>), as the eval script's fixed question gives."""
    c = PIPELINE_CHAIN
    g = torch.Generator().manual_seed(c["batch"])
    image = torch.randn(4, 3, 224, 224, generator=g)
    yy, xx = torch.meshgrid(torch.arange(224.0), torch.arange(224.0), indexing="ij")
    maps = torch.rand(4, 1, 224, 224, generator=g) * 0.1
    for i, (cy, cx, s) in enumerate(((40, 50, 18.0), (170, 60, 30.0), (112, 112, 45.0), (60, 180, 12.0))):
        maps[i, 0] += 0.9 * torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
    maps = maps.clamp_(0, 1)
    before = torch.randint(3000, 29000, (1, 4), generator=g).expand(4, -1).contiguous()      # clear of the chain ids
    after = torch.randint(3000, 29000, (1, 28), generator=g).expand(4, -1).contiguous()
    return image, maps, before, after


def ve_stem_weights(prefix: str, g, sd, dim_in: int = 1):
    c = dim_in
    for idx in (0, 3, 6, 9, 12):
        co = c * 4
        bound = 1.0 / math.sqrt(c * 9)
        sd[prefix + f"meta_net.{idx}.weight"] = (torch.rand(co, c, 3, 3, generator=g) * 2 - 1) * bound
        sd[prefix + f"meta_net.{idx}.bias"] = (torch.rand(co, generator=g) * 2 - 1) * bound
        c = co
    return c


def adapter_weights(seed: int, D_vit: int = 1408, rank: int = 4, d_q: int = 768, d_llm: int = 4096,
                    with_tokenizer: bool = True, with_instructor: bool = True) -> Dict[str, torch.Tensor]:
    """expert_adaptor + VEInstructor + VETokenizer weights (reference init distributions:
    networks.py:78-79 N(0,.02); conv default kaiming-uniform bound 1/sqrt(fan_in); base_prompts N(0,1))."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    sd["expert_adaptor.conv1.weight"] = _gen((rank, D_vit), g)
    sd["expert_adaptor.conv2.weight"] = _gen((D_vit, rank), g)
    if with_instructor:
        c = ve_stem_weights("VEInstructor.", g, sd)
        bound = 1.0 / math.sqrt(c)
        sd["VEInstructor.meta_net.15.weight"] = (torch.rand(d_q, c, 1, 1, generator=g) * 2 - 1) * bound
        sd["VEInstructor.meta_net.15.bias"] = (torch.rand(d_q, generator=g) * 2 - 1) * bound
    if with_tokenizer:
        c = ve_stem_weights("VETokenizer.", g, sd)
        bound = 1.0 / math.sqrt(c * 25)
        sd["VETokenizer.meta_net.15.weight"] = (torch.rand(d_llm, c, 5, 5, generator=g) * 2 - 1) * bound
        sd["VETokenizer.meta_net.15.bias"] = (torch.rand(d_llm, generator=g) * 2 - 1) * bound
        sd["VETokenizer.base_prompts"] = torch.randn(9, d_llm, generator=g)
    return sd


def glue_weights(seed: int, n_query: int = 32, d_q: int = 768, D_vit: int = 1408, d_llm: int = 4096):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    sd["query_tokens"] = _gen((1, n_query, d_q), g)
    sd["ln_vision.weight"] = _gen((D_vit,), g, kind="ones")
    sd["ln_vision.bias"] = _gen((D_vit,), g, kind="small")
    sd["llama_proj.weight"] = _gen((d_llm, d_q), g)
    sd["llama_proj.bias"] = _gen((d_llm,), g, kind="small")
    return sd


def synthetic_batch(B: int, vocab: int, seed: int, n_before: int = 4, n_after: int = 28, n_tgt: int = 16,
                    img: int = 224, pad_tail: int = 0):
    """SURVEY 8(d) synthetic inputs: image N(0,1), maps U[0,1), ids uniform in [3, V)."""
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(B, 3, img, img, generator=g)
    maps = torch.rand(B, 1, 224, 224, generator=g)
    before = torch.randint(3, vocab, (B, n_before), generator=g)
    after = torch.randint(3, vocab, (B, n_after), generator=g)
    # prompts are identical across the batch in the reference (one fixed question string)
    before = before[:1].expand(B, -1).contiguous()
    after = after[:1].expand(B, -1).contiguous()
    tgt = torch.randint(3, vocab, (B, n_tgt), generator=g)
    mask = torch.ones(B, n_tgt, dtype=torch.long)
    if pad_tail:
        # ragged targets: row b loses (b % (pad_tail+1)) trailing tokens (right padding, pad id 2)
        for b in range(B):
            k = b % (pad_tail + 1)
            if k:
                tgt[b, -k:] = 2
                mask[b, -k:] = 0
    return image, maps, before, after, tgt, mask


def expert_weights(D: int, blocks: int, C: int, n_taps: int, seed: int, n_tok: int = 257) -> Dict[str, torch.Tensor]:
    """ImageBind vision-branch weights under the reference's state_dict names (imagebind_model.py) plus the anomaly
    expert's per-tap decoder `image_decoder.fc.{i}` (adrefexpert_v2.py:16-29, 109)."""
    g = torch.Generator().manual_seed(seed)
    pre, trk, head = "modality_preprocessors.vision.", "modality_trunks.vision.", "modality_heads.vision."
    sd = {pre + "cls_token": _gen((1, 1, D), g), pre + "rgbt_stem.proj.1.weight": _gen((D, 3, 2, 14, 14), g),
          pre + "pos_embedding_helper.pos_embed": _gen((1, n_tok, D), g),
          trk + "pre_transformer_layer.0.weight": _gen((D,), g, kind="ones"), trk + "pre_transformer_layer.0.bias": _gen((D,), g, kind="small")}
    for i in range(blocks):
        p = f"{trk}blocks.{i}."
        sd[p + "attn.in_proj_weight"] = _gen((3 * D, D), g)
        sd[p + "attn.in_proj_bias"] = _gen((3 * D,), g, kind="small")
        sd[p + "attn.out_proj.weight"] = _gen((D, D), g)
        sd[p + "attn.out_proj.bias"] = _gen((D,), g, kind="small")
        sd[p + "norm_1.weight"] = _gen((D,), g, kind="ones")
        sd[p + "norm_1.bias"] = _gen((D,), g, kind="small")
        sd[p + "norm_2.weight"] = _gen((D,), g, kind="ones")
        sd[p + "norm_2.bias"] = _gen((D,), g, kind="small")
        sd[p + "mlp.fc1.weight"] = _gen((4 * D, D), g)
        sd[p + "mlp.fc1.bias"] = _gen((4 * D,), g, kind="small")
        sd[p + "mlp.fc2.weight"] = _gen((D, 4 * D), g)
        sd[p + "mlp.fc2.bias"] = _gen((D,), g, kind="small")
    sd[head + "0.weight"] = _gen((D,), g, kind="ones")
    sd[head + "0.bias"] = _gen((D,), g, kind="small")
    sd[head + "2.weight"] = _gen((C, D), g)
    for i in range(n_taps):
        sd[f"image_decoder.fc.{i}.weight"] = _gen((C, D), g, std=0.05)
        sd[f"image_decoder.fc.{i}.bias"] = _gen((C,), g, kind="small")
    return sd


def expert_inputs(B: int, k: int, C: int, seed: int):
    """(images [B,3,224,224], reference images [B*k,3,224,224], text features [B,2,C] L2-normalised)."""
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(B, 3, 224, 224, generator=g)
    refs = torch.randn(B * k, 3, 224, 224, generator=g)
    text = torch.randn(B, 2, C, generator=g)
    return images, refs, text / text.norm(dim=-1, keepdim=True)


def expert_text_weights(D: int, blocks: int, C: int, seed: int, vocab: int = 49408, ctx: int = 77) -> Dict[str, torch.Tensor]:
    """ImageBind text-branch weights under the reference's state_dict names (imagebind_model.py text modality)."""
    g = torch.Generator().manual_seed(seed)
    pre, trk, head = "modality_preprocessors.text.", "modality_trunks.text.", "modality_heads.text."
    sd = {pre + "token_embedding.weight": _gen((vocab, D), g), pre + "pos_embed": _gen((1, ctx, D), g, std=0.01),
          pre + "mask": torch.full((ctx, ctx), float("-inf")).triu_(1)}
    for i in range(blocks):
        p = f"{trk}blocks.{i}."
        sd[p + "attn.in_proj_weight"] = _gen((3 * D, D), g, std=0.05)
        sd[p + "attn.in_proj_bias"] = _gen((3 * D,), g, kind="small")
        sd[p + "attn.out_proj.weight"] = _gen((D, D), g, std=0.05)
        sd[p + "attn.out_proj.bias"] = _gen((D,), g, kind="small")
        for n_ in ("norm_1", "norm_2"):
            sd[p + n_ + ".weight"] = _gen((D,), g, kind="ones")
            sd[p + n_ + ".bias"] = _gen((D,), g, kind="small")
        sd[p + "mlp.fc1.weight"] = _gen((4 * D, D), g, std=0.05)
        sd[p + "mlp.fc1.bias"] = _gen((4 * D,), g, kind="small")
        sd[p + "mlp.fc2.weight"] = _gen((D, 4 * D), g, std=0.05)
        sd[p + "mlp.fc2.bias"] = _gen((D,), g, kind="small")
    sd[head + "proj.0.weight"] = _gen((D,), g, kind="ones")
    sd[head + "proj.0.bias"] = _gen((D,), g, kind="small")
    sd[head + "proj.1.weight"] = _gen((C, D), g, std=0.05)
    sd["modality_postprocessors.text.1.log_logit_scale"] = torch.tensor(math.log(1 / 0.07))
    return sd


def expert_prompt_ids(n_obj: int, n_normal: int, n_abnormal: int, seed: int, vocab: int = 49408, ctx: int = 77):
    """Token ids shaped like CLIP-tokenised prompt sentences: SOT, a few word ids, EOT (= the largest id), zero padding.
    Returns (normal [n_obj*n_normal, ctx], abnormal [n_obj*n_abnormal, ctx]) int64."""
    g = torch.Generator().manual_seed(seed)

    def make(n):
        ids = torch.zeros(n, ctx, dtype=torch.long)
        for r in range(n):
            ln = int(torch.randint(4, 14, (1,), generator=g))
            ids[r, 0] = vocab - 2
            ids[r, 1:1 + ln] = torch.randint(1, vocab - 2, (ln,), generator=g)
            ids[r, 1 + ln] = vocab - 1
        return ids

    return make(n_obj * n_normal), make(n_obj * n_abnormal)


# ---- image front-end (image side): name -> (H, W, size, mode, seed); inputs are regenerated from the seed
IMAGE_CASES = {
    "down3_landscape": (97, 131, 32, "train", 11),     # 3x minification, wider than tall, odd sizes
    "down_portrait": (150, 101, 32, "train", 12),      # taller than wide
    "upscale": (20, 27, 32, "train", 13),              # magnification (support not scaled)
    "square_same": (32, 32, 32, "train", 14),          # nothing to resample: both passes are identities
    "one_axis": (64, 32, 32, "train", 15),             # only the vertical axis changes size
    "eval_aniso": (75, 121, 32, "eval", 16),           # Resize((S, S)): different scale per axis, no crop
    "full_224": (260, 346, 224, "train", 17),          # the shipped size on a small input
}


def image_case(H: int, W: int, seed: int):
    """Seeded uint8 RGB image: smooth gradients + blocks + noise, so that clipping (over/undershoot of the cubic) occurs."""
    import numpy as np
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([(xx * 255 // max(W - 1, 1)), (yy * 255 // max(H - 1, 1)), ((xx + yy) * 7 % 256)], -1).astype(np.int64)
    img += rng.integers(-40, 41, (H, W, 3))
    blocks = rng.integers(0, 2, ((H + 7) // 8, (W + 7) // 8, 1)).repeat(8, 0).repeat(8, 1)[:H, :W]
    img = np.where(blocks > 0, img, 255 - img)
    hard = rng.integers(0, 256, (H, W, 3))
    mask = rng.random((H, W, 1)) < 0.05
    return np.clip(np.where(mask, hard, img), 0, 255).astype(np.uint8)

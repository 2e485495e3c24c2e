"""CPU fp32 restatement of the Myriad / MiniGPT-4 hot path (TEST INFRASTRUCTURE).

Every function is a plain functional torch-CPU restatement of one reference
symbol and cites the reference ``file:line`` it follows (paths relative to the
reference checkout).  Weights are passed as a flat ``dict`` keyed by the
reference's own ``state_dict`` names, so a reference module's ``state_dict()``
can be fed in unchanged -- that is how ``tests/test_oracle_golden.py`` pins this
file against vectors produced by the reference's modules.

Gradients come from torch autograd on these fp32 functions (the reference
itself relies on autograd; there is no hand-written backward to restate).

This file must never be imported by ``myriad_amd`` (the product).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------- #
# a-1 .. a-4  EVA ViT-g                                                        #
# --------------------------------------------------------------------------- #
def vit_attention(sd: SD, p: str, x: Tensor, num_heads: int,
                  rel_pos_bias: Optional[Tensor] = None) -> Tensor:
    """`Attention.forward`, minigpt4/models/eva_vit.py:118-148.

    qkv bias = cat(q_bias, zeros, v_bias) (:120-124); q scaled BEFORE q@k^T
    (:128-129); optional additive rel_pos_bias (:139-140).
    """
    B, N, C = x.shape
    w = sd[p + "qkv.weight"]
    bias = None
    if (p + "q_bias") in sd:
        qb, vb = sd[p + "q_bias"], sd[p + "v_bias"]
        bias = torch.cat((qb, torch.zeros_like(vb), vb))
    qkv = F.linear(x, w, bias).reshape(B, N, 3, num_heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    head_dim = q.shape[-1]
    q = q * (head_dim ** -0.5)
    attn = q @ k.transpose(-2, -1)
    if rel_pos_bias is not None:
        attn = attn + rel_pos_bias
    attn = attn.softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B, N, -1)
    return F.linear(out, sd[p + "proj.weight"], sd[p + "proj.bias"])


def vit_block(sd: SD, p: str, x: Tensor, num_heads: int, eps: float,
              rel_pos_bias: Optional[Tensor] = None) -> Tensor:
    """`Block.forward` (gamma_1 is None for EVA-g), eva_vit.py:173-180; `Mlp` :54-61."""
    h = F.layer_norm(x, x.shape[-1:], sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    x = x + vit_attention(sd, p + "attn.", h, num_heads, rel_pos_bias)
    h = F.layer_norm(x, x.shape[-1:], sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
    h = F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    h = F.gelu(h)  # nn.GELU() = erf form
    h = F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + h


def vit_forward(sd: SD, image: Tensor, num_heads: int = 16, eps: float = 1e-6,
                prefix: str = "visual_encoder.",
                rel_pos_bias: Optional[Tensor] = None) -> Tensor:
    """`VisionTransformer.forward_features`, eva_vit.py:324-340 (no final norm/head)
    with `PatchEmbed.forward` :198-204 (conv k=stride=patch)."""
    w = sd[prefix + "patch_embed.proj.weight"]
    patch = w.shape[-1]
    x = F.conv2d(image, w, sd[prefix + "patch_embed.proj.bias"], stride=patch)
    x = x.flatten(2).transpose(1, 2)
    cls = sd[prefix + "cls_token"].expand(x.shape[0], -1, -1)
    x = torch.cat((cls, x), dim=1)
    if (prefix + "pos_embed") in sd:
        x = x + sd[prefix + "pos_embed"]
    depth = 0
    while (prefix + f"blocks.{depth}.norm1.weight") in sd:
        depth += 1
    for i in range(depth):
        x = vit_block(sd, prefix + f"blocks.{i}.", x, num_heads, eps, rel_pos_bias)
    return x


# --------------------------------------------------------------------------- #
# a-5 .. a-7  adapters                                                         #
# --------------------------------------------------------------------------- #
def lora_adaptor(sd: SD, x: Tensor, prefix: str = "expert_adaptor.") -> Tensor:
    """`LoraAdaptorV2.forward`, minigpt4/models/networks.py:81-93: x + conv2(conv1(x))."""
    return x + F.linear(F.linear(x, sd[prefix + "conv1.weight"]), sd[prefix + "conv2.weight"])


def ln_vision(sd: SD, x: Tensor, prefix: str = "ln_vision.") -> Tensor:
    """`LayerNorm.forward` (fp32 LN, default eps 1e-5), minigpt4/models/blip2.py:119-125."""
    return F.layer_norm(x.float(), x.shape[-1:], sd[prefix + "weight"], sd[prefix + "bias"], 1e-5)


def ve_stem(sd: SD, maps: Tensor, prefix: str) -> Tensor:
    """Shared 5 x (conv3x3 pad1 -> ReLU -> maxpool2) stem, networks.py:98-122 / :159-183."""
    x = maps
    for idx in (0, 3, 6, 9, 12):
        x = F.conv2d(x, sd[prefix + f"meta_net.{idx}.weight"], sd[prefix + f"meta_net.{idx}.bias"], padding=1)
        x = F.max_pool2d(F.relu(x), 2)
    return x


def ve_instructor(sd: SD, maps: Tensor, prefix: str = "VEInstructor.") -> Tensor:
    """`VEInstructorV2.forward` (version 0), networks.py:149-153: stem -> conv1x1(1024->768)
    -> reshape(B,768,49).transpose."""
    B = maps.shape[0]
    x = ve_stem(sd, maps, prefix)
    x = F.conv2d(x, sd[prefix + "meta_net.15.weight"], sd[prefix + "meta_net.15.bias"])
    return x.reshape(B, x.shape[1], -1).transpose(-2, -1)


def ve_tokenizer(sd: SD, maps: Tensor, prefix: str = "VETokenizer.") -> Tensor:
    """`VETokenizer.forward`, networks.py:191-197: stem -> conv5x5 valid (1024->4096) ->
    reshape(B,4096,9).transpose, prefixed by the 9 learned base_prompts."""
    B = maps.shape[0]
    x = ve_stem(sd, maps, prefix)
    x = F.conv2d(x, sd[prefix + "meta_net.15.weight"], sd[prefix + "meta_net.15.bias"])
    x = x.reshape(B, x.shape[1], -1).transpose(-2, -1)
    base = sd[prefix + "base_prompts"]
    return torch.cat([base.expand(B, -1, -1), x], dim=1)


# --------------------------------------------------------------------------- #
# a-8  Q-Former (BLIP-2 BERT with query tokens + cross-attention)              #
# --------------------------------------------------------------------------- #
def _bert_attention(sd: SD, p: str, hidden: Tensor, kv_src: Tensor, heads: int, eps: float) -> Tensor:
    """`BertSelfAttention.forward` Qformer.py:169-275 (scores / sqrt(d) AFTER q@k^T :244,
    additive mask is all-zero here) + `BertSelfOutput.forward` :285-289 (dense, LN(x+res))."""
    B, n, D = hidden.shape
    d = D // heads

    def split(t):
        return t.view(B, -1, heads, d).permute(0, 2, 1, 3)

    q = split(F.linear(hidden, sd[p + "self.query.weight"], sd[p + "self.query.bias"]))
    k = split(F.linear(kv_src, sd[p + "self.key.weight"], sd[p + "self.key.bias"]))
    v = split(F.linear(kv_src, sd[p + "self.value.weight"], sd[p + "self.value.bias"]))
    scores = (q @ k.transpose(-1, -2)) / math.sqrt(d)
    probs = scores.softmax(dim=-1)
    ctx = (probs @ v).permute(0, 2, 1, 3).reshape(B, n, D)
    out = F.linear(ctx, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
    return F.layer_norm(out + hidden, (D,), sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)


def qformer_forward(sd: SD, query_embeds: Tensor, enc: Tensor, heads: int = 12,
                    eps: float = 1e-12, cross_freq: int = 2,
                    prefix: str = "Qformer.bert.") -> Tensor:
    """`BertModel.forward` Qformer.py:804-965 on the query-only path used by
    `Myriad.encode_img` (myriad.py:256-261): `BertEmbeddings` LN of the raw queries
    (:104-107), then per `BertLayer.forward` :402-474 self-attn, cross-attn on layers
    with layer_num % cross_freq == 0 (:386-395), `feed_forward_chunk_query` :481-484.
    Dropout inactive (module frozen in eval, myriad.py:163-164)."""
    D = query_embeds.shape[-1]
    h = F.layer_norm(query_embeds, (D,), sd[prefix + "embeddings.LayerNorm.weight"],
                     sd[prefix + "embeddings.LayerNorm.bias"], eps)
    n_layers = 0
    while (prefix + f"encoder.layer.{n_layers}.attention.self.query.weight") in sd:
        n_layers += 1
    for i in range(n_layers):
        p = prefix + f"encoder.layer.{i}."
        h = _bert_attention(sd, p + "attention.", h, h, heads, eps)
        if i % cross_freq == 0:
            h = _bert_attention(sd, p + "crossattention.", h, enc, heads, eps)
        inter = F.gelu(F.linear(h, sd[p + "intermediate_query.dense.weight"], sd[p + "intermediate_query.dense.bias"]))
        out = F.linear(inter, sd[p + "output_query.dense.weight"], sd[p + "output_query.dense.bias"])
        h = F.layer_norm(out + h, (D,), sd[p + "output_query.LayerNorm.weight"],
                         sd[p + "output_query.LayerNorm.bias"], eps)
    return h


# --------------------------------------------------------------------------- #
# a-11  LLaMA                                                                  #
# --------------------------------------------------------------------------- #
def rms_norm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """`LlamaRMSNorm.forward`, modeling_llama.py:66-74."""
    var = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def rotary_tables(dim: int, n_pos: int, base: float = 10000.0) -> Tuple[Tensor, Tensor]:
    """`LlamaRotaryEmbedding.__init__`, modeling_llama.py:78-91: emb = cat(freqs, freqs)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
    t = torch.arange(n_pos, dtype=inv_freq.dtype)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def apply_rotary(q: Tensor, k: Tensor, cos: Tensor, sin: Tensor, position_ids: Tensor):
    """`apply_rotary_pos_emb` + `rotate_half`, modeling_llama.py:109-123.  q,k [B,H,S,d]."""
    c = cos[position_ids][:, None]  # [B,1,S,d]
    s = sin[position_ids][:, None]

    def rot(x):
        h = x.shape[-1] // 2
        return torch.cat((-x[..., h:], x[..., :h]), dim=-1)

    return q * c + rot(q) * s, k * c + rot(k) * s


def decoder_mask(attention_mask: Tensor, q_len: int, past_len: int, dtype=torch.float32) -> Tensor:
    """`_make_causal_mask` + `_expand_mask` + `_prepare_decoder_attention_mask`,
    modeling_llama.py:25-54,442-463 (additive finfo.min masks, summed)."""
    B, kv_len = attention_mask.shape
    fmin = torch.finfo(dtype).min
    m = None
    if q_len > 1:
        c = torch.full((q_len, q_len), fmin, dtype=dtype)
        cond = torch.arange(q_len)
        c.masked_fill_(cond < (cond + 1).view(q_len, 1), 0)
        if past_len > 0:
            c = torch.cat([torch.zeros(q_len, past_len, dtype=dtype), c], dim=-1)
        m = c[None, None].expand(B, 1, q_len, kv_len)
    inv = 1.0 - attention_mask[:, None, None, :].expand(B, 1, q_len, kv_len).to(dtype)
    e = inv.masked_fill(inv.to(torch.bool), fmin)
    return e if m is None else e + m


def lora_linear(x: Tensor, W: Tensor, A: Tensor, Bm: Tensor, alpha: float, r: int,
                dropout_mask: Optional[Tensor] = None) -> Tensor:
    """peft's published `lora.Linear.forward` for the wrap at myriad.py:170-180 (PARITY UNPINNED by the reference: peft
    is un-vendored and absent; pinned instead by the hand-computed known-answer vectors in
    tests/test_oracle_golden.py::test_peft_lora_formula_known_answer):
        y = x W^T + (alpha / r) * (drop(x) A^T) B^T,      drop(x) = x * keep / (1 - p)   (nn.Dropout, training mode)
    `dropout_mask` holds the keep/(1-p) factors (None = eval mode / p = 0)."""
    xin = x if dropout_mask is None else x * dropout_mask
    return F.linear(x, W) + (alpha / r) * F.linear(F.linear(xin, A), Bm)


def llama_layer(sd: SD, p: str, x: Tensor, mask: Tensor, position_ids: Tensor, heads: int,
                eps: float, cos: Tensor, sin: Tensor, past=None, lora: Optional[dict] = None):
    """`LlamaDecoderLayer.forward` modeling_llama.py:247-299 with `LlamaAttention.forward`
    :168-231 (scores / sqrt(d) after q@k^T :197; + mask then max(., finfo.min) :210-211;
    fp32 softmax :214) and `LlamaMLP.forward` :139-140.

    `lora` (row a-14, PARITY UNPINNED -- peft is un-vendored): dict(r, alpha, dropout_mask)
    applies peft's published LoRA form y = W x + (alpha/r) * B(A(drop(x))) on q_proj and
    v_proj (target_modules at myriad.py:171-178).  Each wrapped Linear owns its nn.Dropout, so
    `dropout_mask` is a dict {"q_proj": keep/(1-p) factors, "v_proj": ...} (or one tensor for both)."""
    B, S, D = x.shape
    d = D // heads
    h = rms_norm(x, sd[p + "input_layernorm.weight"], eps)

    def proj(name, inp):
        W = sd[p + f"self_attn.{name}.weight"]
        ka = p + f"self_attn.{name}.lora_A.default.weight"
        if lora is None or ka not in sd:
            return F.linear(inp, W)
        dm = lora.get("dropout_mask")
        if isinstance(dm, dict):
            dm = dm[name]
        return lora_linear(inp, W, sd[ka], sd[p + f"self_attn.{name}.lora_B.default.weight"], lora["alpha"], lora["r"], dm)

    q = proj("q_proj", h).view(B, S, heads, d).transpose(1, 2)
    k = proj("k_proj", h).view(B, S, heads, d).transpose(1, 2)
    v = proj("v_proj", h).view(B, S, heads, d).transpose(1, 2)
    q, k = apply_rotary(q, k, cos, sin, position_ids)
    if past is not None:
        k = torch.cat([past[0], k], dim=2)
        v = torch.cat([past[1], v], dim=2)
    present = (k, v)
    w = (q @ k.transpose(2, 3)) / math.sqrt(d)
    w = w + mask
    w = torch.max(w, torch.tensor(torch.finfo(w.dtype).min))
    w = F.softmax(w, dim=-1, dtype=torch.float32)
    o = (w @ v).transpose(1, 2).reshape(B, S, D)
    x = x + F.linear(o, sd[p + "self_attn.o_proj.weight"])
    h = rms_norm(x, sd[p + "post_attention_layernorm.weight"], eps)
    g = F.linear(h, sd[p + "mlp.gate_proj.weight"])
    u = F.linear(h, sd[p + "mlp.up_proj.weight"])
    x = x + F.linear(F.silu(g) * u, sd[p + "mlp.down_proj.weight"])
    return x, present


def llama_model(sd: SD, inputs_embeds: Tensor, attention_mask: Optional[Tensor], heads: int,
                eps: float = 1e-6, position_ids: Optional[Tensor] = None, past=None,
                prefix: str = "llama_model.model.", lora: Optional[dict] = None,
                max_pos: int = 2048):
    """`LlamaModel.forward`, modeling_llama.py:466-596. Returns (normed hidden, kv list)."""
    B, S, D = inputs_embeds.shape
    past_len = 0 if past is None else past[0][0].shape[2]
    if position_ids is None:
        position_ids = torch.arange(past_len, S + past_len).unsqueeze(0).expand(B, -1)
    if attention_mask is None:
        attention_mask = torch.ones(B, S + past_len)
    mask = decoder_mask(attention_mask.float(), S, past_len, inputs_embeds.dtype)
    cos, sin = rotary_tables(D // heads, max_pos)
    n_layers = 0
    while (prefix + f"layers.{n_layers}.input_layernorm.weight") in sd:
        n_layers += 1
    x = inputs_embeds
    kv = []
    for i in range(n_layers):
        x, pres = llama_layer(sd, prefix + f"layers.{i}.", x, mask, position_ids, heads, eps, cos, sin,
                              None if past is None else past[i], lora)
        kv.append(pres)
    return rms_norm(x, sd[prefix + "norm.weight"], eps), kv


def clamp_ce_loss(logits: Tensor, target: Tensor) -> Tensor:
    """`LlamaForCausalLM.clamp_CE_loss`, modeling_llama.py:718-728:
    softmax -> clamp(1e-7, 1-1e-7) -> log -> NLLLoss (mean over target != -100)."""
    p = torch.clamp(F.softmax(logits, dim=1), min=1e-7, max=1 - 1e-7)
    return F.nll_loss(torch.log(p), target)


def llama_causal_lm(sd: SD, inputs_embeds: Tensor, attention_mask: Tensor, labels: Optional[Tensor],
                    heads: int, eps: float = 1e-6, prefix: str = "llama_model.",
                    lora: Optional[dict] = None):
    """`LlamaForCausalLM.forward`, modeling_llama.py:629-716 (shift :695-703)."""
    hidden, _ = llama_model(sd, inputs_embeds, attention_mask, heads, eps, prefix=prefix + "model.", lora=lora)
    logits = F.linear(hidden, sd[prefix + "lm_head.weight"])
    loss = None
    if labels is not None:
        V = logits.shape[-1]
        loss = clamp_ce_loss(logits[..., :-1, :].reshape(-1, V), labels[..., 1:].reshape(-1))
    return loss, logits


# --------------------------------------------------------------------------- #
# a-9, a-10  projection + prompt wrap + label/mask assembly                    #
# --------------------------------------------------------------------------- #
def encode_img(sd: SD, image: Tensor, maps: Optional[Tensor], stage: int, arch: str = "myriad",
               vit_heads: int = 16, qf_heads: int = 12) -> Tensor:
    """`Myriad.encode_img` myriad.py:241-272 / `MiniGPT4.encode_img` mini_gpt4.py:153-181.
    arch='mini_gpt4': ViT -> ln_vision -> Q-Former(32 queries) -> llama_proj.
    arch='myriad'   : ViT -> expert_adaptor -> ln_vision ; queries += VEInstructor(maps) if
    stage in {1,2} ; Q-Former ; llama_proj ; cat VETokenizer(maps) if stage in {0,1}."""
    x = vit_forward(sd, image, vit_heads)
    if arch == "myriad":
        x = lora_adaptor(sd, x)
    x = ln_vision(sd, x)
    q = sd["query_tokens"].expand(x.shape[0], -1, -1)
    if arch == "myriad" and stage in (1, 2):
        q = torch.cat([q, ve_instructor(sd, maps)], 1)
    qo = qformer_forward(sd, q, x, qf_heads)
    out = F.linear(qo, sd["llama_proj.weight"], sd["llama_proj.bias"])
    if arch == "myriad" and stage in (0, 1):
        out = torch.cat([out, ve_tokenizer(sd, maps)], 1)
    return out


def assemble_inputs(embed_w: Tensor, img_embeds: Tensor, before_ids: Tensor, after_ids: Tensor,
                    target_ids: Tensor, target_mask: Tensor, bos_id: int, pad_id: int):
    """`Myriad.prompt_wrap` myriad.py:354-375 + target/label/mask assembly :395-421, with the
    tokenizer outputs supplied as integer ids (no tokenizer model ships with the reference).
    before_ids/after_ids [B,nb]/[B,na]; target_ids [B,T] right-padded with pad_id,
    target_mask [B,T] (tokenizer attention_mask).  Returns (inputs_embeds, attention_mask, labels)."""
    B = img_embeds.shape[0]
    wrapped = torch.cat([embed_w[before_ids], img_embeds, embed_w[after_ids]], dim=1)
    targets = target_ids.masked_fill(target_ids == pad_id, -100)
    empty = torch.full((B, wrapped.shape[1] + 1), -100, dtype=torch.long)
    labels = torch.cat([empty, targets], dim=1)
    bos = embed_w[torch.full((B, 1), bos_id, dtype=torch.long)]
    inputs_embeds = torch.cat([bos, wrapped, embed_w[target_ids]], dim=1)
    attn = torch.cat([torch.ones(B, 1 + wrapped.shape[1], dtype=torch.long), target_mask.long()], dim=1)
    return inputs_embeds, attn, labels


def model_forward(sd: SD, image: Tensor, maps: Optional[Tensor], stage: int, before_ids: Tensor,
                  after_ids: Tensor, target_ids: Tensor, target_mask: Tensor, arch: str = "myriad",
                  vit_heads: int = 16, qf_heads: int = 12, llm_heads: int = 32, bos_id: int = 1,
                  pad_id: int = 2, lora: Optional[dict] = None) -> Tensor:
    """`Myriad.forward` myriad.py:377-431 / `MiniGPT4.forward` mini_gpt4.py:203-257 with the two
    `random.choice` draws (stage; zero-shot vs one-shot maps) made explicit arguments."""
    img = encode_img(sd, image, maps, stage, arch, vit_heads, qf_heads)
    ew = sd[_embed_key(sd)]
    emb, attn, labels = assemble_inputs(ew, img, before_ids, after_ids, target_ids, target_mask, bos_id, pad_id)
    loss, _ = llama_causal_lm(sd, emb, attn, labels, llm_heads, lora=lora)
    return loss


def _embed_key(sd: SD) -> str:
    return "llama_model.model.embed_tokens.weight"


# --------------------------------------------------------------------------- #
# a-12  generation                                                             #
# --------------------------------------------------------------------------- #
def greedy_generate(sd: SD, inputs_embeds: Tensor, heads: int, max_new_tokens: int = 90,
                    stop_ids: Sequence[Sequence[int]] = ((835,), (2277, 29937)), eos_id: int = 2,
                    min_length: int = 1, eps: float = 1e-6, prefix: str = "llama_model.",
                    return_margins: bool = False, return_scales: bool = False):
    """`Myriad.generate` myriad.py:447-450 -> HF sample loop with do_sample, top_p=0.01, T=1
    (evaluation_aqa_dataset.py:289-301), restated as arg-max (== top-p 0.01 sampling whenever
    p_max >= 0.01), `prepare_inputs_for_generation` modeling_llama.py:730-760
    (position = cumsum(mask)-1 with an all-ones mask), KV cache concat :190-195, `min_length`
    (EOS banned while generated length < min_length), `StoppingCriteriaSub.__call__`
    conversation.py:102-107 (stop when ROW 0 ends with a stop sequence).
    Returns LongTensor [B, T_generated] (generated ids only)."""
    B, S0, _ = inputs_embeds.shape
    ew = sd[prefix + "model.embed_tokens.weight"]
    lm = sd[prefix + "lm_head.weight"]
    past = None
    x = inputs_embeds
    out: List[Tensor] = []
    margins, scales = [], []
    unfinished = torch.ones(B, dtype=torch.long)
    total = S0
    for step in range(max_new_tokens):
        pos = None
        if past is not None:
            pos = torch.full((B, 1), total - 1, dtype=torch.long)
        hidden, past = llama_model(sd, x, torch.ones(B, total), heads, eps, position_ids=pos, past=past,
                                   prefix=prefix + "model.")
        logits = F.linear(hidden[:, -1], lm)
        if step < min_length:
            logits[:, eos_id] = -float("inf")
        top2 = logits.topk(2, dim=-1).values
        margins.append((top2[:, 0] - top2[:, 1]))
        scales.append(logits[torch.isfinite(logits).all(-1)].abs().amax(-1) if torch.isfinite(logits).all() else
                      torch.where(torch.isfinite(logits), logits, torch.zeros_like(logits)).abs().amax(-1))
        nxt = logits.argmax(-1)
        nxt = nxt * unfinished + eos_id * (1 - unfinished)  # HF pads finished rows with pad(=eos)
        unfinished = unfinished * (nxt != eos_id).long()
        out.append(nxt)
        ids0 = [int(t[0]) for t in out]
        if any(len(ids0) >= len(s) and ids0[-len(s):] == list(s) for s in stop_ids):
            break
        if int(unfinished.max()) == 0:
            break
        x = ew[nxt][:, None]
        total += 1
    ids = torch.stack(out, dim=1)
    if return_scales:          # + the largest |logit| of every step: the scale a bf16 ulp is measured against
        return ids, torch.stack(margins, dim=1), torch.stack(scales, dim=1)
    if return_margins:
        return ids, torch.stack(margins, dim=1)
    return ids


# --------------------------------------------------------------------------- #
# a-13  optimiser / schedule                                                   #
# --------------------------------------------------------------------------- #
def cosine_lr(step: int, max_step: int, init_lr: float, min_lr: float) -> float:
    """`cosine_lr_schedule`, minigpt4/common/optims.py:99-112."""
    return (init_lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * step / max_step)) + min_lr


def warmup_lr(step: int, max_step: int, init_lr: float, max_lr: float) -> float:
    """`warmup_lr_schedule`, optims.py:115-125."""
    return min(max_lr, init_lr + (max_lr - init_lr) * step / max(max_step, 1))


def lr_at(cur_epoch: int, cur_step: int, iters_per_epoch: int, max_epoch: int, init_lr: float,
          min_lr: float, warmup_steps: int = 0, warmup_start_lr: float = -1) -> float:
    """`LinearWarmupCosineLRScheduler.step`, optims.py:79-96."""
    total = cur_epoch * iters_per_epoch + cur_step
    if total < warmup_steps:
        return warmup_lr(cur_step, warmup_steps, warmup_start_lr if warmup_start_lr >= 0 else init_lr, init_lr)
    return cosine_lr(total, max_epoch * iters_per_epoch, init_lr, min_lr)


def uses_weight_decay(name: str, ndim: int) -> bool:
    """Parameter grouping of `RunnerBase.optimizer`, runners/runner_base.py:115-118."""
    return not (ndim < 2 or "bias" in name or "ln" in name or "bn" in name)


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, wd: float,
               beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
    """torch.optim.AdamW single-tensor update (the optimiser built at runner_base.py:132-137):
    decoupled decay p *= 1 - lr*wd, then Adam with bias correction.  In place; step is 1-based."""
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
    return p, m, v


def allreduce_mean_reference(grads_per_rank: List[Dict[str, Optional[Tensor]]]) -> Dict[str, Tensor]:
    """DDP(find_unused_parameters=True) semantics, runner_base.py:96-98: a parameter unused on a
    rank contributes zero; the sum is divided by world size."""
    world = len(grads_per_rank)
    names = set()
    for g in grads_per_rank:
        names.update(g.keys())
    out = {}
    for n in names:
        ref = next(g[n] for g in grads_per_rank if g.get(n) is not None)
        acc = torch.zeros_like(ref)
        for g in grads_per_rank:
            if g.get(n) is not None:
                acc += g[n]
        out[n] = acc / world
    return out

"""CPU restatement of the vision expert's vision side -- TEST INFRASTRUCTURE ONLY, like myriad_ref.py:
imported by tests/ and tools/ only, never by the product path.

Restates, in plain fp32 torch keyed by the reference's own state_dict names:
  * the ImageBind vision trunk as `imagebind_huge` builds it
      PadIm2Video(repeat x2) + Conv3d(3 -> D, (2,14,14), no bias)      imagebind_model.py:151-164,
                                                                        multimodal_preprocessors.py:121-158, 423-444
      cls token + learnable position table (257 tokens at 224 px)       multimodal_preprocessors.py:195-299
      pre-transformer LayerNorm(eps 1e-6), N pre-LN blocks of nn.MultiheadAttention + GELU MLP (ratio 4),
      block outputs tapped at `out_layers` BEFORE any final norm       transformer.py:94-96, 104-177, 245-287;
                                                                        imagebind_model.py:296-328
      head: LayerNorm -> cls -> Linear(D -> C, no bias) -> L2 normalise imagebind_model.py:383-387, 437-441
  * the anomaly-map heads of `adrefexpert.forward`                      adrefexpert_v2.py:16-29, 243-301
      zero-shot: per tap Linear(D -> C) on the 256 patch tokens, L2 normalise, 100 * cos-sim against the
                 [normal, abnormal] text embeddings, softmax over the pair at 16x16 (mask) and after bilinear
                 (align_corners=True) upsampling to 224 (map), class 1, mean over taps
      one-shot:  per tap cosine similarity of every query patch against all patches of its k reference images,
                 max over references, mean over taps, 1 - sim at 16x16 (mask) and after bilinear upsampling (map)
The text-prompt ensemble (adrefexpert_v2.py:69-99) depends only on the class name; its [B, 2, C] output is an INPUT here.
Pinned by tests/golden/expert_*.npz, produced by the reference's own modules / forward code (tools/make_golden_expert.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

PRE = "modality_preprocessors.vision."
TRK = "modality_trunks.vision."
HEAD = "modality_heads.vision."


def patch_embed(sd: Dict[str, torch.Tensor], image: torch.Tensor) -> torch.Tensor:
    """[B,3,H,W] -> [B, 1 + (H/14)(W/14), D]: the image is repeated along time, so the (2,14,14) Conv3d equals a 14x14
    Conv2d with the two temporal kernel slices summed; then cls token and the position table."""
    w = sd[PRE + "rgbt_stem.proj.1.weight"].float()            # [D, 3, 2, 14, 14]
    x = F.conv2d(image.float(), w.sum(dim=2), stride=14)       # [B, D, h, w]
    x = x.flatten(2).transpose(1, 2)
    cls = sd[PRE + "cls_token"].float().expand(x.shape[0], -1, -1)
    x = torch.cat((cls, x), dim=1)
    pos = sd[PRE + "pos_embedding_helper.pos_embed"].float()
    assert pos.shape[1] == x.shape[1], "position-table interpolation (non-224 inputs) is not on this path"
    return x + pos


def _mha(x, w_in, b_in, w_out, b_out, heads):
    B, L, D = x.shape
    qkv = x @ w_in.t() + b_in
    q, k, v = qkv.split(D, dim=-1)
    hd = D // heads

    def sh(t):
        return t.view(B, L, heads, hd).transpose(1, 2)

    a = (sh(q) @ sh(k).transpose(-1, -2)) / math.sqrt(hd)
    o = (a.softmax(-1) @ sh(v)).transpose(1, 2).reshape(B, L, D)
    return o @ w_out.t() + b_out


def vision_trunk(sd: Dict[str, torch.Tensor], image: torch.Tensor, heads: int, out_layers: Sequence[int],
                 n_blocks: int) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """Returns (image embedding [B, C] L2-normalised, [tap [B, L, D] for each out_layer])."""
    g = lambda k: sd[k].float()
    x = patch_embed(sd, image)
    D = x.shape[-1]
    x = F.layer_norm(x, (D,), g(TRK + "pre_transformer_layer.0.weight"), g(TRK + "pre_transformer_layer.0.bias"), 1e-6)
    taps = []
    for i in range(n_blocks):
        p = f"{TRK}blocks.{i}."
        h = F.layer_norm(x, (D,), g(p + "norm_1.weight"), g(p + "norm_1.bias"), 1e-6)
        x = x + _mha(h, g(p + "attn.in_proj_weight"), g(p + "attn.in_proj_bias"), g(p + "attn.out_proj.weight"),
                     g(p + "attn.out_proj.bias"), heads)
        h = F.layer_norm(x, (D,), g(p + "norm_2.weight"), g(p + "norm_2.bias"), 1e-6)
        h = F.gelu(h @ g(p + "mlp.fc1.weight").t() + g(p + "mlp.fc1.bias"))
        x = x + h @ g(p + "mlp.fc2.weight").t() + g(p + "mlp.fc2.bias")
        if i in out_layers:
            taps.append(x)
    e = F.layer_norm(x, (D,), g(HEAD + "0.weight"), g(HEAD + "0.bias"), 1e-6)[:, 0] @ g(HEAD + "2.weight").t()
    return e / e.norm(dim=-1, keepdim=True), taps


def zero_shot_logits(taps: List[torch.Tensor], decoder_sd: Dict[str, torch.Tensor], text_feats: torch.Tensor) -> List[torch.Tensor]:
    """The pair logits 100 * cos(decoded patch, [normal, abnormal] text) of adrefexpert_v2.py:277-289, per tap: [B, L, 2].
    (What zero_shot_maps soft-maxes; tests compare these, where a cosine error is not amplified through the softmax.)"""
    out = []
    for i, t in enumerate(taps):
        p = t[:, 1:].float() @ decoder_sd[f"fc.{i}.weight"].float().t() + decoder_sd[f"fc.{i}.bias"].float()
        p = p / p.norm(dim=-1, keepdim=True)
        out.append(100.0 * p @ text_feats.float().transpose(-2, -1))
    return out


def zero_shot_maps(taps: List[torch.Tensor], decoder_sd: Dict[str, torch.Tensor], text_feats: torch.Tensor,
                   out_size: int = 224) -> Tuple[torch.Tensor, torch.Tensor]:
    """adrefexpert_v2.py:277-301.  taps [B, 1+L, D]; decoder_sd keys `fc.{i}.weight/bias`; text_feats [B, 2, C]
    (row 0 normal, row 1 abnormal, each L2-normalised).  Returns (maps [B,1,S,S], masks [B,1,h,h])."""
    maps, masks = [], []
    for i, t in enumerate(taps):
        p = t[:, 1:].float() @ decoder_sd[f"fc.{i}.weight"].float().t() + decoder_sd[f"fc.{i}.bias"].float()
        p = p / p.norm(dim=-1, keepdim=True)
        logits = 100.0 * p @ text_feats.float().transpose(-2, -1)              # [B, L, 2]
        B, L, _ = logits.shape
        h = int(math.sqrt(L))
        grid = logits.permute(0, 2, 1).reshape(B, 2, h, h)
        masks.append(torch.softmax(grid, dim=1)[:, 1:])
        up = F.interpolate(grid, size=out_size, mode="bilinear", align_corners=True)
        maps.append(torch.softmax(up, dim=1)[:, 1:])
    return torch.stack(maps).mean(0), torch.stack(masks).mean(0)


def one_shot_maps(query_taps: List[torch.Tensor], ref_taps: List[torch.Tensor], out_size: int = 224
                  ) -> Tuple[torch.Tensor, torch.Tensor]:
    """adrefexpert_v2.py:243-276.  query_taps [B, 1+L, D]; ref_taps [B*k, 1+L, D] with the k references of sample b
    at rows b*k .. b*k+k-1.  Returns (anomaly map [B,1,S,S] = 1 - upsampled sim, simmask [B,1,h,h] = 1 - sim)."""
    sims = []
    for q, r in zip(query_taps, ref_taps):
        q, r = q[:, 1:].float(), r[:, 1:].float()
        B, L, D = q.shape
        r = r.reshape(B, -1, D)
        cs = F.cosine_similarity(q.view(B, L, 1, D), r.view(B, 1, -1, D), dim=-1)
        sims.append(cs.max(dim=-1).values)
    sim = torch.stack(sims).mean(0)
    h = int(math.sqrt(sim.shape[1]))
    sim = sim.reshape(-1, 1, h, h)
    up = F.interpolate(sim, size=out_size, mode="bilinear", align_corners=True)
    return 1 - up, 1 - sim


# ------------------------------------------------------------------------------------------------ text tower
TPRE, TTRK, THEAD, TPOST = ("modality_preprocessors.text.", "modality_trunks.text.", "modality_heads.text.",
                            "modality_postprocessors.text.")


def text_trunk(sd: Dict[str, torch.Tensor], ids: torch.Tensor, heads: int, n_blocks: int) -> torch.Tensor:
    """ImageBind text branch (multimodal_preprocessors.py:326-403; imagebind_model.py:330-337 trunk without the
    pre-transformer LayerNorm, 388-393 head, 423-425 post-processor): token + position embedding, causal pre-LN blocks,
    the row at the EOT token (= arg-max id, OpenCLIP pooling), LayerNorm -> Linear (no bias) -> L2 normalise ->
    times min(exp(log_logit_scale), 100).  ids [n, 77] int64 -> [n, C]."""
    g = lambda k: sd[k].float()
    x = g(TPRE + "token_embedding.weight")[ids] + g(TPRE + "pos_embed")
    n, L, D = x.shape
    mask = torch.full((L, L), float("-inf")).triu_(1)
    hd = D // heads
    for i in range(n_blocks):
        p = f"{TTRK}blocks.{i}."
        h = F.layer_norm(x, (D,), g(p + "norm_1.weight"), g(p + "norm_1.bias"), 1e-6)
        qkv = h @ g(p + "attn.in_proj_weight").t() + g(p + "attn.in_proj_bias")
        q, k, v = (t.view(n, L, heads, hd).transpose(1, 2) for t in qkv.split(D, dim=-1))
        a = ((q @ k.transpose(-1, -2)) / math.sqrt(hd) + mask).softmax(-1)
        o = (a @ v).transpose(1, 2).reshape(n, L, D)
        x = x + o @ g(p + "attn.out_proj.weight").t() + g(p + "attn.out_proj.bias")
        h = F.layer_norm(x, (D,), g(p + "norm_2.weight"), g(p + "norm_2.bias"), 1e-6)
        x = x + F.gelu(h @ g(p + "mlp.fc1.weight").t() + g(p + "mlp.fc1.bias")) @ g(p + "mlp.fc2.weight").t() + g(p + "mlp.fc2.bias")
    eot = x[torch.arange(n), ids.argmax(dim=-1)]
    e = F.layer_norm(eot, (D,), g(THEAD + "proj.0.weight"), g(THEAD + "proj.0.bias"), 1e-6) @ g(THEAD + "proj.1.weight").t()
    e = F.normalize(e, dim=-1)
    return torch.clip(g(TPOST + "1.log_logit_scale").exp(), max=100.0) * e


def text_prompt_ensemble(emb_normal: torch.Tensor, emb_abnormal: torch.Tensor, n_obj: int) -> torch.Tensor:
    """adrefexpert_v2.py:69-99: per object, mean over its normal / abnormal prompt sentences, L2 normalise each ->
    [n_obj, 2, C] with row 0 normal, row 1 abnormal."""
    C = emb_normal.shape[-1]
    outs = []
    for e in (emb_normal, emb_abnormal):
        m = e.reshape(n_obj, -1, C).mean(dim=1, keepdim=True)
        outs.append(m / m.norm(dim=-1, keepdim=True))
    return torch.cat(outs, dim=1)

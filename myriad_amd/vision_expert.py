"""The vision expert on the HIP kernels: ImageBind-Huge vision trunk + the anomaly-map heads of
`adrefexpert.forward` (frozen, forward only).  Produces the `anomaly_maps` / `oneshot_anomaly_maps` that
`Myriad.prepare_sample` (reference myriad.py:340-345) feeds to the hot path.

Mirrors
  * the trunk `imagebind_huge` builds: PadIm2Video(repeat) + Conv3d(3->D,(2,14,14)) patch stem, cls token + position table,
    pre-transformer LayerNorm, pre-LN blocks of nn.MultiheadAttention (fused in_proj with bias) + GELU MLP, block outputs
    tapped at `out_layers` (imagebind_model.py:151-164, 296-328; multimodal_preprocessors.py:121-299, 423-444;
    transformer.py:94-177, 245-287);
  * `adrefexpert.forward` (adrefexpert_v2.py:243-301): zero-shot maps from per-tap Linear decoders against the cached
    [normal, abnormal] text embeddings, one-shot maps from the best-matching reference patch.
The text-prompt ensemble depends only on the class name (adrefexpert_v2.py:69-99): its [B, 2, C] result is an input.

MI355X mapping: the repeated-frame Conv3d is one 14x14 patch GEMM with the two temporal kernel slices summed (the position
table rides the residual epilogue); blocks are LN(fp32->bf16) -> qkv GEMM(+bias) -> fused attention (head_dim 80, padded to 96
in LDS) -> out_proj GEMM(+bias +fp32 residual) -> LN -> fc1 GEMM(+bias +erf-GELU) -> fc2 GEMM(+bias +residual) -- the same
kernels as the EVA encoder; the query x reference similarity is one bf16 MFMA GEMM per sample and tap on L2-normalised
tokens, followed by a row-max; everything else is the small fp32 kernels of csrc/expert.hip.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32
PRE, TRK, HEAD = "modality_preprocessors.vision.", "modality_trunks.vision.", "modality_heads.vision."


class ImageBindVisionHIP:
    def __init__(self, sd: Dict[str, torch.Tensor], n_heads: int, out_layers: Sequence[int], device, eps: float = 1e-6):
        dev = self.dev = torch.device(device)
        self.H, self.eps, self.out_layers = n_heads, eps, list(out_layers)

        def bf(t):
            return t.detach().to(device=dev, dtype=BF16).contiguous()

        def f32(t):
            return t.detach().to(device=dev, dtype=F32).contiguous()

        w3 = sd[PRE + "rgbt_stem.proj.1.weight"].float()                 # [D, 3, T, 14, 14]; the frame is repeated T times
        pw = w3.sum(dim=2)
        self.D, self.C, self.P = pw.shape[0], pw.shape[1], pw.shape[2]
        K = self.C * self.P * self.P
        w = torch.zeros(self.D, ops.round_up(K, 64), dtype=torch.float32)
        w[:, :K] = pw.reshape(self.D, K)
        self.patch_w = bf(w)
        pos = sd[PRE + "pos_embedding_helper.pos_embed"].reshape(-1, self.D).float()
        self.cls_row = f32(sd[PRE + "cls_token"].reshape(1, self.D).float() + pos[:1])
        self.pos_patches = f32(pos[1:])
        self.pre_w, self.pre_b = f32(sd[TRK + "pre_transformer_layer.0.weight"]), f32(sd[TRK + "pre_transformer_layer.0.bias"])
        self.hd = self.D // n_heads
        self.blocks: List[dict] = []
        i = 0
        while (TRK + f"blocks.{i}.norm_1.weight") in sd:
            p = TRK + f"blocks.{i}."
            self.blocks.append(dict(
                n1w=f32(sd[p + "norm_1.weight"]), n1b=f32(sd[p + "norm_1.bias"]),
                wqkv=bf(sd[p + "attn.in_proj_weight"]), bqkv=f32(sd[p + "attn.in_proj_bias"]),
                wproj=bf(sd[p + "attn.out_proj.weight"]), bproj=f32(sd[p + "attn.out_proj.bias"]),
                n2w=f32(sd[p + "norm_2.weight"]), n2b=f32(sd[p + "norm_2.bias"]),
                w1=bf(sd[p + "mlp.fc1.weight"]), b1=f32(sd[p + "mlp.fc1.bias"]),
                w2=bf(sd[p + "mlp.fc2.weight"]), b2=f32(sd[p + "mlp.fc2.bias"])))
            i += 1
        self.head_nw, self.head_nb = f32(sd[HEAD + "0.weight"]), f32(sd[HEAD + "0.bias"])
        self.head_w = bf(sd[HEAD + "2.weight"])

    @torch.no_grad()
    def forward(self, image: torch.Tensor, want_embedding: bool = False) -> Tuple[Optional[torch.Tensor], List[torch.Tensor]]:
        """image [B,3,224,224] f32 (device) -> (image embedding [B, C] f32 L2-normalised or None, taps [B, 257, D] f32)."""
        B = image.shape[0]
        D, H, hd = self.D, self.H, self.hd
        patches = ops.patchify(image.contiguous(), self.P)               # [B*np, Kpad] bf16
        np_ = patches.shape[0] // B
        N = np_ + 1
        if np_ != self.pos_patches.shape[0]:
            raise ValueError(f"position table has {self.pos_patches.shape[0]} patch rows, image gives {np_} (224 px only)")
        x = torch.empty((B, N, D), dtype=F32, device=self.dev)
        ops.copy3d(self.cls_row.view(1, 1, D).expand(B, 1, D), x[:, :1])
        for b in range(B):   # per image so the position table rides the residual epilogue and rows land at x[b,1:]
            ops.gemm(patches[b * np_:(b + 1) * np_], self.patch_w, out=x[b, 1:], residual=self.pos_patches)
        M = B * N
        _, h = ops.layernorm_fwd(x.view(M, D), self.pre_w, self.pre_b, self.eps, want_bf16=False, want_f32=True)
        scale = hd ** -0.5
        taps = []
        for li, blk in enumerate(self.blocks):
            xn, _ = ops.layernorm_fwd(h, blk["n1w"], blk["n1b"], self.eps)
            qkv = ops.gemm(xn, blk["wqkv"], bias=blk["bqkv"]).view(B, N, 3 * D)
            o, _ = ops.attn_fwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H, hd, scale, need_lse=False)
            h = ops.gemm(o.view(M, D), blk["wproj"], bias=blk["bproj"], residual=h, out_dtype=F32)
            xn, _ = ops.layernorm_fwd(h, blk["n2w"], blk["n2b"], self.eps)
            a = ops.gemm(xn, blk["w1"], bias=blk["b1"], gelu=True)
            h = ops.gemm(a, blk["w2"], bias=blk["b2"], residual=h, out_dtype=F32)
            if li in self.out_layers:
                taps.append(h.view(B, N, D))         # every GEMM writes a fresh buffer: the tap stays valid
        emb = None
        if want_embedding:
            cls = h.view(B, N, D)[:, 0].contiguous()
            cn, _ = ops.layernorm_fwd(cls, self.head_nw, self.head_nb, self.eps)
            e = ops.gemm(cn, self.head_w, out_dtype=F32)
            _, emb = ops.l2norm_rows(e, want_bf16=False, want_f32=True)
        return emb, taps


class VisionExpertHIP:
    """`adrefexpert` (adrefexpert_v2.py:101-301) without its file-system side: reference images and the per-class text
    embeddings are passed in."""

    def __init__(self, sd: Dict[str, torch.Tensor], n_heads: int = 16, out_layers: Sequence[int] = (7, 15, 23, 31),
                 device="cuda:0", out_size: int = 224):
        self.dev = torch.device(device)
        self.trunk = ImageBindVisionHIP(sd, n_heads, out_layers, device)
        self.out_size = out_size
        self.dec_w, self.dec_b = [], []
        i = 0
        while f"image_decoder.fc.{i}.weight" in sd:
            self.dec_w.append(sd[f"image_decoder.fc.{i}.weight"].detach().to(self.dev, BF16).contiguous())
            self.dec_b.append(sd[f"image_decoder.fc.{i}.bias"].detach().to(self.dev, F32).contiguous())
            i += 1
        if self.dec_w and len(self.dec_w) != len(list(out_layers)):
            raise ValueError("one image_decoder.fc per tapped layer is required")

    @torch.no_grad()
    def zero_shot(self, images: torch.Tensor, text_feats: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """images [B,3,224,224], text_feats [B,2,C] (L2-normalised [normal, abnormal]) -> (maps [B,1,S,S], masks [B,1,h,h])."""
        _, taps = self.trunk.forward(images.to(self.dev, F32))
        return self._zero_shot_from_taps(taps, text_feats)

    def _zero_shot_from_taps(self, taps, text_feats, keep_logits=None):
        """`keep_logits`: a list that receives each tap's pair logits [B*L, 2] (100 * cosine; tests compare these)."""
        B, N, D = taps[0].shape
        L = N - 1
        h = int(round(L ** 0.5))
        S = self.out_size
        mask = torch.zeros((B, h, h), dtype=F32, device=self.dev)
        amap = torch.zeros((B, S, S), dtype=F32, device=self.dev)
        text = text_feats.to(self.dev, F32).contiguous()
        p = torch.empty((B * L, self.dec_w[0].shape[0]), dtype=F32, device=self.dev)
        for t, w, b in zip(taps, self.dec_w, self.dec_b):
            tok = ops.to_bf16(t.reshape(B * N, D))
            for i in range(B):                                                        # patch rows only: class token dropped
                ops.gemm(tok[i * N + 1:(i + 1) * N], w, bias=b, out=p[i * L:(i + 1) * L])   # image_decoder.fc[tap]
            logits = ops.pair_logits(p, text, L, 100.0)                               # 100 * cos(p, text)
            if keep_logits is not None:
                keep_logits.append(logits.clone())
            ops.zs_accumulate(logits, mask, amap, 1.0 / len(taps))
        return amap.view(B, 1, S, S), mask.view(B, 1, h, h)

    @torch.no_grad()
    def one_shot(self, images: torch.Tensor, ref_images: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """images [B,3,224,224], ref_images [B*k,3,224,224] (the k normal references of sample b at rows b*k..) ->
        (anomaly map [B,1,S,S] = 1 - upsampled similarity, simmask [B,1,h,h] = 1 - similarity)."""
        _, qt = self.trunk.forward(images.to(self.dev, F32))
        _, rt = self.trunk.forward(ref_images.to(self.dev, F32))
        return self._one_shot_from_taps(qt, rt)

    def _one_shot_from_taps(self, qt, rt):
        B, N, D = qt[0].shape
        k = rt[0].shape[0] // B
        L = N - 1
        h = int(round(L ** 0.5))
        sim = torch.zeros((B * L,), dtype=F32, device=self.dev)
        ncol = k * N
        scores = torch.empty((L, ops.round_up(ncol, 4)), dtype=F32, device=self.dev)
        for q, r in zip(qt, rt):
            qn, _ = ops.l2norm_rows(q.reshape(B * N, D), eps=1e-8)                    # cosine_similarity operands
            rn, _ = ops.l2norm_rows(r.reshape(B * k * N, D), eps=1e-8)
            for b in range(B):
                ops.gemm(qn[b * N + 1:(b + 1) * N], rn[b * k * N:(b + 1) * k * N], out=scores[:, :ncol])
                ops.rowmax_skip(scores[:, :ncol], sim[b * L:(b + 1) * L], N, 1.0 / len(qt))   # class-token columns skipped
        sim = sim.view(B, h, h)
        amap = ops.bilinear_ac(sim, self.out_size, self.out_size, one_minus=True)
        simmask = ops.bilinear_ac(sim, h, h, one_minus=True)                          # same grid: just 1 - sim
        return amap.view(B, 1, self.out_size, self.out_size), simmask.view(B, 1, h, h)

    @torch.no_grad()
    def forward(self, images: torch.Tensor, text_feats: torch.Tensor, ref_images: torch.Tensor):
        """Both map pairs `Myriad.prepare_sample` asks for (myriad.py:340-345) from ONE trunk pass over
        [images ; references] (the reference runs the trunk three times: zero-shot queries, one-shot queries,
        references).  Returns ((maps, masks), (oneshot_maps, oneshot_masks)) with the shapes of the two methods above."""
        B = images.shape[0]
        both = torch.cat([images.to(self.dev, F32), ref_images.to(self.dev, F32)], dim=0)
        _, taps = self.trunk.forward(both)
        qt = [t[:B] for t in taps]
        rt = [t[B:] for t in taps]
        return self._zero_shot_from_taps(qt, text_feats), self._one_shot_from_taps(qt, rt)


TPRE, TTRK, THEAD, TPOST = ("modality_preprocessors.text.", "modality_trunks.text.", "modality_heads.text.",
                            "modality_postprocessors.text.")


class ImageBindTextHIP:
    """ImageBind text tower (multimodal_preprocessors.py:326-403, imagebind_model.py:330-337, 388-393, 423-425) and the
    prompt ensemble of `encode_text_with_prompt_ensemble` (adrefexpert_v2.py:69-99).  Token ids are host inputs (the CLIP
    BPE tokenisation is host-side string work).  The result depends only on the class name: compute once, cache, and
    hand the [n_obj, 2, C] pair to `VisionExpertHIP.zero_shot`."""

    def __init__(self, sd: Dict[str, torch.Tensor], n_heads: int, device, eps: float = 1e-6):
        import math
        dev = self.dev = torch.device(device)
        self.H, self.eps = n_heads, eps

        def bf(t):
            return t.detach().to(device=dev, dtype=BF16).contiguous()

        def f32(t):
            return t.detach().to(device=dev, dtype=F32).contiguous()

        self.tok = f32(sd[TPRE + "token_embedding.weight"])              # fp32 table: nn.Embedding is not autocast
        self.pos = f32(sd[TPRE + "pos_embed"])                           # [1, 77, D]
        self.ctx, self.D = self.pos.shape[1], self.pos.shape[2]
        self.hd = self.D // n_heads
        self.blocks: List[dict] = []
        i = 0
        while (TTRK + f"blocks.{i}.norm_1.weight") in sd:
            p = TTRK + f"blocks.{i}."
            self.blocks.append(dict(
                n1w=f32(sd[p + "norm_1.weight"]), n1b=f32(sd[p + "norm_1.bias"]),
                wqkv=bf(sd[p + "attn.in_proj_weight"]), bqkv=f32(sd[p + "attn.in_proj_bias"]),
                wproj=bf(sd[p + "attn.out_proj.weight"]), bproj=f32(sd[p + "attn.out_proj.bias"]),
                n2w=f32(sd[p + "norm_2.weight"]), n2b=f32(sd[p + "norm_2.bias"]),
                w1=bf(sd[p + "mlp.fc1.weight"]), b1=f32(sd[p + "mlp.fc1.bias"]),
                w2=bf(sd[p + "mlp.fc2.weight"]), b2=f32(sd[p + "mlp.fc2.bias"])))
            i += 1
        self.head_nw, self.head_nb = f32(sd[THEAD + "proj.0.weight"]), f32(sd[THEAD + "proj.0.bias"])
        self.head_w = bf(sd[THEAD + "proj.1.weight"])
        self.logit_scale = min(math.exp(float(sd[TPOST + "1.log_logit_scale"])), 100.0)

    @torch.no_grad()
    def forward(self, ids: torch.Tensor) -> torch.Tensor:
        """ids [n, 77] int (host or device) -> [n, C] f32 = logit_scale * L2-normalised sentence embeddings."""
        ids = ids.cpu().long()
        n, L = ids.shape
        D, H, hd = self.D, self.H, self.hd
        M = n * L
        h = ops.gather_rows_f32(self.tok, ids.reshape(-1).to(torch.int32).to(self.dev))
        ops.copy3d(self.pos.expand(n, L, D), h.view(n, L, D), accumulate=True)
        scale = hd ** -0.5
        for blk in self.blocks:
            xn, _ = ops.layernorm_fwd(h, blk["n1w"], blk["n1b"], self.eps)
            qkv = ops.gemm(xn, blk["wqkv"], bias=blk["bqkv"]).view(n, L, 3 * D)
            o, _ = ops.attn_fwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H, hd, scale, causal=True, need_lse=False)
            h = ops.gemm(o.view(M, D), blk["wproj"], bias=blk["bproj"], residual=h, out_dtype=F32)
            xn, _ = ops.layernorm_fwd(h, blk["n2w"], blk["n2b"], self.eps)
            a = ops.gemm(xn, blk["w1"], bias=blk["b1"], gelu=True)
            h = ops.gemm(a, blk["w2"], bias=blk["b2"], residual=h, out_dtype=F32)
        eot = (torch.arange(n) * L + ids.argmax(dim=-1)).to(torch.int32).to(self.dev)      # OpenCLIP pooling: the EOT row
        e = ops.gather_rows_f32(h, eot)
        en, _ = ops.layernorm_fwd(e, self.head_nw, self.head_nb, self.eps)
        e = ops.gemm(en, self.head_w, out_dtype=F32)
        _, e = ops.l2norm_rows(e, want_bf16=False, want_f32=True)
        return ops.scale_(e, self.logit_scale)

    @torch.no_grad()
    def prompt_ensemble(self, ids_normal: torch.Tensor, ids_abnormal: torch.Tensor, n_obj: int) -> torch.Tensor:
        """[n_obj * n_normal, 77] and [n_obj * n_abnormal, 77] prompt token ids -> [n_obj, 2, C]: per object the mean over
        its sentences, L2-normalised (normalising the sum is the same vector)."""
        en, ea = self.forward(ids_normal), self.forward(ids_abnormal)
        C = en.shape[1]
        out = torch.empty((n_obj * 2, C), dtype=F32, device=self.dev)
        for src, col in ((en, 0), (ea, 1)):
            per = src.shape[0] // n_obj
            for i in range(n_obj):
                out[i * 2 + col].copy_(ops.colsum(src[i * per:(i + 1) * per]))        # device-to-device row placement
        _, feats = ops.l2norm_rows(out, want_bf16=False, want_f32=True)
        return feats.view(n_obj, 2, C)

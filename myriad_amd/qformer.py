"""BLIP-2 Q-Former (query-only path) on the HIP kernels: forward + dgrad-only backward to BOTH inputs.

Mirrors `BertModel.forward` (reference minigpt4/models/Qformer.py:804-965) as called by `Myriad.encode_img`
(myriad.py:256-261): `BertEmbeddings` LayerNorm of the raw query embeddings (:104-107), then per `BertLayer`
(:402-474) self-attention + `BertSelfOutput` (:285-289), cross-attention to the image tokens on layers with
layer_num % 2 == 0 (:386-395), and the query FFN `feed_forward_chunk_query` (:481-484).  Post-LN, eps 1e-12,
scores scaled by 1/sqrt(64) after q.k^T (:244), dropout inactive (frozen/eval, myriad.py:159-165).
The module is frozen but gradients flow to the query embeddings (VEInstructor tokens) and to the image tokens
(expert_adaptor), SURVEY 3.3.  The hidden/residual stream is fp32, GEMM operands bf16.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32


class QFormerHIP:
    def __init__(self, sd: Dict[str, torch.Tensor], n_heads: int, device, eps: float = 1e-12, cross_freq: int = 2,
                 prefix: str = "Qformer.bert.", need_backward: bool = True):
        dev = self.dev = torch.device(device)
        self.H, self.eps = n_heads, eps

        def bf(t):
            return t.detach().to(device=dev, dtype=BF16).contiguous()

        def f32(t):
            return t.detach().to(device=dev, dtype=F32).contiguous()

        self.emb_w = f32(sd[prefix + "embeddings.LayerNorm.weight"])
        self.emb_b = f32(sd[prefix + "embeddings.LayerNorm.bias"])
        self.D = self.emb_w.numel()
        self.hd = self.D // n_heads
        self.layers: List[dict] = []
        i = 0
        while (prefix + f"encoder.layer.{i}.attention.self.query.weight") in sd:
            p = prefix + f"encoder.layer.{i}."
            a = p + "attention."
            L = dict(
                wqkv=bf(torch.cat([sd[a + "self.query.weight"], sd[a + "self.key.weight"], sd[a + "self.value.weight"]], 0)),
                bqkv=f32(torch.cat([sd[a + "self.query.bias"], sd[a + "self.key.bias"], sd[a + "self.value.bias"]], 0)),
                wo=bf(sd[a + "output.dense.weight"]), bo=f32(sd[a + "output.dense.bias"]),
                ln_a_w=f32(sd[a + "output.LayerNorm.weight"]), ln_a_b=f32(sd[a + "output.LayerNorm.bias"]),
                w1=bf(sd[p + "intermediate_query.dense.weight"]), b1=f32(sd[p + "intermediate_query.dense.bias"]),
                w2=bf(sd[p + "output_query.dense.weight"]), b2=f32(sd[p + "output_query.dense.bias"]),
                ln_f_w=f32(sd[p + "output_query.LayerNorm.weight"]), ln_f_b=f32(sd[p + "output_query.LayerNorm.bias"]),
                cross=(i % cross_freq == 0))
            if L["cross"]:
                c = p + "crossattention."
                L.update(cwq=bf(sd[c + "self.query.weight"]), cbq=f32(sd[c + "self.query.bias"]),
                         cwkv=bf(torch.cat([sd[c + "self.key.weight"], sd[c + "self.value.weight"]], 0)),
                         cbkv=f32(torch.cat([sd[c + "self.key.bias"], sd[c + "self.value.bias"]], 0)),
                         cwo=bf(sd[c + "output.dense.weight"]), cbo=f32(sd[c + "output.dense.bias"]),
                         ln_c_w=f32(sd[c + "output.LayerNorm.weight"]), ln_c_b=f32(sd[c + "output.LayerNorm.bias"]))
                if L["cwkv"].shape[1] % 64:
                    raise ValueError("encoder width must be a multiple of 64 (GEMM K granule)")
            if need_backward:
                for k in ("wqkv", "wo", "w1", "w2", "cwq", "cwkv", "cwo"):
                    if k in L:
                        L[k + "T"] = L[k].t().contiguous()
            self.layers.append(L)
            i += 1
        # The key / value projections of every cross-attention layer read the same image tokens (Qformer.py:172-176 with
        # encoder_hidden_states): one [n_cross * 2D, We] matrix turns 6 GEMMs inside the dependent layer chain into ONE in front
        # of it (forward) and 6 accumulating dgrad GEMMs into ONE K = n_cross * 2D product behind it (backward); same values
        # per layer -- each output column is the same dot product.
        cross = [L for L in self.layers if L["cross"]]
        self.n_cross = len(cross)
        if cross:
            self.cwkv_all = torch.cat([L["cwkv"] for L in cross], 0).contiguous()
            self.cbkv_all = torch.cat([L["cbkv"] for L in cross], 0).contiguous()
            self.cwkvT_all = torch.cat([L["cwkvT"] for L in cross], 1).contiguous() if need_backward else None
            for L in cross:
                L.pop("cwkv"); L.pop("cbkv"); L.pop("cwkvT", None)
        self._saved = None

    def forward(self, query_embeds: torch.Tensor, enc_b: torch.Tensor, save_for_backward: bool = True):
        """query_embeds [B,nq,D] f32; enc_b [B,Nenc,We] bf16 image tokens.  Returns [B,nq,D] f32."""
        B, nq, D = query_embeds.shape
        Ne, We = enc_b.shape[1], enc_b.shape[2]
        M, H, hd = B * nq, self.H, self.hd
        scale = 1.0 / math.sqrt(hd)
        q_in = query_embeds.reshape(M, D).contiguous()
        hb, h = ops.layernorm_fwd(q_in, self.emb_w, self.emb_b, self.eps, want_bf16=True, want_f32=True)
        enc2 = enc_b.reshape(B * Ne, We)
        ckv_all = ops.gemm(enc2, self.cwkv_all, bias=self.cbkv_all).view(B, Ne, self.n_cross * 2 * D) if self.n_cross else None
        ci = 0
        saved = []
        for L in self.layers:
            s = {}
            # self-attention
            qkv = ops.gemm(hb, L["wqkv"], bias=L["bqkv"]).view(B, nq, 3 * D)
            ctx, lse = ops.attn_fwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H, hd, scale)
            y = ops.gemm(ctx.view(M, D), L["wo"], bias=L["bo"], residual=h, out_dtype=F32)
            hb, h = ops.layernorm_fwd(y, L["ln_a_w"], L["ln_a_b"], self.eps, want_bf16=True, want_f32=True)
            s.update(qkv=qkv, ctx=ctx, lse=lse, y_a=y)
            if L["cross"]:
                cq = ops.gemm(hb, L["cwq"], bias=L["cbq"]).view(B, nq, D)
                ckv = ckv_all[:, :, ci * 2 * D:(ci + 1) * 2 * D]
                ci += 1
                cctx, clse = ops.attn_fwd(cq, ckv[:, :, :D], ckv[:, :, D:], H, hd, scale)
                yc = ops.gemm(cctx.view(M, D), L["cwo"], bias=L["cbo"], residual=h, out_dtype=F32)
                hb, h = ops.layernorm_fwd(yc, L["ln_c_w"], L["ln_c_b"], self.eps, want_bf16=True, want_f32=True)
                s.update(cq=cq, ckv=ckv, cctx=cctx, clse=clse, y_c=yc)
            pre, act = ops.gemm_gelu_fwd(hb, L["w1"], L["b1"])       # GELU in the product's epilogue (one launch)
            yf = ops.gemm(act, L["w2"], bias=L["b2"], residual=h, out_dtype=F32)
            hb, h = ops.layernorm_fwd(yf, L["ln_f_w"], L["ln_f_b"], self.eps, want_bf16=True, want_f32=True)
            s.update(pre=pre, y_f=yf)
            saved.append(s)
        if save_for_backward:
            self._saved = dict(layers=saved, q_in=q_in, B=B, nq=nq, Ne=Ne, We=We, scale=scale)
        return h.view(B, nq, D)

    def backward(self, dout: torch.Tensor):
        """dout [B,nq,D] f32 -> (d_query_embeds [B,nq,D] f32, d_enc [B,Nenc,We] f32)."""
        sv = self._saved
        if sv is None:
            raise RuntimeError("backward() without saved forward")
        B, nq, Ne, We, scale = sv["B"], sv["nq"], sv["Ne"], sv["We"], sv["scale"]
        D, H, hd = self.D, self.H, self.hd
        M = B * nq
        dh = dout.reshape(M, D).contiguous()
        dckv_all = torch.empty((B, Ne, self.n_cross * 2 * D), dtype=BF16, device=self.dev) if self.n_cross else None
        ci = self.n_cross
        for L, s in zip(reversed(self.layers), reversed(sv["layers"])):
            # FFN:  h_out = LN(y_f),  y_f = act(h W1^T+b1) W2^T + b2 + h
            dy, dyb = ops.layernorm_bwd(dh, s["y_f"], L["ln_f_w"], self.eps, want_bf16=True)
            dpre = ops.gemm_gelu_bwd(dyb, L["w2T"], s["pre"])        # gelu'(pre) in the dgrad's epilogue (one launch)
            dh = ops.gemm(dpre, L["w1T"], residual=dy, out_dtype=F32)
            if L["cross"]:
                dy, dyb = ops.layernorm_bwd(dh, s["y_c"], L["ln_c_w"], self.eps, want_bf16=True)
                dctx = ops.gemm(dyb, L["cwoT"]).view(B, nq, D)
                ckv = s["ckv"]
                ci -= 1
                dckv = dckv_all[:, :, ci * 2 * D:(ci + 1) * 2 * D]
                dcq, _, _ = ops.attn_bwd(s["cq"], ckv[:, :, :D], ckv[:, :, D:], s["cctx"], dctx, s["clse"], H, hd, scale,
                                         dk=dckv[:, :, :D], dv=dckv[:, :, D:])
                dh = ops.gemm(dcq.view(M, D), L["cwqT"], residual=dy, out_dtype=F32)
            dy, dyb = ops.layernorm_bwd(dh, s["y_a"], L["ln_a_w"], self.eps, want_bf16=True)
            dctx = ops.gemm(dyb, L["woT"]).view(B, nq, D)
            qkv = s["qkv"]
            dqkv = torch.empty_like(qkv)
            ops.attn_bwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], s["ctx"], dctx, s["lse"], H, hd, scale,
                         dq=dqkv[:, :, :D], dk=dqkv[:, :, D:2 * D], dv=dqkv[:, :, 2 * D:])
            dh = ops.gemm(dqkv.view(M, 3 * D), L["wqkvT"], residual=dy, out_dtype=F32)
        dq_in, _ = ops.layernorm_bwd(dh, sv["q_in"], self.emb_w, self.eps)
        if self.n_cross:                                  # d(image tokens) = sum over the cross layers, as one K = n_cross*2D product
            denc = ops.gemm(dckv_all.view(B * Ne, self.n_cross * 2 * D), self.cwkvT_all, out_dtype=F32)
        else:
            denc = torch.zeros((B * Ne, We), dtype=F32, device=self.dev)
        self._saved = None
        return dq_in.view(B, nq, D), denc.view(B, Ne, We)

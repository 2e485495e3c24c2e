"""EVA ViT-g/14 vision encoder on the HIP kernels (frozen, forward only).

Mirrors `VisionTransformer.forward_features` (reference minigpt4/models/eva_vit.py:324-340; no final norm/head),
`PatchEmbed.forward` (:198-204), `Block.forward` (:173-180), `Attention.forward` (:118-148: fused qkv with bias
cat(q_bias, 0, v_bias), q scaled by head_dim^-0.5, optional additive rel-pos bias) and `Mlp.forward` (:54-61).

MI355X mapping: patch embedding = patchify + MFMA GEMM with the positional embedding fused as the residual
epilogue; per block LN(fp32->bf16) -> qkv GEMM(+bias) -> fused attention (head_dim 88 padded to 96 in LDS) ->
proj GEMM(+bias +fp32 residual) -> LN -> fc1 GEMM(+bias +erf-GELU epilogue) -> fc2 GEMM(+bias +residual).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32


class EvaViTHIP:
    def __init__(self, sd: Dict[str, torch.Tensor], n_heads: int, device, eps: float = 1e-6,
                 prefix: str = "visual_encoder."):
        dev = self.dev = torch.device(device)
        self.H, self.eps = n_heads, eps

        def bf(t):
            return t.detach().to(device=dev, dtype=BF16).contiguous()

        def f32(t):
            return t.detach().to(device=dev, dtype=F32).contiguous()

        pw = sd[prefix + "patch_embed.proj.weight"]
        self.D, self.C, self.P = pw.shape[0], pw.shape[1], pw.shape[2]
        K = self.C * self.P * self.P
        Kpad = ops.round_up(K, 64)
        w = torch.zeros(self.D, Kpad, dtype=pw.dtype, device=pw.device)
        w[:, :K] = pw.reshape(self.D, K)
        self.patch_w = bf(w)
        self.patch_b = f32(sd[prefix + "patch_embed.proj.bias"])
        cls = sd[prefix + "cls_token"].reshape(1, self.D).float()
        pos = sd.get(prefix + "pos_embed")
        pos = pos.reshape(-1, self.D).float() if pos is not None else torch.zeros(1, self.D, device=pw.device)
        self.n_tok = pos.shape[0] if (prefix + "pos_embed") in sd else None
        self.cls_row = f32(cls + pos[:1])          # x[:,0] = cls + pos[0]
        self.pos_patches = f32(pos[1:]) if pos.shape[0] > 1 else None
        self.hd = self.D // n_heads
        self.blocks: List[dict] = []
        i = 0
        while (prefix + f"blocks.{i}.norm1.weight") in sd:
            p = prefix + f"blocks.{i}."
            qb, vb = sd[p + "attn.q_bias"], sd[p + "attn.v_bias"]
            w1, b1, w2 = sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], sd[p + "mlp.fc2.weight"]
            Hd = w1.shape[0]
            Hp = ops.round_up(Hd, 64)   # int(1408*4.3637)=6144 already; tiny configs get zero-padded
            if Hp != Hd:
                w1 = torch.cat([w1, torch.zeros(Hp - Hd, w1.shape[1], dtype=w1.dtype, device=w1.device)], 0)
                b1 = torch.cat([b1, torch.zeros(Hp - Hd, dtype=b1.dtype, device=b1.device)], 0)
                w2 = torch.cat([w2, torch.zeros(w2.shape[0], Hp - Hd, dtype=w2.dtype, device=w2.device)], 1)
            self.blocks.append(dict(
                n1w=f32(sd[p + "norm1.weight"]), n1b=f32(sd[p + "norm1.bias"]),
                wqkv=bf(sd[p + "attn.qkv.weight"]), bqkv=f32(torch.cat([qb, torch.zeros_like(vb), vb])),
                wproj=bf(sd[p + "attn.proj.weight"]), bproj=f32(sd[p + "attn.proj.bias"]),
                n2w=f32(sd[p + "norm2.weight"]), n2b=f32(sd[p + "norm2.bias"]),
                w1=bf(w1), b1=f32(b1), w2=bf(w2), b2=f32(sd[p + "mlp.fc2.bias"])))
            i += 1

    @torch.no_grad()
    def forward(self, image: torch.Tensor, rel_pos_bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        """image [B,3,H,W] f32 (device) -> [B, 1+np, D] f32."""
        state = self.embed(image)
        state = self.run_blocks(state, 0, len(self.blocks), rel_pos_bias)
        return self.finish(state)

    @torch.no_grad()
    def embed(self, image: torch.Tensor):
        """Patch embedding + cls / positional rows and the first block's LayerNorm: the state that run_blocks carries."""
        B = image.shape[0]
        D = self.D
        patches = ops.patchify(image.contiguous(), self.P)               # [B*np, Kpad] bf16
        np_ = patches.shape[0] // B
        N = np_ + 1
        x = torch.empty((B, N, D), dtype=F32, device=self.dev)
        ops.copy3d(self.cls_row.view(1, 1, D).expand(B, 1, D), x[:, :1])
        for b in range(B):   # per image so the pos-embed rides the residual epilogue and rows land at x[b,1:]
            ops.gemm(patches[b * np_:(b + 1) * np_], self.patch_w, out=x[b, 1:], bias=self.patch_b,
                     residual=self.pos_patches)
        h = x.view(B * N, D)
        xn = ops.layernorm_fwd(h, self.blocks[0]["n1w"], self.blocks[0]["n1b"], self.eps)[0] if self.blocks else None
        return (h, xn, B, N)

    @torch.no_grad()
    def run_blocks(self, state, lo: int, hi: int, rel_pos_bias: Optional[torch.Tensor] = None):
        """Blocks lo .. hi-1 on a state from embed() / an earlier run_blocks (a forward may be issued in pieces)."""
        h, xn, B, N = state
        D, H, hd = self.D, self.H, self.hd
        M = B * N
        scale = hd ** -0.5
        # each LayerNorm rides the split-K reduce of the GEMM that produces its input when that GEMM is split
        # (ops.gemm_residual_layernorm: fc2 -> next block's norm1, proj -> norm2); only block 0's norm1 is its own launch
        nb = len(self.blocks)
        for bi in range(lo, hi):
            blk = self.blocks[bi]
            qkv = ops.gemm(xn, blk["wqkv"], bias=blk["bqkv"]).view(B, N, 3 * D)
            o, _ = ops.attn_fwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H, hd, scale, bias=rel_pos_bias,
                                need_lse=False)
            h, xn = ops.gemm_residual_layernorm(o.view(M, D), blk["wproj"], blk["bproj"], h, blk["n2w"], blk["n2b"], self.eps)
            a = ops.gemm(xn, blk["w1"], bias=blk["b1"], gelu=True)
            if bi + 1 < nb:
                nxt = self.blocks[bi + 1]
                h, xn = ops.gemm_residual_layernorm(a, blk["w2"], blk["b2"], h, nxt["n1w"], nxt["n1b"], self.eps)
            else:
                h = ops.gemm(a, blk["w2"], bias=blk["b2"], residual=h, out_dtype=F32)
        return (h, xn, B, N)

    @staticmethod
    def finish(state) -> torch.Tensor:
        h, _, B, N = state
        return h.view(B, N, -1)

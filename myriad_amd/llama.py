"""Vicuna / LLaMA decoder on the HIP kernels: training forward + dgrad-only backward, and KV-cache decode.

Mirrors the reference's `LlamaForCausalLM` (minigpt4/models/modeling_llama.py:629-716), `LlamaModel.forward`
(:466-596), `LlamaDecoderLayer.forward` (:247-299), `LlamaAttention.forward` (:168-231) and
`clamp_CE_loss` (:718-728) for the call patterns Myriad uses (inputs_embeds + attention_mask + labels; greedy
decode from inputs_embeds).  Weights are frozen (myriad.py:202-205), so the backward is dgrad only.

MI355X layout decisions (288 GB HBM): every frozen weight is kept twice in bf16 -- W [out,in] for forward and
W^T [in,out] for dgrad -- so forward and backward both run the single K-contiguous MFMA GEMM; q/k/v and
gate/up are fused into one GEMM each; the residual stream and all norm statistics are fp32; the lm_head and
the clamp-CE loss run only on label-bearing rows (the other rows have zero gradient and no loss term).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32


def _bf(t: torch.Tensor, dev) -> torch.Tensor:
    return t.detach().to(device=dev, dtype=BF16).contiguous()


def _f32(t: torch.Tensor, dev) -> torch.Tensor:
    return t.detach().to(device=dev, dtype=F32).contiguous()


class LlamaHIP:
    def __init__(self, sd: Dict[str, torch.Tensor], n_heads: int, device, eps: float = 1e-6,
                 prefix: str = "llama_model.", max_pos: int = 2048, need_backward: bool = True):
        self.dev = torch.device(device)
        self.H = n_heads
        self.eps = eps
        p = prefix + "model."
        self.embed = _bf(sd[p + "embed_tokens.weight"], self.dev)
        self.V, self.D = self.embed.shape
        self.hd = self.D // n_heads
        self.layers: List[dict] = []
        i = 0
        while (p + f"layers.{i}.input_layernorm.weight") in sd:
            lp = p + f"layers.{i}."
            wq, wk, wv = (sd[lp + f"self_attn.{n}_proj.weight"] for n in "qkv")
            wqkv = _bf(torch.cat([wq, wk, wv], 0), self.dev)
            wo = _bf(sd[lp + "self_attn.o_proj.weight"], self.dev)
            # intermediate size padded to a multiple of 128 (the gate|up interleave block; also the GEMM K granule): zero
            # rows/cols are exact no-ops.  gate and up rows are interleaved in blocks of 128 so that one 256-column tile of the
            # gate|up GEMM holds g and u of the same columns and silu(g)*u rides its epilogue (ops.gemm_swiglu_fwd)
            wg, wu, wdn = sd[lp + "mlp.gate_proj.weight"], sd[lp + "mlp.up_proj.weight"], sd[lp + "mlp.down_proj.weight"]
            I0 = wg.shape[0]
            Ip = ops.round_up(I0, ops.SWIGLU_BLK)
            if Ip != I0:
                z = torch.zeros(Ip - I0, wg.shape[1], dtype=wg.dtype, device=wg.device)
                wg, wu = torch.cat([wg, z], 0), torch.cat([wu, z], 0)
                wdn = torch.cat([wdn, torch.zeros(wdn.shape[0], Ip - I0, dtype=wdn.dtype, device=wdn.device)], 1)
            wgu = ops.interleave_gate_up(_bf(wg, self.dev), _bf(wu, self.dev))
            wd = _bf(wdn, self.dev)
            L = dict(wqkv=wqkv, wo=wo, wgu=wgu, wd=wd,
                     ln1=_f32(sd[lp + "input_layernorm.weight"], self.dev),
                     ln2=_f32(sd[lp + "post_attention_layernorm.weight"], self.dev))
            if need_backward:
                L.update(wqkvT=wqkv.t().contiguous(), woT=wo.t().contiguous(), wguT=wgu.t().contiguous(),
                         wdT=wd.t().contiguous())
            self.layers.append(L)
            i += 1
        self.I = self.layers[0]["wd"].shape[1]
        self.norm = _f32(sd[p + "norm.weight"], self.dev)
        self.lm_head = _bf(sd[prefix + "lm_head.weight"], self.dev)
        self.Vpad = ops.round_up(self.V, 64)
        if need_backward:
            lmT = torch.zeros((self.D, self.Vpad), dtype=BF16, device=self.dev)
            lmT[:, :self.V] = self.lm_head.t()
            self.lm_headT = lmT
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, self.hd, 2).float() / self.hd))
        fr = torch.einsum("i,j->ij", torch.arange(max_pos).float(), inv_freq)
        self._pos_cache = {}
        self.cos = fr.cos().contiguous().to(self.dev)
        self.sin = fr.sin().contiguous().to(self.dev)
        self._saved = None
        self.lora = None
        # decode keeps a second, stream-ordered copy of the frozen weights (ops.gemv_pack; +1x the LLM's bf16 bytes, 13.5 GB
        # for Vicuna-7B out of 288 GB) built at the first generate(); MYRIAD_PACK_DECODE=0 streams the row-major ones
        self.pack_decode = os.environ.get("MYRIAD_PACK_DECODE", "1") != "0"
        # the LoRA weight gradients feed only the optimiser: queued during the dgrad chain, launched on a side stream after it
        # (55.7 -> 55.1 ms per step: they run beside the Q-Former backward); MYRIAD_LORA_DEFER=0 computes them in place
        self.defer_lora_wgrad = os.environ.get("MYRIAD_LORA_DEFER", "1") != "0"
        self.decode_fused = os.environ.get("MYRIAD_DECODE_FUSED", "1") != "0"
        self.last_layer_rows = os.environ.get("MYRIAD_LAST_LAYER_ROWS", "1") != "0"
        self._packed = None
        self._decode_ws = {}

    def _pack_for_decode(self) -> None:
        """(Re)build the packed copies the single-token step streams.  Frozen matrices are packed once; the bordered qkv
        weight carries the LoRA B columns, which training moves, so it is re-packed (in place) at every generate()."""
        qkv_key = "wqkv" if self.lora is None else "wqkv_ext"
        if self._packed is None:
            self._packed = dict(layers=[{k: ops.gemv_pack(L[k]) for k in ("wo", "wgu", "wd")} for L in self.layers],
                                lm_head=ops.gemv_pack(self.lm_head), qkv_key=None)
        if self._packed["qkv_key"] != qkv_key or self.lora is not None:
            for L, P in zip(self.layers, self._packed["layers"]):
                P["wqkv"] = ops.gemv_pack(L[qkv_key], out=P.get("wqkv") if self._packed["qkv_key"] == qkv_key else None)
            self._packed["qkv_key"] = qkv_key

    def attach_lora(self, lora) -> None:
        """Enable PEFT-style LoRA on q_proj/v_proj (myriad_amd.lora.LoraQV); replaces W_qkv by its bordered copy."""
        for L in self.layers:
            lora.extend_weights(L)
        self.lora = lora

    # ------------------------------------------------------------------ training forward
    def forward_loss(self, x: torch.Tensor, attention_mask: torch.Tensor, labels: torch.Tensor,
                     save_for_backward: bool = True, lora_training: Optional[bool] = None) -> torch.Tensor:
        """x: [B,S,D] f32 inputs_embeds (device); attention_mask/labels: [B,S] CPU or device int64.
        Returns the 0-d f32 loss tensor (device).  Stores what `backward()` needs."""
        B, S, D = x.shape
        M = B * S
        H, hd, W = self.H, self.hd, self.D
        am = attention_mask.to("cpu")
        kv_len_host = am.sum(-1).to(torch.int32)
        if not bool((am == (torch.arange(S)[None] < kv_len_host[:, None])).all()):
            raise ValueError("attention_mask must be right-padded (ones then zeros), as the reference builds it")
        kv_len = ops.h2d(kv_len_host, self.dev)                         # async uploads: no launch-thread stall
        pos = self._pos_ids(B, S)                                       # position_ids = arange (modeling_llama.py:519-523)
        scale = 1.0 / math.sqrt(hd)
        h = x.reshape(M, D)
        saved = []
        lora = self.lora
        if lora is not None:
            lora.refresh(self.layers)
        # The RMSNorm that consumes a residual-stream GEMM rides that GEMM's split-K reduce (ops.gemm_residual_rmsnorm):
        # o_proj -> post-attention norm, down_proj -> the NEXT layer's input norm (written straight into that layer's
        # bordered LoRA operand when LoRA is on).  Only layer 0's input norm is a launch of its own.
        def norm_target(li):
            if lora is None:
                return None
            return lora.x_ext(li, M)[:, :D]

        # whole-sequence attention with the rotary embedding fused (csrc/attn_seq.hip) when the sequence fits a CU's LDS:
        # qkv is then saved PRE-rotary and the backward kernel rotates again / un-rotates dq, dk itself
        fused_attn = ops.attn_rope_supported(S, hd) and os.environ.get("MYRIAD_ATTN_SEQ", "1") != "0"
        # loss on label-bearing rows only: row (b,s) predicts labels[b,s+1]
        lab = labels.to("cpu")
        shift = lab[:, 1:]
        bi, si = torch.nonzero(shift != -100, as_tuple=True)
        n_valid = int(bi.numel())
        if n_valid == 0:
            raise ValueError("no valid labels")
        row_idx = (bi * S + si).to(torch.int32)
        rows = ops.h2d(row_idx, self.dev)
        tgt = ops.h2d(shift[bi, si], self.dev)
        # The LAST layer's o_proj, post-attention norm and MLP (modeling_llama.py:281-293) run on those rows only: the other rows
        # of its output feed nothing (the final norm and lm_head read label rows), and in the backward their gradient is exactly
        # zero down to that layer's attention, which mixes rows -- so its k / v projections and everything below stay dense.
        # Identical results, ~1.4 % fewer executed FLOPs at the bench shape; MYRIAD_LAST_LAYER_ROWS=0 keeps every row.
        # (not for <= 16 label rows, the batch-1 step: those products would run on the weight-streaming GEMV, which reads the
        # row-major matrices slower than the 160-row tiles run all 148 rows -- measured +0.25 ms per batch-1 step)
        last_rows = self.last_layer_rows and len(self.layers) > 0 and 16 < n_valid <= M // 2
        inv = None
        if last_rows:
            inv_host = torch.full((M,), -1, dtype=torch.int32)
            inv_host[row_idx.long()] = torch.arange(n_valid, dtype=torch.int32)
            inv = ops.h2d(inv_host, self.dev)
        xn = ops.rmsnorm_fwd(h, self.layers[0]["ln1"], self.eps, out=norm_target(0)) if self.layers else None
        hr = None
        for li, L in enumerate(self.layers):
            lsave = None
            if lora is None:
                qkv = ops.gemm(xn, L["wqkv"])                               # [M, 3W] bf16
            else:   # q/v LoRA rides the qkv GEMM as a 64-column K border (myriad_amd/lora.py)
                x_ext = lora.x_ext(li, M)                                   # [:, :D] already holds rmsnorm(h)
                p_eff, seed = lora.forward_border(li, x_ext, training=save_for_backward if lora_training is None
                                                  else lora_training)
                qkv = ops.gemm(x_ext, L["wqkv_ext"])
                lsave = (x_ext, p_eff, seed)
            q3 = qkv.view(B, S, 3 * W)
            if fused_attn:
                o, lse = ops.attn_rope_fwd(q3, H, hd, scale, pos, self.cos, self.sin, kv_len=kv_len)
            else:
                ops.rope_(qkv, 0, 2 * H, hd, pos, self.cos, self.sin, 1.0)  # q and k heads
                o, lse = ops.attn_fwd(q3[:, :, :W], q3[:, :, W:2 * W], q3[:, :, 2 * W:], H, hd, scale, causal=True,
                                      kv_len=kv_len)
            if last_rows and li + 1 == len(self.layers):
                o_r = ops.gather_rows(o.view(M, W), rows)                   # [R, W] bf16
                h_r = ops.gather_rows(h, rows)                              # [R, D] f32
                h2, xn2 = ops.gemm_residual_rmsnorm(o_r, L["wo"], h_r, L["ln2"], self.eps)
                gu, act = ops.gemm_swiglu_fwd(xn2, L["wgu"])
                hr = ops.gemm(act, L["wd"], residual=h2, out_dtype=F32)     # the final hidden state of the label rows
                if save_for_backward:
                    saved.append((h, qkv, o, lse, h2, gu, lsave))
                break
            h2, xn2 = ops.gemm_residual_rmsnorm(o.view(M, W), L["wo"], h, L["ln2"], self.eps)
            gu, act = ops.gemm_swiglu_fwd(xn2, L["wgu"])                    # [M, 2I] (interleaved g|u blocks), [M, I]
            if li + 1 < len(self.layers):
                h3, xn = ops.gemm_residual_rmsnorm(act, L["wd"], h2, self.layers[li + 1]["ln1"], self.eps,
                                                   y_out=norm_target(li + 1))
            else:
                h3 = ops.gemm(act, L["wd"], residual=h2, out_dtype=F32)     # the final norm runs on label rows only
            if save_for_backward:
                saved.append((h, qkv, o, lse, h2, gu, lsave))
            h = h3
        if hr is None:
            hr = ops.gather_rows_f32(h, rows)
        hn = ops.rmsnorm_fwd(hr, self.norm, self.eps)
        logits = ops.gemm(hn, self.lm_head, out_dtype=F32)                  # [R, V] f32
        row_loss, dlogits = ops.clamp_ce(logits, tgt, 1.0 / n_valid, want_grad=save_for_backward, ldd=self.Vpad)
        loss = ops.sum_f32(row_loss, 1.0 / n_valid)
        if save_for_backward:
            self._saved = dict(layers=saved, rows=rows, hr=hr, dlogits=dlogits, B=B, S=S, kv_len=kv_len, pos=pos,
                               scale=scale, fused_attn=fused_attn, inv=inv)
        return loss.view(())

    # ------------------------------------------------------------------ dgrad-only backward
    def backward(self, loss_scale: float = 1.0, defer_lora_join: bool = False) -> torch.Tensor:
        """Returns d(loss)/d(inputs_embeds) as [B,S,D] f32.  The LoRA weight gradients of all layers are launched on a side
        stream once the dgrad chain is done; with defer_lora_join the caller joins them (lora.join_wgrads()) after its own
        remaining backward work, otherwise they are joined here."""
        sv = self._saved
        if sv is None:
            raise RuntimeError("backward() called without a saved forward")
        B, S = sv["B"], sv["S"]
        M, D, W, H, hd = B * S, self.D, self.D, self.H, self.hd
        dlog = sv["dlogits"]
        if loss_scale != 1.0:
            raise NotImplementedError("loss scaling is unnecessary in bf16; pass 1.0")
        dhn = ops.gemm(dlog, self.lm_headT, out_dtype=F32)                  # [R, D]
        dhr, _ = ops.rmsnorm_bwd(dhn, sv["hr"], self.norm, self.eps)
        inv = sv.get("inv")
        if inv is None:
            dh = torch.zeros((M, D), dtype=F32, device=self.dev)
            ops.scatter_rows(dhr, sv["rows"], dh)
            dh_b = ops.to_bf16(dh)
        n_layers = len(self.layers)
        for ri, (L, (h_in, qkv, o, lse, h2, gu, lsave)) in enumerate(zip(reversed(self.layers), reversed(sv["layers"]))):
            li = n_layers - 1 - ri
            rows_mode = inv is not None and ri == 0      # the last layer ran its o_proj / MLP on the label rows only (forward_loss)
            if rows_mode:
                dgu = ops.gemm_swiglu_bwd(ops.to_bf16(dhr), L["wdT"], gu)   # [R, 2I]: every other row's gradient is exactly zero
                dh2_r, dh2_rb = ops.gemm_rmsnorm_bwd(dgu, L["wguT"], h2, L["ln2"], self.eps, dres=dhr)
                do_r = ops.gemm(dh2_rb, L["woT"])                           # [R, W] bf16
                dh2 = ops.expand_rows(dh2_r, inv, M)                        # back to all rows (zeros elsewhere): the residual path
                do = ops.expand_rows(do_r, inv, M)                          # ... and the attention's dO
            else:
                dgu = ops.gemm_swiglu_bwd(dh_b, L["wdT"], gu)               # down dgrad + gate backward: [M, 2I]
                # gate|up dgrad [M, D] and the post-attention norm's backward in one call (split-K slabs summed in the norm kernel)
                dh2, dh2_b = ops.gemm_rmsnorm_bwd(dgu, L["wguT"], h2, L["ln2"], self.eps, dres=dh)
            q3 = qkv.view(B, S, 3 * W)
            dqkv = torch.empty_like(qkv)
            d3 = dqkv.view(B, S, 3 * W)
            if rows_mode:
                if sv["fused_attn"]:
                    ops.attn_rope_bwd(q3, o, do.view(B, S, W), lse, H, hd, sv["scale"], sv["pos"], self.cos, self.sin,
                                      kv_len=sv["kv_len"], dqkv=d3)
                else:
                    ops.attn_bwd(q3[:, :, :W], q3[:, :, W:2 * W], q3[:, :, 2 * W:], o, do.view(B, S, W), lse, H, hd,
                                 sv["scale"], causal=True, kv_len=sv["kv_len"], dq=d3[:, :, :W], dk=d3[:, :, W:2 * W],
                                 dv=d3[:, :, 2 * W:])
                    ops.rope_(dqkv, 0, 2 * H, hd, sv["pos"], self.cos, self.sin, -1.0)
            elif sv["fused_attn"]:
                # o_proj dgrad + attention backward: the split-K slabs of dO are summed inside the attention kernel
                ops.gemm_attn_rope_bwd(dh2_b, L["woT"], q3, o, lse, H, hd, sv["scale"], sv["pos"], self.cos, self.sin,
                                       kv_len=sv["kv_len"], dqkv=d3)
            else:
                do = ops.gemm(dh2_b, L["woT"])                              # [M, W] bf16
                ops.attn_bwd(q3[:, :, :W], q3[:, :, W:2 * W], q3[:, :, 2 * W:], o, do.view(B, S, W), lse, H, hd,
                             sv["scale"], causal=True, kv_len=sv["kv_len"], dq=d3[:, :, :W], dk=d3[:, :, W:2 * W],
                             dv=d3[:, :, 2 * W:])
                ops.rope_(dqkv, 0, 2 * H, hd, sv["pos"], self.cos, self.sin, -1.0)
            if self.lora is None:
                dxn = ops.gemm(dqkv, L["wqkvT"], out_dtype=F32)
                dh, dh_b = ops.rmsnorm_bwd(dxn, h_in, L["ln1"], self.eps, dres=dh2, want_bf16=True)
            else:
                # [M, D+64] = base dgrad | d(s*t), the LoRA correction of d(xn) and the input norm's backward in one call: the
                # split-K slabs are summed inside the one kernel that does both (csrc/lora.hip)
                dh, dh_b = self.lora.backward_from_dqkv_norm(li, dqkv, L["wqkvT_ext"], lsave[0], lsave[1], lsave[2], h_in,
                                                             L["ln1"], self.eps, dh2, defer_wgrad=self.defer_lora_wgrad)
        self._saved = None
        if self.lora is not None:
            self.lora.run_deferred_wgrads()
            if not defer_lora_join:
                self.lora.join_wgrads()
        return dh.view(B, S, D)

    def _pos_ids(self, B: int, S: int) -> torch.Tensor:
        key = (B, S)
        if key not in self._pos_cache:
            self._pos_cache[key] = torch.arange(S, dtype=torch.int32).repeat(B).to(self.dev)
        return self._pos_cache[key]

    # ------------------------------------------------------------------ generation
    def _decode_block(self, h, B, S, caches, scale, pos, past=None, pos_dev=None, kvlen_dev=None):
        """All decoder layers for a prefill chunk (host-known `past`) or for one decode token whose position lives
        in device memory (`pos_dev`/`kvlen_dev`), which makes the launch sequence replayable from a hipGraph."""
        H, hd, W, D = self.H, self.hd, self.D, self.D
        M = B * S
        packed = self._packed["layers"] if (pos_dev is not None and M <= 16 and self._packed is not None) else None

        def lin(li, name, x, **kw):
            if packed is not None:
                return ops.gemv_packed(x, packed[li]["wqkv" if name.startswith("wqkv") else name], **kw)
            return ops.gemm(x, self.layers[li][name], **kw)

        # Single-token step on the packed copies: four launches per layer instead of nine -- the two RMSNorms
        # and the SiLU gate are rebuilt by every workgroup of the product that consumes them (mh_gemv_packed_rmsnorm /
        # _silu), rotary + KV append ride the attention launch (mh_attn_decode_rope); each fused form is bit-identical
        # to the launches it replaces (tests/test_kernels_gpu.py), MYRIAD_DECODE_FUSED=0 keeps the separate launches.
        # With LoRA attached the qkv product takes the bordered operand [xn | s A xn]: the norm and the LoRA down projection are
        # one launch (LoraQV.norm_border, <= 2 rows), the bordered packed weight the next -- five launches per layer become six.
        fused = packed is not None and self.decode_fused
        for li, (L, cache) in enumerate(zip(self.layers, caches)):
            if fused:
                P = packed[li]
                if self.lora is not None:
                    x_ext = self.lora.x_ext(li, M)
                    if not self.lora.norm_border(li, h, L["ln1"], self.eps, x_ext):
                        ops.rmsnorm_fwd(h, L["ln1"], self.eps, out=x_ext[:, :D])
                        self.lora.forward_border(li, x_ext, training=False)
                    qkv = ops.gemv_packed(x_ext, P["wqkv"])
                else:
                    qkv = ops.gemv_packed_rmsnorm(h, L["ln1"], self.eps, P["wqkv"])
                    if qkv is None:
                        qkv = ops.gemv_packed(ops.rmsnorm_fwd(h, L["ln1"], self.eps), P["wqkv"])
                o = ops.attn_decode_rope(qkv, cache, pos, pos_dev, kvlen_dev, self.cos, self.sin, H, hd, scale)
                h2 = ops.gemv_packed(o, P["wo"], residual=h, out_dtype=F32)
                gu = ops.gemv_packed_rmsnorm(h2, L["ln2"], self.eps, P["wgu"])
                if gu is None:
                    gu = ops.gemv_packed(ops.rmsnorm_fwd(h2, L["ln2"], self.eps), P["wgu"])
                hn = ops.gemv_packed_silu(gu, P["wd"], residual=h2, out_dtype=F32)
                h = hn if hn is not None else ops.gemv_packed(ops.silu_mul_fwd_blk(gu), P["wd"], residual=h2, out_dtype=F32)
                continue
            if self.lora is None:
                xn = ops.rmsnorm_fwd(h, L["ln1"], self.eps)
                qkv = lin(li, "wqkv", xn)
            else:
                x_ext = self.lora.x_ext(li, M)
                ops.rmsnorm_fwd(h, L["ln1"], self.eps, out=x_ext[:, :D])
                self.lora.forward_border(li, x_ext, training=False)
                qkv = lin(li, "wqkv_ext", x_ext)
            q3 = qkv.view(B, S, 3 * W)
            if pos_dev is None:
                ops.rope_(qkv, 0, 2 * H, hd, pos, self.cos, self.sin, 1.0)
                ops.copy3d_bf16(q3[:, :, W:], cache[:, past:past + S])       # append k|v (modeling_llama.py:190-195)
                kc = cache[:, :past + S]
                o, _ = ops.attn_fwd(q3[:, :, :W], kc[:, :, :W], kc[:, :, W:], H, hd, scale, causal=True, need_lse=False)
            else:
                ops.rope_kv_append(qkv, H, hd, pos, self.cos, self.sin, cache, pos_dev)   # rotary + append, one launch
                o, _ = ops.attn_fwd(q3[:, :, :W], cache[:, :, :W], cache[:, :, W:], H, hd, scale, causal=False,
                                    kv_len=kvlen_dev, need_lse=False)
            h2 = lin(li, "wo", o.view(M, W), residual=h, out_dtype=F32)
            xn2 = ops.rmsnorm_fwd(h2, L["ln2"], self.eps)
            act = ops.silu_mul_fwd_blk(lin(li, "wgu", xn2))
            h = lin(li, "wd", act, residual=h2, out_dtype=F32)
        return h

    def _decode_workspace(self, B: int, T_need: int, inv_temp: float):
        """Buffers (and, once captured, the hipGraph) of the single-token step for a batch size: KV caches, device-resident
        counters, id / logit / result buffers and per-step histories.  Kept across generate() calls -- an evaluation run
        calls generate() once per batch, and re-capturing ~290 launches each time cost ~9 ms per call."""
        T_cap = ops.round_up(T_need + 2, 64)
        key = (B, T_cap, float(inv_temp), id(self._packed), None if self._packed is None else self._packed.get("qkv_key"), self.decode_fused,
               self.lora is not None)
        ws = self._decode_ws.get(key)
        if ws is None:
            if len(self._decode_ws) >= 3:                               # a few shapes at most: evict the oldest
                self._decode_ws.pop(next(iter(self._decode_ws)))
            dev, i32 = self.dev, torch.int32
            ws = dict(T=T_cap, graph=None, warm=False,
                      caches=[torch.zeros((B, T_cap, 2 * self.D), dtype=BF16, device=dev) for _ in self.layers],
                      pos=torch.zeros((B,), dtype=i32, device=dev), kvlen=torch.zeros((B,), dtype=i32, device=dev),
                      ids=torch.zeros((B,), dtype=torch.long, device=dev), x_in=torch.empty((B, self.D), dtype=F32, device=dev),
                      logits=torch.empty((B, self.V), dtype=F32, device=dev),
                      nxt=torch.empty((B,), dtype=torch.long, device=dev), mar=torch.empty((B,), dtype=F32, device=dev),
                      pmx=torch.empty((B,), dtype=F32, device=dev), step=torch.zeros((1,), dtype=i32, device=dev),
                      rec=torch.zeros((3, B), dtype=F32, device=dev))
            self._decode_ws[key] = ws
        return ws

    @torch.no_grad()
    def greedy_generate(self, inputs_embeds: torch.Tensor, max_new_tokens: int = 90,
                        stop_ids=((835,), (2277, 29937)), eos_id: int = 2, min_length: int = 1,
                        return_margins: bool = False, use_graph: bool = True, do_sample: bool = False,
                        top_p: float = 1.0, temperature: float = 1.0, generator: Optional[torch.Generator] = None,
                        top_k: int = 50):
        """Decode from [B,S0,D] f32 embeddings with a KV cache (prefill + 1-token steps).  Same contract
        as the oracle's greedy_generate: stop when ROW 0 ends with a stop sequence (conversation.py:102-107),
        EOS banned while fewer than `min_length` tokens were generated, finished rows padded with EOS.

        The single-token step (~290 launches) is captured into a hipGraph once per batch size, kept across generate() calls
        (an evaluation run calls generate() once per batch) and replayed; everything it needs lives on the device -- position /
        valid-length counters, and the token it just picked is fed back as the next input by the step itself
        (mh_decode_record), which packs the step's picks into one small record: between two steps the host makes ONE
        device->host copy.  (Measured and dropped: launching step t+1 before reading step t -- back-to-back launches of one
        executable graph cost more than the host's 0.07 ms per step; writing the record straight into pinned host memory --
        +0.2 ms per token.)

        `do_sample=True, top_p, temperature` are the eval script's arguments (evaluation_aqa_dataset.py:289-301).  HF's
        top-p warper keeps the smallest descending-probability set whose mass reaches top_p (at least one token), so a
        step whose p_max >= top_p IS the arg-max; the kernel reports p_max per row and only a row below the threshold is
        drawn on the host from that row's logits (a genuine sample: reproducible here through `generator`, never
        bit-comparable with another framework's RNG) and replaces the fed-back id.  `last_generate_stats` counts such steps.
        The host draw applies HF's default `top_k = 50` filter first, then top-p; the device test p_max >= top_p is taken over the
        full vocabulary, which is the conservative side: the top-k renormalisation only raises p_max, and a row whose
        renormalised p_max reaches top_p keeps exactly one token in the host draw -- the arg-max again."""
        B, S0, D = inputs_embeds.shape
        scale = 1.0 / math.sqrt(self.hd)
        out_ids, margins = [], []
        unfinished = torch.ones(B, dtype=torch.long)
        inv_temp = 1.0 / float(temperature) if do_sample else 1.0
        stats = dict(steps=0, sampled_rows=0, min_pmax=1.0)
        self.last_generate_stats = stats
        if self.lora is not None:
            self.lora.refresh(self.layers)
        if self.pack_decode and B <= 16:
            self._pack_for_decode()
        elif not self.pack_decode:
            self._packed = None                                      # MYRIAD_PACK_DECODE=0: stream the row-major matrices
        ws = self._decode_workspace(B, S0 + max_new_tokens, inv_temp)
        caches = ws["caches"]

        def sample_row(logits_row: torch.Tensor, ban: int) -> int:
            """HF TopPLogitsWarper + multinomial on one row (host)."""
            lg = logits_row.float().cpu() * inv_temp
            if ban >= 0:
                lg[ban] = float("-inf")
            if top_k and 0 < top_k < lg.numel():                     # HF applies TopKLogitsWarper (default top_k = 50) before top-p
                lg = lg.masked_fill(lg < torch.topk(lg, top_k).values[-1], float("-inf"))
            srt, idx = torch.sort(lg, descending=False)
            cum = srt.softmax(-1).cumsum(-1)
            remove = cum <= (1.0 - top_p)
            remove[-1:] = False                                      # min_tokens_to_keep = 1
            srt = srt.masked_fill(remove, float("-inf"))
            probs = torch.zeros_like(lg).scatter(0, idx, srt.softmax(-1))
            return int(torch.multinomial(probs, 1, generator=generator))

        def record(nxt: torch.Tensor, mar: torch.Tensor, pm: torch.Tensor, ban: int, logits_of=None):
            """Host bookkeeping of one step's picks.  Returns (done, redrawn): redrawn = a live row was re-drawn on the host
            (finished rows are fed their raw arg-max instead of EOS by the device: rows are independent and their outputs are
            overwritten with EOS here)."""
            nonlocal unfinished
            margins.append(mar)
            stats["steps"] += 1
            redrawn = False
            if do_sample:
                stats["min_pmax"] = min(stats["min_pmax"], float(pm[unfinished.bool()].min()) if int(unfinished.sum()) else 1.0)
                for row in range(B):
                    if int(unfinished[row]) and float(pm[row]) < top_p:
                        nxt[row] = sample_row(logits_of()[row], ban)
                        stats["sampled_rows"] += 1
                        redrawn = True
            nxt = nxt * unfinished + eos_id * (1 - unfinished)       # HF pads finished rows with pad(=eos)
            unfinished = unfinished * (nxt != eos_id).long()
            out_ids.append(nxt)
            row0 = [int(t[0]) for t in out_ids]
            if any(len(row0) >= len(st) and row0[-len(st):] == list(st) for st in stop_ids):
                return True, redrawn
            return int(unfinished.max()) == 0, redrawn

        # ---- prefill (eager, host-known lengths)
        pos = torch.arange(S0, dtype=torch.int32).repeat(B).to(self.dev)
        h = self._decode_block(inputs_embeds.reshape(B * S0, D).contiguous(), B, S0, caches, scale, pos, past=0)
        last = h.view(B, S0, D)[:, -1].contiguous()
        logits0 = ops.gemm(ops.rmsnorm_fwd(last, self.norm, self.eps), self.lm_head, out_dtype=F32)
        ban0 = eos_id if 0 < min_length else -1
        ops.argmax_pmax_rows(logits0, ws["nxt"], ws["mar"], ws["pmx"], ban_id=ban0, inv_temp=inv_temp)
        done, _ = record(ws["nxt"].cpu(), ws["mar"].cpu(), ws["pmx"].cpu(), ban0, logits_of=lambda: logits0)

        # ---- single-token steps: everything the step reads is on the device
        ws["pos"].fill_(S0)                                          # position of the incoming token
        ws["kvlen"].fill_(S0 + 1)                                    # valid keys after the append
        ws["step"].zero_()

        def token_step(ban):
            ops.embed_gather(self.embed, ws["ids"], ws["x_in"])
            done_lm = None
            hh = self._decode_block(ws["x_in"], B, 1, caches, scale, ws["pos"], pos_dev=ws["pos"], kvlen_dev=ws["kvlen"])
            if done_lm is None and self._packed is not None and B <= 16 and self.decode_fused:
                done_lm = ops.gemv_packed_rmsnorm(hh, self.norm, self.eps, self._packed["lm_head"], out=ws["logits"], out_dtype=F32)
            if done_lm is None:
                hn = ops.rmsnorm_fwd(hh, self.norm, self.eps)
                if self._packed is not None and B <= 16:
                    ops.gemv_packed(hn, self._packed["lm_head"], out=ws["logits"], out_dtype=F32)
                else:
                    ops.gemm(hn, self.lm_head, out=ws["logits"])
            ops.argmax_pmax_rows(ws["logits"], ws["nxt"], ws["mar"], ws["pmx"], ban_id=ban, inv_temp=inv_temp)
            ops.decode_advance(ws["nxt"], ws["mar"], ws["pmx"], ws["rec"], ws["ids"], ws["step"], ws["pos"], ws["kvlen"])

        def launch(ban):
            """Enqueue one token step: a replay of the captured graph when there is one."""
            if ban == -1 and use_graph and ws["graph"] is not None:
                ws["graph"].replay()
                return
            token_step(ban)
            if ban == -1 and use_graph and ws["warm"]:
                # the eager step above was this batch size's second: capture the next one (kernels are warm, buffers fixed)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    token_step(-1)
                ws["graph"] = g
            ws["warm"] = True

        step = 1                                                     # tokens generated so far (= index of the next one)
        if not done and step < max_new_tokens:
            ws["ids"].copy_(out_ids[-1].to(self.dev))
        while not done and step < max_new_tokens:
            ban = eos_id if step < min_length else -1
            launch(ban)
            rec = ws["rec"].cpu()                                    # the one device->host copy of the step (it also waits for it)
            done, redrawn = record(rec[0].long(), rec[1].clone(), rec[2].clone(), ban, logits_of=lambda: ws["logits"])
            if redrawn and not done:
                ws["ids"].copy_(out_ids[-1].to(self.dev))            # a host draw replaces the arg-max the step fed back to itself
            step += 1
        ids = torch.stack(out_ids, 1)
        if return_margins:
            return ids, torch.stack(margins, 1)
        return ids

    def embed_tokens_into(self, ids: torch.Tensor, out2d: torch.Tensor, dst_rows: Optional[torch.Tensor] = None):
        ops.embed_gather(self.embed, ids, out2d, dst_rows)

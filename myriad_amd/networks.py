"""Trainable adapters of Myriad on the HIP kernels (reference minigpt4/models/networks.py:71-197):
`LoraAdaptorV2` (rank-4 residual adaptor, :81-93), `VEInstructorV2` (:95-153) and `VETokenizer` (:156-197).

The conv stacks run as im2col + the MFMA GEMM on NHWC bf16 activations (csrc/conv.hip); conv weights are held
in GEMM order [Cout, (ky,kx,ci)] in the flat fp32 trainable buffer and converted to/from the reference's
[Cout,Cin,kh,kw] only at the checkpoint boundary (`to_reference_layout` / `from_reference_layout`).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32

STEM_CH = [(1, 4), (4, 16), (16, 64), (64, 256), (256, 1024)]
STEM_IDX = (0, 3, 6, 9, 12)


def ve_param_specs(prefix: str, head_out: int, head_k: int) -> List[Tuple[str, Tuple[int, ...], Tuple[int, ...]]]:
    """[(reference name, internal shape, reference shape)] for one VE network."""
    specs = []
    for idx, (ci, co) in zip(STEM_IDX, STEM_CH):
        specs.append((f"{prefix}meta_net.{idx}.weight", (co, 9 * ci), (co, ci, 3, 3)))
        specs.append((f"{prefix}meta_net.{idx}.bias", (co,), (co,)))
    specs.append((f"{prefix}meta_net.15.weight", (head_out, head_k * head_k * 1024), (head_out, 1024, head_k, head_k)))
    specs.append((f"{prefix}meta_net.15.bias", (head_out,), (head_out,)))
    return specs


def from_reference_layout(t: torch.Tensor, internal_shape) -> torch.Tensor:
    if t.dim() == 4:   # [Cout,Cin,kh,kw] -> [Cout, (ky,kx,ci)]
        return t.permute(0, 2, 3, 1).reshape(internal_shape).contiguous()
    return t.reshape(internal_shape).contiguous()


def to_reference_layout(t: torch.Tensor, ref_shape) -> torch.Tensor:
    if len(ref_shape) == 4:
        co, ci, kh, kw = ref_shape
        return t.reshape(co, kh, kw, ci).permute(0, 3, 1, 2).contiguous()
    return t.reshape(ref_shape).contiguous()


class VENet:
    """5 x (conv3x3 pad1 -> ReLU -> maxpool2) stem + head (1x1 conv -> 49 query tokens, or 5x5 valid conv ->
    9 LLM tokens).  `params`/`grads` map reference names to fp32 views of the flat trainable buffers."""

    def __init__(self, prefix: str, head_k: int, head_out: int, params: Dict[str, torch.Tensor],
                 grads: Dict[str, torch.Tensor], device):
        self.prefix, self.head_k, self.head_out = prefix, head_k, head_out
        self.p, self.g, self.dev = params, grads, torch.device(device)
        self._saved = None

    def _w(self, idx):
        return self.p[f"{self.prefix}meta_net.{idx}.weight"], self.p[f"{self.prefix}meta_net.{idx}.bias"]

    def forward(self, maps: torch.Tensor, save_for_backward: bool = True) -> torch.Tensor:
        """maps [B,1,224,224] f32 -> tokens [B, T, head_out] f32 (T = 49 or 9)."""
        B, _, H, W = maps.shape
        x = ops.to_bf16(maps.contiguous()).view(B, H, W, 1)   # C=1: NCHW == NHWC
        saved = []
        for idx, (ci, co) in zip(STEM_IDX, STEM_CH):
            wm, bias = self._w(idx)
            wp = ops.conv_pack(wm, bias)
            col = ops.im2col(x, 3, 3, 1)
            y = ops.gemm(col, wp, out_dtype=F32)               # [B*H*W, co] fp32 pre-activation (bias = ones column)
            saved.append((col, y, wp, H, W, ci, co))
            x = ops.relu_pool_fwd(y, B, H, W, co)
            H, W = H // 2, W // 2
        wm, bias = self._w(15)
        if self.head_k == 1:
            x2 = x.view(B * H * W, 1024)
            wb = ops.to_bf16(wm)
            out = ops.gemm(x2, wb, bias=bias, out_dtype=F32)
            head = (x2, wb, None, True)
            T = H * W
        else:
            k = self.head_k
            T = (H - k + 1) * (W - k + 1)
            if wm.shape[1] % 64 == 0:
                # the 105 M-parameter VETokenizer head (K = 25 * 1024): no bias column, so the bf16 operand is a plain cast and the
                # weight gradient GEMM writes straight into the flat gradient buffer (no packed copy to unpack: -0.8 GB per step)
                wp = ops.to_bf16(wm)
                col = ops.im2col(x, k, k, 0, bias_col=False)
                out = ops.gemm(col, wp, bias=bias, out_dtype=F32)
                head = (col, wp, (H, W), True)
            else:
                wp = ops.conv_pack(wm, bias)
                col = ops.im2col(x, k, k, 0)
                out = ops.gemm(col, wp, out_dtype=F32)
                head = (col, wp, (H, W), False)
        if save_for_backward:
            self._saved = dict(stem=saved, head=head, B=B)
        return out.view(B, T, self.head_out)

    @property
    def head_accumulates_in_place(self) -> bool:
        """True when backward(accumulate_head=True) really accumulates: a k x k head whose GEMM writes the gradient directly."""
        return self.head_k != 1 and (self.head_k * self.head_k * 1024) % 64 == 0

    def backward(self, dtokens: torch.Tensor, accumulate_head: bool = False) -> None:
        """dtokens [B,T,head_out] f32.  Writes weight/bias gradients into self.g (overwrites).  accumulate_head: the weight
        gradient of a k x k head without bias column (the VETokenizer's 105 M-weight conv: 91 % of the model's trainables) is ADDED
        to what self.g holds -- the GEMM's residual operand, in place -- instead of overwriting it (gradient accumulation windows,
        base_task.py:262-271: no copy of the 420 MB segment is made, MyriadHIP.backward)."""
        sv = self._saved
        if sv is None:
            raise RuntimeError("backward() without saved forward")
        B = sv["B"]
        dy32 = dtokens.reshape(-1, self.head_out).contiguous()
        dyb = ops.to_bf16(dy32)
        gw, gb = self.g[f"{self.prefix}meta_net.15.weight"], self.g[f"{self.prefix}meta_net.15.bias"]
        a, wq, hw = sv["head"][:3]
        dyT = ops.transpose_to_bf16(dyb, 64)
        if self.head_k == 1:
            ops.gemm_auto_f32(dyT, ops.transpose_to_bf16(a, 64), gw)
            gb.copy_(ops.colsum(dy32))
            dp = ops.gemm(dyb, ops.transpose_to_bf16(wq, 64), out_dtype=F32)          # [B*49, 1024] f32 == NHWC
        else:
            Kpad = wq.shape[1]
            if sv["head"][3]:                                      # no bias column: the product IS the weight gradient
                if accumulate_head:
                    ops.gemm(dyT, ops.transpose_to_bf16(a, 64), out=gw, residual=gw)      # gw += dy^T . col (C may alias residual)
                else:
                    ops.gemm_auto_f32(dyT, ops.transpose_to_bf16(a, 64), gw)
                gb.copy_(ops.colsum(dy32))
            else:
                dwp = torch.empty((self.head_out, Kpad), dtype=F32, device=self.dev)
                ops.gemm_auto_f32(dyT, ops.transpose_to_bf16(a, 64), dwp)
                ops.conv_unpack_grad(dwp, gw, gb)
            dcol = ops.gemm(dyb, ops.transpose_to_bf16(wq, 64))                          # [B*T, Kpad] bf16
            H, W = hw
            dp = ops.col2im(dcol, B, H, W, 1024, self.head_k, self.head_k, 0)
        for li in range(4, -1, -1):
            col, y, wp, H, W, ci, co = sv["stem"][li]
            idx = STEM_IDX[li]
            dy, dyfull = ops.relu_pool_bwd(dp, y, B, H, W, co, pad_cols_to=64)
            dwp = torch.empty((co, wp.shape[1]), dtype=F32, device=self.dev)
            ops.gemm_auto_f32(ops.transpose_to_bf16(dy, 64), ops.transpose_to_bf16(col, 64), dwp)
            ops.conv_unpack_grad(dwp, self.g[f"{self.prefix}meta_net.{idx}.weight"],
                                 self.g[f"{self.prefix}meta_net.{idx}.bias"])
            if li > 0:
                wpT = ops.transpose_to_bf16(wp, 64)            # [Kpad, cpad] (rows >= co zero)
                dcol = ops.gemm(dyfull, wpT)                    # [M, Kpad] bf16
                dp = ops.col2im(dcol, B, H, W, ci, 3, 3, 1)
        self._saved = None


class LoraAdaptor:
    """y = x + conv2(conv1(x)) on the fp32 image-token stream (reference networks.py:81-93)."""

    def __init__(self, params, grads, prefix="expert_adaptor."):
        self.A, self.Bm = params[prefix + "conv1.weight"], params[prefix + "conv2.weight"]
        self.gA, self.gB = grads[prefix + "conv1.weight"], grads[prefix + "conv2.weight"]
        self._saved = None

    def forward(self, x2d: torch.Tensor, save_for_backward=True):
        y, t = ops.lowrank_fwd(x2d, self.A, self.Bm)
        if save_for_backward:
            self._saved = (x2d, t)
        return y

    def backward(self, dy2d: torch.Tensor):
        x2d, t = self._saved
        ops.lowrank_bwd(dy2d, x2d, t, self.A, self.Bm, self.gA, self.gB, need_dx=False)
        self._saved = None

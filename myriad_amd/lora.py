"""PEFT-style LoRA on q_proj / v_proj of the LLaMA decoder (reference minigpt4/models/myriad.py:170-180,198-200:
LoraConfig(r=8, lora_alpha=16, lora_dropout=0.05, target_modules=["q_proj","v_proj"]); the arithmetic lives in the
un-vendored `peft` package -- restated from its published form  y = W x + (alpha/r) * B(A(dropout(x))),
A ~ kaiming-uniform, B = 0 at init.  SURVEY 8 a-14: parity unpinned by the reference, pinned here to the oracle.)

MI355X formulation -- the rank-r UP projection rides the main MFMA GEMMs as a K-border instead of separate kernels:

    forward   [ x | s*t ] . [ W_qkv | B_ext ]^T          t = dropout(x) A_qv^T        (K = 4096 + 64)
    backward  dqkv . [ W_qkv | B_ext ]  ->  [ dx_base | d(s*t) ]   in ONE dgrad GEMM

so q/k/v are rounded to bf16 once and no extra pass over the [B*S, 12288] qkv buffer is needed; the skinny parts
(lora-down, dx correction, dA/dB) are wavefront-primitive kernels in csrc/lora.hip that regenerate the dropout mask
from (seed, index) instead of storing it -- one mask per wrapped Linear (peft gives q_proj and v_proj their own nn.Dropout).
The 64-column border holds 64 / 2r groups of [s*t_q (r) | s*t_v (r)]: lora_down splits D over the groups (4x the workgroups of a
latency-bound kernel) and the weight border repeats [B_q | B_v] per group, so the GEMM adds the partial products up.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch

from . import _lib, ops

BF16, F32 = torch.bfloat16, torch.float32
BORDER = 64
V_TAG = 1 << 63      # v_proj's dropout mask is the second draw of q_proj's hash: seed | bit 63 (csrc/common.h dropout_keep_pair); seeds are 63-bit
PEFT_PREFIX = "llama_model.base_model.model.model.layers."


def lora_param_specs(n_layers: int, D: int, r: int) -> List[Tuple[str, Tuple[int, ...], Tuple[int, ...]]]:
    """Reference (PEFT) state_dict names; per layer the order A_q, A_v, B_q, B_v keeps A_q|A_v contiguous."""
    specs = []
    for i in range(n_layers):
        p = f"{PEFT_PREFIX}{i}.self_attn."
        specs.append((p + "q_proj.lora_A.default.weight", (r, D), (r, D)))
        specs.append((p + "v_proj.lora_A.default.weight", (r, D), (r, D)))
        specs.append((p + "q_proj.lora_B.default.weight", (D, r), (D, r)))
        specs.append((p + "v_proj.lora_B.default.weight", (D, r), (D, r)))
    return specs


def init_lora_weights(n_layers: int, D: int, r: int, seed: int, device, zero_b: bool = True) -> Dict[str, torch.Tensor]:
    """peft init: lora_A kaiming_uniform(a=sqrt(5)) => U(-1/sqrt(D), 1/sqrt(D)); lora_B zeros."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = {}
    bound = 1.0 / math.sqrt(D)
    for name, ishape, _ in lora_param_specs(n_layers, D, r):
        if "lora_A" in name:
            out[name] = ((torch.rand(ishape, generator=g) * 2 - 1) * bound).to(device)
        else:
            out[name] = (torch.zeros(ishape) if zero_b else torch.randn(ishape, generator=g) * 0.02).to(device)
    return out


class LoraQV:
    def __init__(self, n_layers: int, D: int, r: int, alpha: float, dropout: float, params: Dict[str, torch.Tensor],
                 grads: Dict[str, torch.Tensor], device):
        if r not in (8, 16):
            raise ValueError("LoRA rank must be 8 or 16")
        self.L, self.D, self.r, self.s, self.p = n_layers, D, r, alpha / r, dropout
        self.dev = torch.device(device)
        self.P, self.G = params, grads
        self.step_seed = 0
        self.base_seed = 0          # run seed + rank (train.py:63-72): ranks and runs draw different dropout masks
        self._xext: Dict[Tuple[int, int], torch.Tensor] = {}
        self._ws = torch.empty((_lib.load().mh_lora_wgrad_ws_floats(D, 2 * r),), dtype=F32, device=self.dev)
        self._deferred, self._side, self._wgrad_ev = [], None, None

    def names(self, i):
        p = f"{PEFT_PREFIX}{i}.self_attn."
        return (p + "q_proj.lora_A.default.weight", p + "v_proj.lora_A.default.weight",
                p + "q_proj.lora_B.default.weight", p + "v_proj.lora_B.default.weight")

    def _aqv(self, store, i):
        """[2r, D] fp32 view over the adjacent A_q | A_v segments of a flat buffer dict."""
        aq, av = store[self.names(i)[0]], store[self.names(i)[1]]
        assert av.data_ptr() == aq.data_ptr() + aq.numel() * 4, "A_q and A_v must be adjacent in the flat buffer"
        return torch.as_strided(aq, (2 * self.r, self.D), (self.D, 1))

    def extend_weights(self, layer: dict) -> None:
        """Allocate the bordered copies of the frozen qkv weight (once)."""
        W3, D = layer["wqkv"].shape
        ext = torch.zeros((W3, D + BORDER), dtype=BF16, device=self.dev)
        ext[:, :D].copy_(layer["wqkv"])
        layer["wqkv_ext"] = ext
        if layer.get("wqkvT") is not None:
            extT = torch.zeros((D + BORDER, W3), dtype=BF16, device=self.dev)
            extT[:D].copy_(layer["wqkvT"])
            layer["wqkvT_ext"] = extT
        layer["wqkv"] = layer["wqkvT"] = None     # the bordered copies replace them

    def x_ext(self, layer_idx: int, M: int) -> torch.Tensor:
        """Persistent bordered activation buffer [M, D+64] bf16 per layer (columns >= D+2r stay zero forever)."""
        key = (layer_idx, M)
        if key not in self._xext:
            self._xext[key] = torch.zeros((M, self.D + BORDER), dtype=BF16, device=self.dev)
        return self._xext[key]

    def refresh(self, layers: List[dict]) -> None:
        """Per optimisation step: push the current fp32 B_q / B_v into the bf16 borders of W_ext and W_ext^T -- one launch for
        all layers over a device table of their (fixed) addresses."""
        D, r, W = self.D, self.r, self.D
        key = tuple(id(Lr) for Lr in layers)
        if getattr(self, "_refresh_key", None) != key:
            ptrs, lds = [], set()
            for i, Lr in enumerate(layers):
                _, _, nbq, nbv = self.names(i)
                ext, extT = Lr["wqkv_ext"], Lr.get("wqkvT_ext")
                ptrs += [self.P[nbq].data_ptr(), self.P[nbv].data_ptr(), ext.data_ptr(), 0 if extT is None else extT.data_ptr()]
                lds.add((ext.stride(0), 0 if extT is None else extT.stride(0)))
            if len(lds) != 1:
                raise _lib.MyriadHipError("lora.refresh: layers with different W_ext strides")
            self._refresh_tab = torch.tensor(ptrs, dtype=torch.int64).to(self.dev)
            self._refresh_ld, self._refresh_key = lds.pop(), key
        _lib.check(_lib.load().mh_lora_refresh_borders(self._refresh_tab.data_ptr(), len(layers), self._refresh_ld[0],
                                                       self._refresh_ld[1], W, D, r, ops._s()), "mh_lora_refresh_borders")

    def _seed(self, layer_idx: int) -> int:
        return (self.base_seed * 0x9E3779B97F4A7C15 + self.step_seed * 1315423911 + layer_idx * 2654435761 + 12345) \
            & 0x7FFFFFFFFFFFFFFF

    # ---- forward: border = s * dropout(x) A_qv^T ------------------------------------------------------------------
    def forward_border(self, layer_idx: int, x_ext: torch.Tensor, training: bool = True):
        D, r = self.D, self.r
        p = self.p if training else 0.0
        seed = self._seed(layer_idx)
        A = self._aqv(self.P, layer_idx)
        M = x_ext.shape[0]
        _lib.check(_lib.load().mh_lora_down(x_ext.data_ptr(), x_ext.stride(0), A.data_ptr(), x_ext[:, D:].data_ptr(),
                                            x_ext.stride(0), M, D, 2 * r, self.s, p, seed, ops._s()), "mh_lora_down")
        return p, seed

    def norm_border(self, layer_idx: int, h: torch.Tensor, norm_w: torch.Tensor, eps: float, x_ext: torch.Tensor) -> bool:
        """Single-token decode (<= 2 rows, no dropout): x_ext <- [bf16(rmsnorm(h)) | border] in ONE launch, the bits of
        ops.rmsnorm_fwd(out=x_ext[:, :D]) + forward_border(training=False).  False: not supported for this shape, nothing written."""
        rc = _lib.load().mh_rmsnorm_lora_down(h.data_ptr(), h.stride(0), norm_w.data_ptr(), float(eps),
                                              self._aqv(self.P, layer_idx).data_ptr(), x_ext.data_ptr(), x_ext.stride(0), h.shape[0],
                                              self.D, 2 * self.r, self.s, ops._s())
        if rc == -3:                                   # MH_ERR_UNSUPPORTED
            return False
        _lib.check(rc, "mh_rmsnorm_lora_down")
        return True

    # ---- backward ------------------------------------------------------------------------------------------------
    def backward(self, layer_idx: int, dx_ext: torch.Tensor, dqkv: torch.Tensor, x_ext: torch.Tensor, p: float,
                 seed: int, defer_wgrad: bool = False) -> torch.Tensor:
        """dx_ext [M, D+64] f32 = dqkv . [W | B_ext];  returns the full d(xn) [M, D] f32 (contiguous) and writes
        dA_q, dA_v, dB_q, dB_v into the flat gradient buffer.  The weight gradients feed nothing but the optimiser, so with
        defer_wgrad they are only queued: run_deferred_wgrads() launches them on a side stream beside whatever the caller
        does next (the Q-Former / adapter backward) and join_wgrads() makes the current stream wait for them."""
        self._queue_wgrad(layer_idx, (dx_ext, dx_ext.data_ptr(), dx_ext.stride(0)), dqkv, x_ext, p, seed, defer_wgrad)
        D, r = self.D, self.r
        M = dx_ext.shape[0]
        A = self._aqv(self.P, layer_idx)
        dxn = torch.empty((M, D), dtype=F32, device=self.dev)
        _lib.check(_lib.load().mh_lora_dx(dx_ext.data_ptr(), dx_ext.stride(0), A.data_ptr(), dxn.data_ptr(), M, D, 2 * r, self.s,
                                          p, seed, ops._s()), "mh_lora_dx")
        return dxn

    def backward_from_dqkv(self, layer_idx: int, dqkv: torch.Tensor, wqkvT_ext: torch.Tensor, x_ext: torch.Tensor, p: float,
                           seed: int, defer_wgrad: bool = False) -> torch.Tensor:
        """The dgrad GEMM dqkv . [W | B_ext] and backward() in one library call (mh_gemm_lora_dx): when the GEMM policy splits
        K -- it does at the training shapes -- the dx kernel sums the fp32 partial slabs itself, so neither the reduce launch
        nor the [M, D+64] fp32 product exists; only the summed 64-column border is kept for the weight gradients."""
        D, r = self.D, self.r
        M, K = dqkv.shape
        _, splits = ops.gemm_plan(M, D + BORDER, K)
        A = self._aqv(self.P, layer_idx)
        dxn = torch.empty((M, D), dtype=F32, device=self.dev)
        if splits > 1:
            border = torch.empty((M, BORDER), dtype=F32, device=self.dev)
            keep, gptr, ldg, buf_ptr = border, border.data_ptr() - 4 * D, BORDER, None     # the kernels address the border at column D
        else:
            buf = torch.empty((M, D + BORDER), dtype=F32, device=self.dev)
            keep, gptr, ldg, buf_ptr, border = buf, buf.data_ptr(), buf.stride(0), buf.data_ptr(), None
        _lib.check(_lib.load().mh_gemm_lora_dx(dqkv.data_ptr(), dqkv.stride(0), wqkvT_ext.data_ptr(), wqkvT_ext.stride(0), buf_ptr,
                                               A.data_ptr(), dxn.data_ptr(), None if border is None else border.data_ptr(), M, D, K,
                                               2 * r, self.s, p, seed, ops._s()), f"mh_gemm_lora_dx M={M} K={K}")
        self._queue_wgrad(layer_idx, (keep, gptr, ldg), dqkv, x_ext, p, seed, defer_wgrad)
        return dxn

    def backward_from_dqkv_norm(self, layer_idx: int, dqkv: torch.Tensor, wqkvT_ext: torch.Tensor, x_ext: torch.Tensor, p: float,
                                seed: int, h_in: torch.Tensor, norm_w: torch.Tensor, eps: float, dres: torch.Tensor,
                                defer_wgrad: bool = False, want_bf16: bool = True):
        """backward_from_dqkv() and the backward of the layer's input RMSNorm (modeling_llama.py:257-259 under autograd) in one
        library call (mh_gemm_lora_rmsnorm_bwd): at D <= 4096, r = 8 the LoRA dx correction and the norm backward are one kernel
        that sums the dgrad's split-K slabs itself, so d(xn) [M, D] is never written.  Returns (dh f32, dh bf16) -- the gradient
        of the residual stream below this layer."""
        D, r = self.D, self.r
        M, K = dqkv.shape
        _, splits = ops.gemm_plan(M, D + BORDER, K)
        A = self._aqv(self.P, layer_idx)
        lib = _lib.load()
        # the library's own condition (lora.hip: mh_launch_lora_dx_rmsnorm_bwd); an unknown option name reads as a negative error code
        fused = lib.mh_get_option(b"lora_norm_fused") == 1 and D <= 4096 and D % 256 == 0 and 2 * r == 16
        dxn = None if fused else torch.empty((M, D), dtype=F32, device=self.dev)
        if splits > 1:
            border = torch.empty((M, BORDER), dtype=F32, device=self.dev)
            keep, gptr, ldg, buf_ptr = border, border.data_ptr() - 4 * D, BORDER, None
        else:
            buf = torch.empty((M, D + BORDER), dtype=F32, device=self.dev)
            keep, gptr, ldg, buf_ptr, border = buf, buf.data_ptr(), buf.stride(0), buf.data_ptr(), None
        dh = torch.empty((M, D), dtype=F32, device=self.dev)
        dhb = torch.empty((M, D), dtype=BF16, device=self.dev) if want_bf16 else None
        _lib.check(lib.mh_gemm_lora_rmsnorm_bwd(dqkv.data_ptr(), dqkv.stride(0), wqkvT_ext.data_ptr(), wqkvT_ext.stride(0), buf_ptr,
                                                A.data_ptr(), None if dxn is None else dxn.data_ptr(),
                                                None if border is None else border.data_ptr(), h_in.data_ptr(), norm_w.data_ptr(),
                                                dres.data_ptr(), dh.data_ptr(), None if dhb is None else dhb.data_ptr(), M, D, K,
                                                2 * r, self.s, p, seed, float(eps), ops._s()),
                   f"mh_gemm_lora_rmsnorm_bwd M={M} K={K}")
        self._queue_wgrad(layer_idx, (keep, gptr, ldg), dqkv, x_ext, p, seed, defer_wgrad)
        return dh, dhb

    def _queue_wgrad(self, layer_idx, g, dqkv, x_ext, p, seed, defer_wgrad) -> None:
        if defer_wgrad:
            self._deferred.append((layer_idx, g, dqkv, x_ext, p, seed))
        else:
            self._wgrad(layer_idx, g, dqkv, x_ext, p, seed)

    def _wgrad(self, layer_idx, g, dqkv, x_ext, p, seed) -> None:
        """g = (tensor that owns the memory, address of a [M, >= D+2r] f32 view whose columns D.. hold d(s*t), its row stride)."""
        D, r, W = self.D, self.r, self.D
        M = dqkv.shape[0]
        _, _, nb_q, nb_v = self.names(layer_idx)
        gA = self._aqv(self.G, layer_idx)
        dv = dqkv[:, 2 * W:]
        _lib.check(_lib.load().mh_lora_wgrad(x_ext.data_ptr(), x_ext.stride(0), g[1], g[2],
                                             dqkv.data_ptr(), dv.data_ptr(), dqkv.stride(0), x_ext[:, D:].data_ptr(),
                                             x_ext.stride(0), gA.data_ptr(), self.G[nb_q].data_ptr(), self.G[nb_v].data_ptr(),
                                             self._ws.data_ptr(), M, D, 2 * r, self.s, p, seed, ops._s()), "mh_lora_wgrad")

    def run_deferred_wgrads(self) -> None:
        """Launch the queued weight gradients on the side stream (ordered after everything the current stream has queued)."""
        if not self._deferred:
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            for (li, g, dqkv, x_ext, p, seed) in self._deferred:
                for t in (g[0], dqkv):
                    t.record_stream(self._side)          # allocated on the main stream, read here
                self._wgrad(li, g, dqkv, x_ext, p, seed)
            self._wgrad_ev = torch.cuda.Event()
            self._wgrad_ev.record()
        self._deferred = []

    def join_wgrads(self) -> None:
        if self._deferred:                               # never launched: run them here, in order
            for item in self._deferred:
                self._wgrad(*item)
            self._deferred = []
        if self._wgrad_ev is not None:
            torch.cuda.current_stream().wait_event(self._wgrad_ev)
            self._wgrad_ev = None

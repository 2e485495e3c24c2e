"""Thin tensor->pointer wrappers over the C ABI (include/myriad_hip.h).  torch is used only for device memory
and the current HIP stream; all arithmetic happens in libmyriad_hip.so.  No fallbacks."""
from __future__ import annotations

import math
from typing import Optional

import os

import torch

from . import _lib

BF16 = torch.bfloat16
F32 = torch.float32

GEMM_OUT_F32 = 1
GEMM_GELU = 2
GEMM_REGSTAGE = 4
GEMM_VARIANT = 0      # 0 = library default; 1..5 force a kernel variant (tools/gemm_bench.py A/B tests)


def _L():
    return _lib.load()


_DEV_IDX = None


def _s() -> int:
    """Raw handle of torch's current HIP stream.  `torch.cuda.current_stream().cuda_stream` builds a Stream object per
    call (~8 us, 1200 times per step = the largest single host cost of a batch-1 step); the raw getter is a C call."""
    global _DEV_IDX
    if _DEV_IDX is None:
        _DEV_IDX = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(_DEV_IDX)


class _PinnedRing:
    """Small host->device uploads (index vectors, lengths, labels) without stalling the launch thread: `.to(device)` from
    pageable memory blocks until the stream drains (12 such stalls per step cost ~14 ms at batch 1).  The payload is
    staged in a slot of a pinned ring and copied with non_blocking=True; a slot is reused only after its copy event."""

    def __init__(self, slots: int = 64, slot_bytes: int = 1 << 16):
        self.buf = torch.empty((slots, slot_bytes), dtype=torch.uint8).pin_memory()
        self.events = [None] * slots
        self.slots, self.slot_bytes, self.i = slots, slot_bytes, 0

    def upload(self, t: torch.Tensor, device) -> torch.Tensor:
        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        if nbytes == 0 or nbytes > self.slot_bytes or t.is_cuda:
            return t.to(device)
        k = self.i
        self.i = (k + 1) % self.slots
        if self.events[k] is not None:
            self.events[k].synchronize()
        stage = self.buf[k, :nbytes].view(t.dtype).view(t.shape)
        stage.copy_(t)
        out = torch.empty(t.shape, dtype=t.dtype, device=device)
        out.copy_(stage, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        return out


_RING = None


def h2d(t: torch.Tensor, device) -> torch.Tensor:
    """Asynchronous upload of a small host tensor (see _PinnedRing)."""
    global _RING
    if _RING is None:
        _RING = _PinnedRing()
    return _RING.upload(t, device)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk2d(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1:
        raise _lib.MyriadHipError(f"{name}: need cuda {dtype} 2-D tensor with unit inner stride, got "
                                  f"{t.dtype} {tuple(t.shape)} strides {t.stride()} cuda={t.is_cuda}")


_WS = {}


def ensure_workspace(device, nbytes: int = 256 << 20):
    """Register a split-K scratch buffer with the library (once per device)."""
    key = str(device)
    if key not in _WS:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _lib.check(_L().mh_set_workspace(buf.data_ptr(), nbytes), "mh_set_workspace")
        _WS[key] = buf
    return _WS[key]


_SIDE = {}


def side_stream(device, name: str) -> "torch.cuda.Stream":
    """A named side stream of the device with its OWN split-K scratch (two streams that both split K must not share one;
    mh_set_stream_workspace).  One per (device, name) for the whole process: every model instance shares them."""
    key = (str(device), name)
    if key not in _SIDE:
        main_ws = ensure_workspace(device)
        st = torch.cuda.Stream(device=device)
        ws = torch.empty(main_ws.numel(), dtype=torch.uint8, device=device)
        _lib.check(_L().mh_set_stream_workspace(st.cuda_stream, ws.data_ptr(), ws.numel()), "mh_set_stream_workspace")
        _SIDE[key] = (st, ws)
    return _SIDE[key][0]


def drop_workspace():
    """Unregister the split-K scratch (tests: the no-workspace paths).  ensure_workspace() registers it again."""
    torch.cuda.synchronize()
    _lib.check(_L().mh_set_workspace(None, 0), "mh_set_workspace")     # also drops every side-stream scratch
    _WS.clear()
    _SIDE.clear()


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# --------------------------------------------------------------------------- GEMM
# One GEMM backend: every product below is the library's own kernels.  (The vendor-library comparison lives in
# tools/gemm_vendor_calib.py, outside the product.)


def gemm(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
         residual: Optional[torch.Tensor] = None, out_dtype=BF16, gelu: bool = False, alpha: float = 1.0,
         regstage: bool = False, variant: int = -1) -> torch.Tensor:
    """out[M,N] = alpha * a[M,K] @ b[N,K]^T (+bias) (gelu) (+residual f32)."""
    _chk2d(a, BF16, "gemm.a")
    _chk2d(b, BF16, "gemm.b")
    M, K = a.shape
    N, K2 = b.shape
    if K != K2:
        raise _lib.MyriadHipError(f"gemm: K mismatch {K} vs {K2}")
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    else:
        _chk2d(out, out.dtype, "gemm.out")
        if out.shape != (M, N):
            raise _lib.MyriadHipError(f"gemm: out shape {tuple(out.shape)} != {(M, N)}")
    flags = (GEMM_OUT_F32 if out.dtype == F32 else 0) | (GEMM_GELU if gelu else 0) | (GEMM_REGSTAGE if regstage else 0)
    flags |= (GEMM_VARIANT if variant < 0 else variant) << 8
    if bias is not None and (bias.dtype != F32 or bias.numel() != N):
        raise _lib.MyriadHipError("gemm: bias must be f32 [N]")
    ldr = 0
    if residual is not None:
        _chk2d(residual, F32, "gemm.residual")
        ldr = residual.stride(0)
    rc = _L().mh_gemm_bf16_nt(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K, _p(bias),
                              _p(residual), ldr, flags, float(alpha), _s())
    _lib.check(rc, f"mh_gemm_bf16_nt M={M} N={N} K={K}")
    return out


GEMM_KERNEL_NAMES = {0: "gemv_kernel", 1: "gemm_nt_kernel", 2: "gemm_x8_kernel", 3: "gemm_nt_kernel", 4: "gemm_nt_kernel",
                     5: "gemm_nt_kernel", 6: "gemm_nt_kernel"}   # 3: 128x64 tile; 4 / 5: 160x128 / 160x96 row tile (M <= 320); 6: 64x64, 8-deep ring


def gemm_plan(M: int, N: int, K: int, out_f32: bool = False, gelu: bool = False):
    """(kernel id, K splits) the library will use for this shape -- mh_gemm_plan."""
    import ctypes
    kernel, splits = ctypes.c_int(0), ctypes.c_int(0)
    flags = (1 if out_f32 else 0) | (2 if gelu else 0)
    _lib.check(_L().mh_gemm_plan(M, N, K, flags, ctypes.addressof(kernel), ctypes.addressof(splits)), "mh_gemm_plan")
    return kernel.value, splits.value


class PackedWeight:
    """Stream-ordered copy of a frozen [N, K] bf16 weight for the decode kernel (mh_gemv_pack)."""
    __slots__ = ("data", "N", "K")

    def __init__(self, data: torch.Tensor, N: int, K: int):
        self.data, self.N, self.K = data, N, K


def gemv_pack(w: torch.Tensor, out: Optional[PackedWeight] = None) -> PackedWeight:
    """Permute w [N, K] into the order the skinny-M kernel streams it in; `out` re-uses an earlier copy's storage."""
    _chk2d(w, BF16, "gemv_pack.w")
    N, K = w.shape
    n = _L().mh_gemv_pack_elems(N, K)
    if n < 0:
        raise _lib.MyriadHipError(f"gemv_pack: unsupported dims N={N} K={K}")
    if out is None:
        out = PackedWeight(torch.empty((n,), dtype=BF16, device=w.device), N, K)
    elif (out.N, out.K) != (N, K):
        raise _lib.MyriadHipError("gemv_pack: out was packed for another shape")
    _lib.check(_L().mh_gemv_pack(_p(w), w.stride(0), N, K, _p(out.data), _s()), "mh_gemv_pack")
    return out


def gemv_packed(a: torch.Tensor, pw: PackedWeight, out: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None, out_dtype=BF16, alpha: float = 1.0) -> torch.Tensor:
    """out[M<=16, N] = alpha * a @ W^T (+bias) (+residual f32) with W given as its packed copy; same bits as gemm()."""
    _chk2d(a, BF16, "gemv_packed.a")
    M, K = a.shape
    if K != pw.K or M > 16:
        raise _lib.MyriadHipError(f"gemv_packed: a is {tuple(a.shape)}, weight was packed as [{pw.N}, {pw.K}], M must be <= 16")
    if out is None:
        out = torch.empty((M, pw.N), dtype=out_dtype, device=a.device)
    ldr = 0
    if residual is not None:
        _chk2d(residual, F32, "gemv_packed.residual")
        ldr = residual.stride(0)
    rc = _L().mh_gemv_packed(_p(a), a.stride(0), _p(pw.data), _p(out), out.stride(0), M, pw.N, K, _p(bias), _p(residual), ldr,
                             1 if out.dtype == F32 else 0, float(alpha), _s())
    _lib.check(rc, f"mh_gemv_packed M={M} N={pw.N} K={K}")
    return out


def _gemv_pro(fn, name, a, lda, pw, out, residual, out_dtype, alpha, M, *pre):
    if out is None:
        out = torch.empty((M, pw.N), dtype=out_dtype, device=a.device)
    ldr = 0
    if residual is not None:
        _chk2d(residual, F32, name + ".residual")
        ldr = residual.stride(0)
    rc = fn(_p(a), lda, *pre, _p(pw.data), _p(out), out.stride(0), M, pw.N, pw.K, None, _p(residual), ldr,
            1 if out.dtype == F32 else 0, float(alpha), _s())
    if rc == -3:                                   # MH_ERR_UNSUPPORTED: the operand rows do not fit the kernel's LDS budget
        return None
    _lib.check(rc, f"{name} M={M} N={pw.N} K={pw.K}")
    return out


def gemv_packed_rmsnorm(h: torch.Tensor, norm_w: torch.Tensor, eps: float, pw: PackedWeight, out=None, residual=None,
                        out_dtype=BF16, alpha: float = 1.0):
    """out[M<=16, N] = alpha * rmsnorm(h; norm_w, eps) @ W^T (+residual): mh_rmsnorm_fwd + mh_gemv_packed in one launch, same bits.
    Returns None when the operand rows exceed the fused kernel's LDS budget (run the two launches instead)."""
    _chk2d(h, F32, "gemv_packed_rmsnorm.h")
    M, K = h.shape
    if K != pw.K or M > 16:
        raise _lib.MyriadHipError(f"gemv_packed_rmsnorm: h is {tuple(h.shape)}, weight was packed as [{pw.N}, {pw.K}]")
    return _gemv_pro(_L().mh_gemv_packed_rmsnorm, "mh_gemv_packed_rmsnorm", h, h.stride(0), pw, out, residual, out_dtype, alpha, M,
                     _p(norm_w), float(eps))


def gemv_packed_silu(gu: torch.Tensor, pw: PackedWeight, out=None, residual=None, out_dtype=BF16, alpha: float = 1.0):
    """out[M<=16, N] = alpha * (silu(g) * u) @ W^T (+residual) for gu [M, 2K] bf16 in the 128-blocked gate|up layout:
    mh_silu_mul_fwd_blk + mh_gemv_packed in one launch, same bits.  None when the rows exceed the LDS budget."""
    _chk2d(gu, BF16, "gemv_packed_silu.gu")
    M = gu.shape[0]
    if gu.shape[1] != 2 * pw.K or M > 16:
        raise _lib.MyriadHipError(f"gemv_packed_silu: gu is {tuple(gu.shape)}, weight was packed as [{pw.N}, {pw.K}]")
    return _gemv_pro(_L().mh_gemv_packed_silu, "mh_gemv_packed_silu", gu, gu.stride(0), pw, out, residual, out_dtype, alpha, M)


def attn_decode_rope(qkv2d: torch.Tensor, cache: torch.Tensor, pos: torch.Tensor, pos_dev: torch.Tensor, kv_len: torch.Tensor,
                     cos_tab: torch.Tensor, sin_tab: torch.Tensor, n_heads: int, head_dim: int, scale: float):
    """One decode token: rope_kv_append + attn_fwd(Sq = 1) as one launch (same bits).  qkv2d [B, >=3W] bf16 (q rotated in place),
    cache [B, T, 2W]; returns o [B, W] bf16."""
    _chk2d(qkv2d, BF16, "attn_decode_rope.qkv")
    B, W = qkv2d.shape[0], n_heads * head_dim
    if qkv2d.shape[1] < 3 * W or cache.shape[2] != 2 * W:
        raise _lib.MyriadHipError("attn_decode_rope: qkv must be [B, >=3W] and cache [B, T, 2W]")
    out = torch.empty((B, W), dtype=BF16, device=qkv2d.device)
    _lib.check(_L().mh_attn_decode_rope(_p(qkv2d), qkv2d.stride(0), _p(cache), cache.stride(0), cache.stride(1), _p(pos), _p(pos_dev),
                                        _p(kv_len), _p(cos_tab), _p(sin_tab), _p(out), out.stride(0), B, n_heads, head_dim,
                                        cache.shape[1], float(scale), _s()), "mh_attn_decode_rope")
    return out


def gemm_auto_f32(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """f32 out = a @ b^T, choosing split-K when the output is small and the reduction long (wgrad shapes)."""
    M, K = a.shape
    N = b.shape[0]
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    kt = K // 64
    if tiles >= 128 or kt < 16:
        return gemm(a, b, out=out)
    splits = max(1, min(kt // 4, 512 // tiles))
    ws = torch.empty((_L().mh_gemm_splitk_ws_floats(M, N, splits),), dtype=F32, device=a.device)
    rc = _L().mh_gemm_bf16_nt_splitk(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K, splits,
                                     _p(ws), _s())
    _lib.check(rc, f"mh_gemm_bf16_nt_splitk M={M} N={N} K={K} splits={splits}")
    return out


# --------------------------------------------------------------------------- attention
def attn_fwd(q, k, v, H: int, D: int, scale: float, causal: bool = False, bias=None, kv_len=None, out=None,
             need_lse: bool = True):
    """q [B,Sq,Wq], k/v [B,Sk,W*] bf16 views (head h at cols h*D..), returns (o [B,Sq,H*D] bf16, lse [B,H,Sq] f32)."""
    B, Sq = q.shape[0], q.shape[1]
    Sk = k.shape[1]
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        if t.dtype != BF16 or t.stride(2) != 1:
            raise _lib.MyriadHipError(f"attn_fwd.{n}: need bf16 with unit inner stride")
    if out is None:
        out = torch.empty((B, Sq, H * D), dtype=BF16, device=q.device)
    lse = torch.empty((B, H, Sq), dtype=F32, device=q.device) if need_lse else None
    rc = _L().mh_attn_fwd(_p(q), _p(k), _p(v), _p(out), _p(lse), _p(bias), _p(kv_len), B, H, Sq, Sk, D,
                          q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
                          out.stride(0), out.stride(1), float(scale), int(causal), _s())
    _lib.check(rc, f"mh_attn_fwd B={B} H={H} Sq={Sq} Sk={Sk} D={D}")
    return out, lse


def attn_bwd(q, k, v, o, dout, lse, H: int, D: int, scale: float, causal: bool = False, bias=None, kv_len=None,
             dq=None, dk=None, dv=None):
    B, Sq = q.shape[0], q.shape[1]
    Sk = k.shape[1]
    dev = q.device
    if dq is None:
        dq = torch.empty((B, Sq, H * D), dtype=BF16, device=dev)
    if dk is None:
        dk = torch.empty((B, Sk, H * D), dtype=BF16, device=dev)
    if dv is None:
        dv = torch.empty((B, Sk, H * D), dtype=BF16, device=dev)
    delta = torch.empty((B, H, Sq), dtype=F32, device=dev)
    rc = _L().mh_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(dout), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), _p(bias),
                          _p(kv_len), B, H, Sq, Sk, D, q.stride(0), q.stride(1), k.stride(0), k.stride(1),
                          v.stride(0), v.stride(1), o.stride(0), o.stride(1), dout.stride(0), dout.stride(1),
                          dq.stride(0), dq.stride(1), dk.stride(0), dk.stride(1), dv.stride(0), dv.stride(1),
                          float(scale), int(causal), _s())
    _lib.check(rc, f"mh_attn_bwd B={B} H={H} Sq={Sq} Sk={Sk} D={D}")
    return dq, dk, dv


def attn_rope_supported(S: int, D: int) -> bool:
    """Whether the whole-sequence fused rotary attention (mh_attn_rope_*) covers this shape."""
    return bool(_L().mh_attn_rope_supported(int(S), int(D)))


def _chk_qkv3(qkv3, H, D, name):
    if qkv3.dtype != BF16 or qkv3.dim() != 3 or qkv3.stride(2) != 1 or qkv3.stride(0) != qkv3.shape[1] * qkv3.stride(1):
        raise _lib.MyriadHipError(f"{name}: need bf16 [B, S, >=3*H*D] with unit inner stride and dense batches")
    if qkv3.shape[2] < 3 * H * D:
        raise _lib.MyriadHipError(f"{name}: last dim {qkv3.shape[2]} < 3*H*D")


def attn_rope_fwd(qkv3: torch.Tensor, H: int, D: int, scale: float, pos, cos, sin, kv_len=None, need_lse: bool = True):
    """LLaMA causal self-attention with rotary applied on load.  qkv3 [B, S, >=3W] bf16 = [q | k | v], PRE-rotary.
    Returns (o [B, S, W] bf16, lse [B, H, S] f32)."""
    _chk_qkv3(qkv3, H, D, "attn_rope_fwd")
    B, S = qkv3.shape[0], qkv3.shape[1]
    o = torch.empty((B, S, H * D), dtype=BF16, device=qkv3.device)
    lse = torch.empty((B, H, S), dtype=F32, device=qkv3.device) if need_lse else None
    rc = _L().mh_attn_rope_fwd(_p(qkv3), qkv3.stride(1), _p(o), o.stride(1), _p(lse), _p(pos), _p(cos), _p(sin), _p(kv_len),
                               B, H, S, D, float(scale), _s())
    _lib.check(rc, f"mh_attn_rope_fwd B={B} H={H} S={S} D={D}")
    return o, lse


def attn_rope_bwd(qkv3, o, dout, lse, H: int, D: int, scale: float, pos, cos, sin, kv_len=None, dqkv=None):
    """Backward of attn_rope_fwd: dqkv [B, S, ld] = [dq | dk | dv] (dq, dk un-rotated).  dout [B*S or B,S, W] bf16."""
    _chk_qkv3(qkv3, H, D, "attn_rope_bwd")
    B, S = qkv3.shape[0], qkv3.shape[1]
    if dqkv is None:
        dqkv = torch.empty_like(qkv3)
    d2 = dout.reshape(B * S, -1)
    _chk2d(d2, BF16, "attn_rope_bwd.dout")
    rc = _L().mh_attn_rope_bwd(_p(qkv3), qkv3.stride(1), _p(o), o.stride(1), _p(d2), d2.stride(0), _p(lse), _p(dqkv),
                               _p(pos), _p(cos), _p(sin), _p(kv_len), B, H, S, D, float(scale), _s())
    _lib.check(rc, f"mh_attn_rope_bwd B={B} H={H} S={S} D={D}")
    return dqkv


def gemm_attn_rope_bwd(a, bw, qkv3, o, lse, H: int, D: int, scale: float, pos, cos, sin, kv_len=None, dqkv=None):
    """dO = a @ bw^T (the o_proj dgrad) and the fused rotary attention backward that consumes it; when the GEMM policy
    splits K the attention kernel sums the partial slabs itself.  Same bits as gemm() + attn_rope_bwd()."""
    _chk_qkv3(qkv3, H, D, "gemm_attn_rope_bwd")
    _chk2d(a, BF16, "gemm_attn_rope_bwd.a")
    _chk2d(bw, BF16, "gemm_attn_rope_bwd.bw")
    B, S = qkv3.shape[0], qkv3.shape[1]
    M, K = a.shape
    if M != B * S or bw.shape != (H * D, K):
        raise _lib.MyriadHipError(f"gemm_attn_rope_bwd: a {tuple(a.shape)} bw {tuple(bw.shape)} vs B*S={B * S} W={H * D}")
    if dqkv is None:
        dqkv = torch.empty_like(qkv3)
    do_buf = torch.empty((M, H * D), dtype=BF16, device=a.device)
    rc = _L().mh_gemm_attn_rope_bwd(_p(a), a.stride(0), _p(bw), bw.stride(0), _p(do_buf), K, _p(qkv3), qkv3.stride(1), _p(o),
                                    o.stride(1), _p(lse), _p(dqkv), _p(pos), _p(cos), _p(sin), _p(kv_len), B, H, S, D,
                                    float(scale), _s())
    _lib.check(rc, f"mh_gemm_attn_rope_bwd M={M} K={K} H={H} S={S}")
    return dqkv


# --------------------------------------------------------------------------- norms
def rmsnorm_fwd(x: torch.Tensor, w: torch.Tensor, eps: float, out=None):
    """out may be a [M, D] view of a wider bf16 buffer (row stride >= D)."""
    M, D = x.shape
    if out is None:
        out = torch.empty((M, D), dtype=BF16, device=x.device)
    _lib.check(_L().mh_rmsnorm_fwd(_p(x), _p(w), _p(out), out.stride(0), M, D, float(eps), _s()), "mh_rmsnorm_fwd")
    return out


def dropout_bf16(x2d: torch.Tensor, p: float, seed: int):
    """x2d: [rows, cols] bf16 (row stride free).  Returns a contiguous masked/rescaled copy."""
    rows, cols = x2d.shape
    y = torch.empty((rows, cols), dtype=BF16, device=x2d.device)
    _lib.check(_L().mh_dropout_bf16(_p(x2d), x2d.stride(0), _p(y), cols, rows, cols, float(p),
                                    int(seed) & 0xFFFFFFFFFFFFFFFF, _s()), "mh_dropout_bf16")
    return y


def dropout_add_(dy2d: torch.Tensor, acc2d: torch.Tensor, p: float, seed: int):
    rows, cols = dy2d.shape
    _lib.check(_L().mh_dropout_add_f32(_p(dy2d), dy2d.stride(0), _p(acc2d), acc2d.stride(0), rows, cols, float(p),
                                       int(seed) & 0xFFFFFFFFFFFFFFFF, _s()), "mh_dropout_add_f32")
    return acc2d


def rmsnorm_bwd(dy, x, w, eps: float, dres=None, want_f32=True, want_bf16=False):
    M, D = x.shape
    dx = torch.empty((M, D), dtype=F32, device=x.device) if want_f32 else None
    dxb = torch.empty((M, D), dtype=BF16, device=x.device) if want_bf16 else None
    _lib.check(_L().mh_rmsnorm_bwd(_p(dy), _p(x), _p(w), _p(dres), _p(dx), _p(dxb), M, D, float(eps), _s()),
               "mh_rmsnorm_bwd")
    return dx, dxb


def gemm_rmsnorm_bwd(a, b, x, w, eps: float, dres=None, want_f32=True, want_bf16=True):
    """dY = a @ b^T (f32), then rmsnorm_bwd(dY, x, w) + dres -> (dx f32, dx bf16): a dgrad Linear and the norm backward that
    reads it, with the split-K partials summed inside the norm kernel.  Same bits as gemm(out f32) + rmsnorm_bwd."""
    _chk2d(a, BF16, "gemm_rmsnorm_bwd.a")
    _chk2d(b, BF16, "gemm_rmsnorm_bwd.b")
    M, K = a.shape
    N = b.shape[0]
    if x.shape != (M, N) or b.shape[1] != K:
        raise _lib.MyriadHipError(f"gemm_rmsnorm_bwd: a {tuple(a.shape)} b {tuple(b.shape)} x {tuple(x.shape)}")
    dy = torch.empty((M, N), dtype=F32, device=a.device)
    dx = torch.empty((M, N), dtype=F32, device=a.device) if want_f32 else None
    dxb = torch.empty((M, N), dtype=BF16, device=a.device) if want_bf16 else None
    rc = _L().mh_gemm_rmsnorm_bwd(_p(a), a.stride(0), _p(b), b.stride(0), _p(dy), _p(x), _p(w), _p(dres), _p(dx), _p(dxb),
                                  M, N, K, float(eps), _s())
    _lib.check(rc, f"mh_gemm_rmsnorm_bwd M={M} N={N} K={K}")
    return dx, dxb


def layernorm_fwd(x, w, b, eps: float, want_bf16=True, want_f32=False):
    M, D = x.shape
    yb = torch.empty((M, D), dtype=BF16, device=x.device) if want_bf16 else None
    yf = torch.empty((M, D), dtype=F32, device=x.device) if want_f32 else None
    _lib.check(_L().mh_layernorm_fwd(_p(x), _p(w), _p(b), _p(yb), _p(yf), M, D, float(eps), _s()), "mh_layernorm_fwd")
    return yb, yf


def layernorm_bwd(dy, x, w, eps: float, dres=None, want_f32=True, want_bf16=False):
    M, D = x.shape
    dx = torch.empty((M, D), dtype=F32, device=x.device) if want_f32 else None
    dxb = torch.empty((M, D), dtype=BF16, device=x.device) if want_bf16 else None
    _lib.check(_L().mh_layernorm_bwd(_p(dy), _p(x), _p(w), _p(dres), _p(dx), _p(dxb), M, D, float(eps), _s()),
               "mh_layernorm_bwd")
    return dx, dxb


# --------------------------------------------------------------------------- elementwise
def rope_(x2d: torch.Tensor, col0: int, n_heads: int, head_dim: int, pos: torch.Tensor, cos, sin, sign: float = 1.0):
    n_tok = x2d.shape[0]
    _lib.check(_L().mh_rope_inplace(_p(x2d), x2d.stride(0), col0, n_tok, n_heads, head_dim, _p(pos), _p(cos), _p(sin),
                                    float(sign), _s()), "mh_rope_inplace")
    return x2d


def silu_mul_fwd(gu: torch.Tensor):
    M, I2 = gu.shape
    h = torch.empty((M, I2 // 2), dtype=BF16, device=gu.device)
    _lib.check(_L().mh_silu_mul_fwd(_p(gu), _p(h), M, I2 // 2, _s()), "mh_silu_mul_fwd")
    return h


def silu_mul_bwd(dh: torch.Tensor, gu: torch.Tensor):
    M, I2 = gu.shape
    dgu = torch.empty_like(gu)
    _lib.check(_L().mh_silu_mul_bwd(_p(dh), _p(gu), _p(dgu), M, I2 // 2, _s()), "mh_silu_mul_bwd")
    return dgu


SWIGLU_BLK = 128      # gate / up interleave of the fused MLP GEMMs (csrc/gemm.hip mh_gemm_swiglu_*)


def interleave_gate_up(wg: torch.Tensor, wu: torch.Tensor, blk: int = SWIGLU_BLK) -> torch.Tensor:
    """[I, D] gate and up weights -> [2I, D] with rows [g 0..blk-1 | u 0..blk-1 | g blk.. | ...] (I % blk == 0)."""
    I, D = wg.shape
    if I % blk:
        raise _lib.MyriadHipError(f"interleave_gate_up: I={I} is not a multiple of {blk}")
    return torch.stack([wg.reshape(I // blk, blk, D), wu.reshape(I // blk, blk, D)], 1).reshape(2 * I, D).contiguous()


def silu_mul_fwd_blk(gu: torch.Tensor, blk: int = SWIGLU_BLK):
    M, I2 = gu.shape
    h = torch.empty((M, I2 // 2), dtype=BF16, device=gu.device)
    _lib.check(_L().mh_silu_mul_fwd_blk(_p(gu), _p(h), M, I2 // 2, blk, _s()), "mh_silu_mul_fwd_blk")
    return h


def silu_mul_bwd_blk(dh: torch.Tensor, gu: torch.Tensor, blk: int = SWIGLU_BLK):
    M, I2 = gu.shape
    dgu = torch.empty_like(gu)
    _lib.check(_L().mh_silu_mul_bwd_blk(_p(dh), _p(gu), _p(dgu), M, I2 // 2, blk, _s()), "mh_silu_mul_bwd_blk")
    return dgu


def gemm_swiglu_fwd(x: torch.Tensor, wgu: torch.Tensor):
    """(gu [M, 2I] bf16 in the interleaved layout, act [M, I] = silu(g) * u): the gate|up projection with the gated product
    in its epilogue (separate launches, same bits, when the policy does not run the fused kernel)."""
    _chk2d(x, BF16, "gemm_swiglu_fwd.x")
    _chk2d(wgu, BF16, "gemm_swiglu_fwd.wgu")
    M, K = x.shape
    I = wgu.shape[0] // 2
    gu = torch.empty((M, 2 * I), dtype=BF16, device=x.device)
    act = torch.empty((M, I), dtype=BF16, device=x.device)
    rc = _L().mh_gemm_swiglu_fwd(_p(x), x.stride(0), _p(wgu), wgu.stride(0), _p(gu), 2 * I, _p(act), I, M, I, K, _s())
    _lib.check(rc, f"mh_gemm_swiglu_fwd M={M} I={I} K={K}")
    return gu, act


def gemm_swiglu_bwd(dh: torch.Tensor, wdT: torch.Tensor, gu: torch.Tensor):
    """dgu [M, 2I] = silu_mul_bwd(dh @ wdT^T, gu): the down projection's dgrad with the gate backward in its epilogue."""
    _chk2d(dh, BF16, "gemm_swiglu_bwd.dh")
    _chk2d(wdT, BF16, "gemm_swiglu_bwd.wdT")
    _chk2d(gu, BF16, "gemm_swiglu_bwd.gu")
    M, K = dh.shape
    I = wdT.shape[0]
    dgu = torch.empty_like(gu)
    dact = torch.empty((M, I), dtype=BF16, device=dh.device)
    rc = _L().mh_gemm_swiglu_bwd(_p(dh), dh.stride(0), _p(wdT), wdT.stride(0), _p(gu), gu.stride(0), _p(dgu), dgu.stride(0),
                                 _p(dact), M, I, K, _s())
    _lib.check(rc, f"mh_gemm_swiglu_bwd M={M} I={I} K={K}")
    return dgu


def gemm_gelu_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None):
    """(pre [M, N] bf16 = x @ w^T + bias, act = gelu(pre)): the MLP's first product with the GELU in its epilogue (separate
    launches, same bits, when the policy does not run an unsplit tile kernel)."""
    _chk2d(x, BF16, "gemm_gelu_fwd.x")
    _chk2d(w, BF16, "gemm_gelu_fwd.w")
    M, K = x.shape
    N = w.shape[0]
    pre = torch.empty((M, N), dtype=BF16, device=x.device)
    act = torch.empty((M, N), dtype=BF16, device=x.device)
    rc = _L().mh_gemm_gelu_fwd(_p(x), x.stride(0), _p(w), w.stride(0), _p(bias), _p(pre), N, _p(act), N, M, N, K, _s())
    _lib.check(rc, f"mh_gemm_gelu_fwd M={M} N={N} K={K}")
    return pre, act


def gemm_gelu_bwd(dy: torch.Tensor, wT: torch.Tensor, pre: torch.Tensor):
    """dpre [M, N] bf16 = bf16(dy @ wT^T) * gelu'(pre): the MLP's second dgrad with the GELU backward in its epilogue."""
    _chk2d(dy, BF16, "gemm_gelu_bwd.dy")
    _chk2d(wT, BF16, "gemm_gelu_bwd.wT")
    _chk2d(pre, BF16, "gemm_gelu_bwd.pre")
    M, K = dy.shape
    N = wT.shape[0]
    dpre = torch.empty((M, N), dtype=BF16, device=dy.device)
    dact = torch.empty((M, N), dtype=BF16, device=dy.device)
    rc = _L().mh_gemm_gelu_bwd(_p(dy), dy.stride(0), _p(wT), wT.stride(0), _p(pre), pre.stride(0), _p(dpre), N, _p(dact), M, N, K, _s())
    _lib.check(rc, f"mh_gemm_gelu_bwd M={M} N={N} K={K}")
    return dpre


def gelu_fwd(x):
    y = torch.empty_like(x)
    _lib.check(_L().mh_gelu_fwd(_p(x), _p(y), x.numel(), _s()), "mh_gelu_fwd")
    return y


def gelu_bwd(dy, x):
    dx = torch.empty_like(x)
    _lib.check(_L().mh_gelu_bwd(_p(dy), _p(x), _p(dx), x.numel(), _s()), "mh_gelu_bwd")
    return dx


def to_bf16(x: torch.Tensor, out=None):
    if out is None:
        out = torch.empty(x.shape, dtype=BF16, device=x.device)
    _lib.check(_L().mh_cast_f32_to_bf16(_p(x), _p(out), x.numel(), _s()), "mh_cast_f32_to_bf16")
    return out


def to_f32(x: torch.Tensor, out=None):
    if out is None:
        out = torch.empty(x.shape, dtype=F32, device=x.device)
    _lib.check(_L().mh_cast_bf16_to_f32(_p(x), _p(out), x.numel(), _s()), "mh_cast_bf16_to_f32")
    return out


def transpose_to_bf16(x2d: torch.Tensor, pad_to: int = 64, out=None):
    """[R,C] (f32 or bf16) -> [C, round_up(R,pad_to)] bf16, zero padded."""
    R, C = x2d.shape
    ldo = round_up(R, pad_to)
    if out is None:
        out = torch.empty((C, ldo), dtype=BF16, device=x2d.device)
    _lib.check(_L().mh_transpose_to_bf16(_p(x2d), int(x2d.dtype == F32), x2d.stride(0), _p(out), out.stride(0), R, C,
                                         _s()), "mh_transpose_to_bf16")
    return out


def copy2d(src: torch.Tensor, dst: torch.Tensor, accumulate: bool = False):
    rows, cols = src.shape
    _lib.check(_L().mh_copy2d_f32(_p(src), src.stride(0), _p(dst), dst.stride(0), rows, cols, int(accumulate), _s()),
               "mh_copy2d_f32")
    return dst


def copy3d(src: torch.Tensor, dst: torch.Tensor, accumulate: bool = False):
    nb, rows, cols = src.shape
    _lib.check(_L().mh_copy3d_f32(_p(src), src.stride(0), src.stride(1), _p(dst), dst.stride(0), dst.stride(1), nb,
                                  rows, cols, int(accumulate), _s()), "mh_copy3d_f32")
    return dst


def embed_gather(table: torch.Tensor, ids: torch.Tensor, out: torch.Tensor, dst_rows: Optional[torch.Tensor] = None):
    """out[dst_rows[i]] = f32(table[ids[i]]); out is a 2-D f32 view."""
    n = ids.numel()
    D = table.shape[1]
    _lib.check(_L().mh_embed_gather(_p(table), _p(ids), _p(dst_rows), _p(out), n, D, out.stride(0), _s()),
               "mh_embed_gather")
    return out


def gather_rows_bf16(src: torch.Tensor, rows: torch.Tensor):
    n, D = rows.numel(), src.shape[1]
    out = torch.empty((n, D), dtype=BF16, device=src.device)
    _lib.check(_L().mh_gather_rows_f32_to_bf16(_p(src), src.stride(0), _p(rows), _p(out), n, D, _s()),
               "mh_gather_rows_f32_to_bf16")
    return out


def gather_rows(src: torch.Tensor, rows: torch.Tensor):
    """src [*, D] (bf16 or f32, row stride free) -> [n, D] dense, same dtype: rows `rows` (int32, device)."""
    n, D = rows.numel(), src.shape[1]
    out = torch.empty((n, D), dtype=src.dtype, device=src.device)
    _lib.check(_L().mh_gather_rows(_p(src), src.stride(0), _p(rows), _p(out), n, D, src.element_size(), _s()), "mh_gather_rows")
    return out


def expand_rows(src: torch.Tensor, inv: torch.Tensor, M: int):
    """[n, D] dense -> [M, D] with row m = src[inv[m]] or zeros where inv[m] < 0 (inv int32 [M], device)."""
    D = src.shape[1]
    out = torch.empty((M, D), dtype=src.dtype, device=src.device)
    _lib.check(_L().mh_expand_rows(_p(src), _p(inv), _p(out), D, M, D, src.element_size(), _s()), "mh_expand_rows")
    return out


def gather_rows_f32(src: torch.Tensor, rows: torch.Tensor):
    n, D = rows.numel(), src.shape[1]
    out = torch.empty((n, D), dtype=F32, device=src.device)
    _lib.check(_L().mh_gather_rows_f32(_p(src), src.stride(0), _p(rows), _p(out), n, D, _s()), "mh_gather_rows_f32")
    return out


def copy3d_bf16(src: torch.Tensor, dst: torch.Tensor):
    nb, rows, cols = src.shape
    _lib.check(_L().mh_copy3d_bf16(_p(src), src.stride(0), src.stride(1), _p(dst), dst.stride(0), dst.stride(1), nb,
                                   rows, cols, _s()), "mh_copy3d_bf16")
    return dst


def patchify(img: torch.Tensor, P: int):
    B, C, H, W = img.shape
    Kpad = round_up(C * P * P, 64)
    out = torch.empty((B * (H // P) * (W // P), Kpad), dtype=BF16, device=img.device)
    _lib.check(_L().mh_patchify_nchw(_p(img), _p(out), B, C, H, W, P, Kpad, _s()), "mh_patchify_nchw")
    return out


def scatter_rows(src: torch.Tensor, rows: torch.Tensor, dst: torch.Tensor, accumulate: bool = False):
    n, D = src.shape
    _lib.check(_L().mh_scatter_rows_f32(_p(src), _p(rows), _p(dst), dst.stride(0), n, D, int(accumulate), _s()),
               "mh_scatter_rows_f32")
    return dst


def colsum(x2d: torch.Tensor):
    R, C = x2d.shape
    out = torch.empty((C,), dtype=F32, device=x2d.device)
    _lib.check(_L().mh_colsum_f32(_p(x2d), x2d.stride(0), _p(out), R, C, _s()), "mh_colsum_f32")
    return out


def scale_(x: torch.Tensor, a: float):
    _lib.check(_L().mh_scale_f32(_p(x), float(a), x.numel(), _s()), "mh_scale_f32")
    return x


# --------------------------------------------------------------------------- low-rank adaptor
def lowrank_fwd(x, A, Bm):
    M, D = x.shape
    R = A.shape[0]
    y = torch.empty_like(x)
    t = torch.empty((M, R), dtype=F32, device=x.device)
    _lib.check(_L().mh_lowrank_fwd(_p(x), _p(A), _p(Bm), _p(y), _p(t), M, D, R, _s()), "mh_lowrank_fwd")
    return y, t


def lowrank_bwd(dy, x, t, A, Bm, dA, dB, need_dx=False):
    M, D = x.shape
    R = A.shape[0]
    ws = torch.empty((_L().mh_lowrank_bwd_ws_floats(M, D, R),), dtype=F32, device=x.device)
    dx = torch.empty_like(x) if need_dx else None
    _lib.check(_L().mh_lowrank_bwd(_p(dy), _p(x), _p(t), _p(A), _p(Bm), _p(dA), _p(dB), _p(dx), _p(ws), M, D, R, _s()),
               "mh_lowrank_bwd")
    return dx


# --------------------------------------------------------------------------- loss / decode
def clamp_ce(logits: torch.Tensor, labels: torch.Tensor, grad_scale: float, want_grad: bool = True, ldd: int = 0):
    R, V = logits.shape
    row_loss = torch.empty((R,), dtype=F32, device=logits.device)
    dlog = None
    if want_grad:
        ldd = ldd or round_up(V, 64)
        dlog = torch.empty((R, ldd), dtype=BF16, device=logits.device)
    _lib.check(_L().mh_clamp_ce(_p(logits), logits.stride(0), _p(labels), _p(row_loss), _p(dlog), ldd, R, V,
                                float(grad_scale), _s()), "mh_clamp_ce")
    return row_loss, dlog


def sum_f32(x: torch.Tensor, scale: float = 1.0):
    out = torch.empty((1,), dtype=F32, device=x.device)
    _lib.check(_L().mh_sum_f32(_p(x), _p(out), x.numel(), float(scale), _s()), "mh_sum_f32")
    return out


def kv_append(src2d: torch.Tensor, cache: torch.Tensor, pos_dev: torch.Tensor):
    """cache[b, pos_dev[0], :] = src2d[b, :] with the position read on the device (graph-replayable)."""
    B, cols = src2d.shape
    _lib.check(_L().mh_kv_append_bf16(_p(src2d), src2d.stride(0), _p(cache), cache.stride(0), cache.stride(1),
                                      _p(pos_dev), B, cols, _s()), "mh_kv_append_bf16")


def rope_kv_append(qkv2d: torch.Tensor, n_heads: int, head_dim: int, pos: torch.Tensor, cos_tab: torch.Tensor,
                   sin_tab: torch.Tensor, cache: torch.Tensor, pos_dev: torch.Tensor):
    """Decode token: rotary on q (in place) and k, k|v written to cache[b, pos_dev[0]] -- one launch."""
    _chk2d(qkv2d, BF16, "rope_kv_append.qkv")
    B = qkv2d.shape[0]
    if qkv2d.shape[1] < 3 * n_heads * head_dim or cache.shape[2] != 2 * n_heads * head_dim:
        raise _lib.MyriadHipError("rope_kv_append: qkv must be [B, >=3W] and cache [B, T, 2W]")
    _lib.check(_L().mh_rope_kv_append(_p(qkv2d), qkv2d.stride(0), n_heads, head_dim, _p(pos), _p(cos_tab), _p(sin_tab),
                                      _p(cache), cache.stride(0), cache.stride(1), _p(pos_dev), B, _s()),
               "mh_rope_kv_append")


def add_i32_(x: torch.Tensor, delta: int):
    _lib.check(_L().mh_add_i32(_p(x), x.numel(), int(delta), _s()), "mh_add_i32")
    return x


def argmax_rows(logits: torch.Tensor, ban_id: int = -1, want_margin: bool = False, out=None, margin_out=None):
    R, V = logits.shape
    if out is None:
        out = torch.empty((R,), dtype=torch.long, device=logits.device)
    margin = margin_out
    if margin is None and want_margin:
        margin = torch.empty((R,), dtype=F32, device=logits.device)
    _lib.check(_L().mh_argmax_rows(_p(logits), logits.stride(0), _p(out), _p(margin), R, V, ban_id, _s()),
               "mh_argmax_rows")
    return (out, margin) if want_margin else out


def argmax_pmax_rows(logits: torch.Tensor, out, margin_out, pmax_out, ban_id: int = -1, inv_temp: float = 1.0):
    """Row arg-max + top-1/top-2 margin + p_max = softmax(logits * inv_temp)[argmax] into the given buffers."""
    R, V = logits.shape
    _lib.check(_L().mh_argmax_pmax_rows(_p(logits), logits.stride(0), _p(out), _p(margin_out), _p(pmax_out), R, V, ban_id,
                                        float(inv_temp), _s()), "mh_argmax_pmax_rows")
    return out, margin_out, pmax_out


def decode_record(nxt, margin, pmax, rec, next_ids, step_dev):
    """rec[3, R] = this step's (ids, margins, p_max) as f32; next_ids = ids; step += 1 -- all on the device (graph-replayable)."""
    R = nxt.numel()
    _lib.check(_L().mh_decode_record(_p(nxt), _p(margin), _p(pmax), _p(rec), _p(next_ids), _p(step_dev), R, _s()),
               "mh_decode_record")


def decode_advance(nxt, margin, pmax, rec, next_ids, step_dev, pos, kvlen):
    """decode_record() and pos += 1, kvlen += 1 in one launch (the end of a KV-cache token step)."""
    R = nxt.numel()
    _lib.check(_L().mh_decode_advance(_p(nxt), _p(margin), _p(pmax), _p(rec), _p(next_ids), _p(step_dev), _p(pos), _p(kvlen), R,
                                      _s()), "mh_decode_advance")


# --------------------------------------------------------------------------- conv stack pieces
def im2col(x_nhwc: torch.Tensor, kh: int, kw: int, pad: int, bias_col: bool = True):
    """[B*OH*OW, Kpad] bf16 patches, (ky, kx, c) order; with bias_col a column of ones follows the K patch columns (the bias
    then rides the GEMM as one more weight column)."""
    B, H, W, C = x_nhwc.shape
    OH, OW = H + 2 * pad - kh + 1, W + 2 * pad - kw + 1
    Kpad = round_up(kh * kw * C + (1 if bias_col else 0), 64)
    col = torch.empty((B * OH * OW, Kpad), dtype=BF16, device=x_nhwc.device)
    _lib.check(_L().mh_im2col_nhwc(_p(x_nhwc), _p(col), B, H, W, C, kh, kw, pad, Kpad, _s()), "mh_im2col_nhwc")
    return col


def col2im(dcol: torch.Tensor, B, H, W, C, kh, kw, pad):
    dx = torch.empty((B, H, W, C), dtype=F32, device=dcol.device)
    _lib.check(_L().mh_col2im_nhwc(_p(dcol), _p(dx), B, H, W, C, kh, kw, pad, dcol.stride(0), _s()), "mh_col2im_nhwc")
    return dx


def relu_pool_fwd(y2d: torch.Tensor, B, H, W, C):
    p = torch.empty((B, H // 2, W // 2, C), dtype=BF16, device=y2d.device)
    _lib.check(_L().mh_relu_maxpool2_fwd(_p(y2d), int(y2d.dtype == F32), y2d.stride(0), _p(p), B, H, W, C, _s()),
               "mh_relu_maxpool2_fwd")
    return p


def relu_pool_bwd(dp: torch.Tensor, y2d: torch.Tensor, B, H, W, C, pad_cols_to: int = 1):
    """dy [B*H*W, C] bf16 (a view of a zero-padded [*, round_up(C,pad_cols_to)] buffer so it can be a GEMM A operand)."""
    cpad = round_up(C, pad_cols_to)
    full = (torch.zeros if cpad != C else torch.empty)((B * H * W, cpad), dtype=BF16, device=y2d.device)
    _lib.check(_L().mh_relu_maxpool2_bwd(_p(dp), _p(y2d), int(y2d.dtype == F32), y2d.stride(0), _p(full),
                                         full.stride(0), B, H, W, C, _s()), "mh_relu_maxpool2_bwd")
    return full[:, :C], full


def conv_pack(Wm: torch.Tensor, bias: Optional[torch.Tensor], out=None):
    Cout, K = Wm.shape
    Kpad = round_up(K + 1, 64)
    if out is None:
        out = torch.empty((Cout, Kpad), dtype=BF16, device=Wm.device)
    _lib.check(_L().mh_conv_pack_weight(_p(Wm), _p(bias), _p(out), Cout, K, Kpad, _s()), "mh_conv_pack_weight")
    return out


def conv_unpack_grad(dWp: torch.Tensor, dW: torch.Tensor, db: Optional[torch.Tensor]):
    Cout, Kpad = dWp.shape
    K = dW.shape[1]
    _lib.check(_L().mh_conv_unpack_grad(_p(dWp), _p(dW), _p(db), Cout, K, Kpad, _s()), "mh_conv_unpack_grad")


# --------------------------------------------------------------------------- optimiser
def adamw_step(p, g, m, v, lr, wd, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, shadow=None):
    _lib.check(_L().mh_adamw_step(_p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), float(lr), float(beta1),
                                  float(beta2), float(eps), float(wd), int(step), float(grad_scale), _s()),
               "mh_adamw_step")


def adamw_gated(p, g, m, v, lr, wd, used: torch.Tensor, steps: torch.Tensor, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """AdamW on one contiguous range iff the device scalar `used` > 0, with step = device scalar `steps` + 1."""
    _lib.check(_L().mh_adamw_gated(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
                                   float(wd), float(grad_scale), _p(used), _p(steps), _s()), "mh_adamw_gated")


def adamw_bump(used: torch.Tensor, steps: torch.Tensor):
    _lib.check(_L().mh_adamw_bump(_p(used), _p(steps), steps.numel(), _s()), "mh_adamw_bump")


# --------------------------------------------------------------------------- vision-expert map heads (K16)
def l2norm_rows(x: torch.Tensor, want_bf16: bool = True, want_f32: bool = False, eps: float = 1e-12):
    """y = x / max(||x||, eps) row-wise; x [M, D] f32 (row stride free)."""
    _chk2d(x, F32, "l2norm_rows.x")
    M, D = x.shape
    yb = torch.empty((M, D), dtype=BF16, device=x.device) if want_bf16 else None
    yf = torch.empty((M, D), dtype=F32, device=x.device) if want_f32 else None
    _lib.check(_L().mh_l2norm_rows(_p(x), x.stride(0), _p(yb), _p(yf), D, M, D, float(eps), _s()), "mh_l2norm_rows")
    return yb, yf


def pair_logits(p: torch.Tensor, text: torch.Tensor, rows_per_batch: int, scale: float = 100.0):
    """p [rows, C] f32, text [B, 2, C] f32 -> [rows, 2] = scale * cos(p, text[row // rows_per_batch])."""
    _chk2d(p, F32, "pair_logits.p")
    rows, C = p.shape
    out = torch.empty((rows, 2), dtype=F32, device=p.device)
    _lib.check(_L().mh_pair_logits(_p(p), p.stride(0), _p(text.contiguous()), _p(out), rows, rows_per_batch, C, float(scale),
                                   _s()), "mh_pair_logits")
    return out


def zs_accumulate(logits: torch.Tensor, mask_acc: torch.Tensor, map_acc: torch.Tensor, w: float):
    B, h, S = mask_acc.shape[0], mask_acc.shape[-1], map_acc.shape[-1]
    _lib.check(_L().mh_zs_accumulate(_p(logits), _p(mask_acc), _p(map_acc), B, h, S, float(w), _s()), "mh_zs_accumulate")


def rowmax_skip(scores: torch.Tensor, acc: torch.Tensor, period: int, w: float):
    """acc[row] += w * max over columns c with c % period != 0 (period 0: all columns)."""
    rows, cols = scores.shape
    _lib.check(_L().mh_rowmax_skip(_p(scores), scores.stride(0), _p(acc), rows, cols, period, float(w), _s()), "mh_rowmax_skip")


def bilinear_ac(x: torch.Tensor, H: int, W: int, one_minus: bool = False):
    """x [B, h, w] f32 -> [B, H, W], align_corners=True; optionally 1 - result."""
    B, h, w = x.shape
    out = torch.empty((B, H, W), dtype=F32, device=x.device)
    _lib.check(_L().mh_bilinear_ac(_p(x.contiguous()), _p(out), B, h, w, H, W, int(one_minus), _s()), "mh_bilinear_ac")
    return out


def gemm_residual_rmsnorm(a: torch.Tensor, b: torch.Tensor, residual: torch.Tensor, norm_w: torch.Tensor, eps: float,
                          y_out: Optional[torch.Tensor] = None):
    """(h, y): h = a @ b^T + residual (f32), y = rmsnorm(h) * norm_w (bf16; y_out may be a [M, N] view of a wider buffer).
    One launch less than gemm + rmsnorm_fwd when the GEMM is split along K; bit-identical results either way."""
    _chk2d(a, BF16, "gemm_residual_rmsnorm.a")
    _chk2d(b, BF16, "gemm_residual_rmsnorm.b")
    _chk2d(residual, F32, "gemm_residual_rmsnorm.residual")
    M, K = a.shape
    N = b.shape[0]
    h = torch.empty((M, N), dtype=F32, device=a.device)
    y = y_out if y_out is not None else torch.empty((M, N), dtype=BF16, device=a.device)
    _chk2d(y, BF16, "gemm_residual_rmsnorm.y")
    rc = _L().mh_gemm_residual_rmsnorm(_p(a), a.stride(0), _p(b), b.stride(0), _p(h), N, _p(residual), residual.stride(0),
                                       _p(norm_w), float(eps), _p(y), y.stride(0), M, N, K, _s())
    _lib.check(rc, f"mh_gemm_residual_rmsnorm M={M} N={N} K={K}")
    return h, y


def gemm_residual_layernorm(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor], residual: torch.Tensor,
                            norm_w: torch.Tensor, norm_b: torch.Tensor, eps: float):
    """(h, y): h = a @ b^T + bias + residual (f32), y = layernorm(h) * norm_w + norm_b (bf16).  Pre-LN ViT blocks."""
    _chk2d(a, BF16, "gemm_residual_layernorm.a")
    _chk2d(b, BF16, "gemm_residual_layernorm.b")
    _chk2d(residual, F32, "gemm_residual_layernorm.residual")
    M, K = a.shape
    N = b.shape[0]
    h = torch.empty((M, N), dtype=F32, device=a.device)
    y = torch.empty((M, N), dtype=BF16, device=a.device)
    rc = _L().mh_gemm_residual_layernorm(_p(a), a.stride(0), _p(b), b.stride(0), _p(h), N, _p(bias), _p(residual),
                                         residual.stride(0), _p(norm_w), _p(norm_b), float(eps), _p(y), M, N, K, _s())
    _lib.check(rc, f"mh_gemm_residual_layernorm M={M} N={N} K={K}")
    return h, y

"""GPU parity tests: every HIP kernel (through the C ABI) against an fp32 torch / oracle reference of the same
op on the same seeded inputs.  Tolerances are stated per test: bf16 operands carry 2^-9 relative rounding, all
accumulation is fp32."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from myriad_amd import ops  # noqa: E402
from oracle import myriad_ref as R  # noqa: E402

DEV = "cuda"


def _opt_hook(name: str):
    """Setter of a library option (include/myriad_hip.h: mh_set_option): the A/B switch between two forms of one op."""
    from myriad_amd import _lib as L
    lib = L.load()
    return lambda v: lib.mh_set_option(name.encode(), int(v))


def bf(x):
    return x.to(torch.bfloat16)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("regstage", [False, True])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 128), (1, 256, 192), (1184, 4096, 4096), (648, 768, 3072),
                                   (72, 1024, 2624), (257, 1408, 640)])
def test_gemm_plain(M, N, K, regstage):
    a = bf(rnd(M, K, seed=1)).to(DEV)
    # asymmetric B so a transposed C-write cannot pass
    b = bf(rnd(N, K, seed=2) + torch.arange(N)[:, None] * 1e-3).to(DEV)
    ref = a.float() @ b.float().T
    out = ops.gemm(a, b, out_dtype=torch.float32, regstage=regstage)
    assert relerr(out, ref) < (2e-5 * math.sqrt(K) + 1e-6 if regstage else slab_tol(M, N, K, 2e-5 * math.sqrt(K) + 1e-6))
    out_b = ops.gemm(a, b, regstage=regstage)
    assert relerr(out_b.float(), ref) < 6e-3


def test_gemm_epilogues_and_strides():
    M, N, K = 300, 520, 256
    a_full = bf(rnd(M, K + 64, seed=3)).to(DEV)
    a = a_full[:, :K]  # lda > K
    b = bf(rnd(N, K, seed=4)).to(DEV)
    bias = rnd(N, seed=5).to(DEV)
    res = rnd(M, N, seed=6).to(DEV)
    ref = a.float() @ b.float().T
    o1 = ops.gemm(a, b, bias=bias, out_dtype=torch.float32)
    assert relerr(o1, ref + bias) < 1e-4
    o2 = ops.gemm(a, b, bias=bias, gelu=True, out_dtype=torch.float32)
    assert relerr(o2, F.gelu(ref + bias)) < 1e-4
    o3 = ops.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32, alpha=0.5)
    assert relerr(o3, 0.5 * ref + bias + res) < 1e-4
    # in-place accumulate: out aliases residual
    acc = res.clone()
    ops.gemm(a, b, out=acc, residual=acc)
    assert relerr(acc, ref + res) < 1e-4
    # strided output view
    wide = torch.zeros(M, N + 72, dtype=torch.bfloat16, device=DEV)
    ops.gemm(a, b, out=wide[:, 8:8 + N])
    assert relerr(wide[:, 8:8 + N].float(), ref) < 6e-3
    assert wide[:, :8].abs().max() == 0 and wide[:, 8 + N:].abs().max() == 0


@pytest.mark.parametrize("M,N,K", [(4, 64, 100352), (16, 64, 25088), (64, 192, 6272), (256, 576, 1600), (768, 1024, 128)])
def test_gemm_splitk_wgrad_shapes(M, N, K):
    a = bf(rnd(M, K, seed=7)).to(DEV)
    b = bf(rnd(N, K, seed=8) + torch.arange(N)[:, None] * 1e-3).to(DEV)
    ref = a.double() @ b.double().T
    out = torch.empty(M, N, device=DEV)
    ops.gemm_auto_f32(a, b, out)
    assert relerr(out, ref) < 1e-4
    # strided output (a view into a wider gradient buffer)
    wide = torch.zeros(M, N + 8, device=DEV)
    ops.gemm_auto_f32(a, b, wide[:, 4:4 + N])
    assert relerr(wide[:, 4:4 + N], ref) < 1e-4 and wide[:, :4].abs().max() == 0


def slab_tol(M, N, K, tight):
    """Accuracy bound of an automatically split product against fp32 math: the partial slabs of the 256x256 kernel and of the
    160-row tiles are bf16 (one 2^-9 rounding per partial sum, csrc/gemm.hip run_splitk) -- the bit-identity of fused and unfused
    forms is unaffected."""
    kernel, splits = ops.gemm_plan(M, N, K)
    return 4e-3 if (kernel in (2, 4, 5) and splits > 1) else tight


def _slab_hook():
    import ctypes
    from myriad_amd import _lib as L
    return _opt_hook("slab_bf16")


@pytest.mark.parametrize("slab_bf16", [1, 0])
@pytest.mark.parametrize("M,N,K", [(1184, 4096, 11008), (2056, 1408, 6144), (648, 768, 3072)])
def test_gemm_auto_splitk_epilogue(M, N, K, slab_bf16):
    """With a registered workspace mh_gemm_bf16_nt splits K for under-filled shapes; the epilogue moves into the
    reduce pass and must give the same results (bias + GELU + residual, bf16 and f32 out, in-place accumulate).
    The 256x256 kernel's partial slabs are bf16 by default (one 2^-9 rounding per partial sum: the product is then as exact as
    a bf16 GEMM output, not as an fp32 one); MYRIAD_SLAB_BF16=0 / the debug hook keep them fp32."""
    ops.ensure_workspace(DEV)
    hook = _slab_hook()
    hook(slab_bf16)
    try:
        lossy = bool(slab_bf16) and ops.gemm_plan(M, N, K)[0] == 2
        a = bf(rnd(M, K, seed=9)).to(DEV)
        b = bf(rnd(N, K, seed=10) * 0.05).to(DEV)
        bias = rnd(N, seed=11).to(DEV)
        res = rnd(M, N, seed=12).to(DEV)
        ref = a.float() @ b.float().T
        o = ops.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32)
        assert relerr(o, ref + bias + res) < (4e-3 if lossy else 1e-4)
        o_plain = ops.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32, variant=1)   # forced single-pass kernel
        assert relerr(o, o_plain) < (4e-3 if lossy else 1e-5)
        ob = ops.gemm(a, b, bias=bias, gelu=True)
        assert relerr(ob.float(), F.gelu(ref + bias)) < 6e-3
        acc = res.clone()
        ops.gemm(a, b, out=acc, residual=acc)
        assert relerr(acc, ref + res) < (4e-3 if lossy else 1e-4)
    finally:
        hook(1)


@pytest.mark.parametrize("M,N,K", [(648, 768, 768), (81, 768, 2304), (81, 3072, 768), (257, 1408, 1408), (196, 2368, 1024), (37, 100, 256)])
def test_small_grid_instances_of_the_gemm_are_bit_identical(M, N, K):
    """The policy's choice for one-round short-K grids (plan kernel 6: 64 x 64 tiles, 8-deep ring) and the other unsplit instances of
    gemm_nt_kernel (128 x 128, 128 x 64, 6-deep 128 x 64) accumulate every output in the same k order: same bits, with every
    epilogue form."""
    assert ops.gemm_plan(M, N, K) == (6, 1)
    a = bf(rnd(M, K, seed=21)).to(DEV)
    b = bf(rnd(N, K, seed=22) * 0.05).to(DEV)
    bias = rnd(N, seed=23).to(DEV)
    res = rnd(M, N, seed=24).to(DEV)
    auto_b, auto_f = ops.gemm(a, b, bias=bias, gelu=True), ops.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32)
    assert relerr(auto_f, a.float() @ b.float().T + bias + res) < 1e-4
    for v in (1, 3, 11, 13):
        assert torch.equal(ops.gemm(a, b, bias=bias, gelu=True, variant=v), auto_b), v
        assert torch.equal(ops.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32, variant=v), auto_f), v


@pytest.mark.parametrize("M,N,K,f32,hb,hr,alpha", [(256, 256, 64, 0, 0, 0, 1.0), (256, 256, 128, 1, 0, 0, 1.0), (300, 1000, 192, 0, 1, 0, 1.0),
                                                  (300, 1000, 320, 0, 1, 0, 0.5), (1184, 4160, 2048, 1, 1, 1, 1.0), (1184, 4096, 12288, 0, 0, 0, 1.0),
                                                  (520, 2176, 4096, 1, 0, 1, 0.25), (2056, 1408, 2112, 0, 1, 0, 1.0)])
def test_four_wave_and_eight_wave_instances_of_the_256_tile_are_bit_identical(M, N, K, f32, hb, hr, alpha):
    """gemm_x4.hip (four waves, 64-deep k-tiles, hand-scheduled loop: the policy's choice from 32 k-tiles per workgroup) and
    gemm_256.hip (eight waves) accumulate every output in the same k order: same bits with every epilogue form, odd and even
    k-tile counts (1, 2, 3, 5, ... k-tiles: the loop's one-tile, two-tile and steady paths), ragged M and N edges, K splits."""
    import ctypes
    from myriad_amd import _lib as L
    hook = _opt_hook("gemm256_impl")
    ops.ensure_workspace(torch.device(DEV))
    a = bf(rnd(M, K, seed=31)).to(DEV)
    b = bf(rnd(N, K, seed=32) * 0.05 + torch.arange(N)[:, None] * 1e-4).to(DEV)
    bias = rnd(N, seed=33).to(DEV) if hb else None
    res = rnd(M, N, seed=34).to(DEV) if hr else None
    dt = torch.float32 if f32 else torch.bfloat16
    ref = alpha * (a.float() @ b.float().T)
    if hb:
        ref = ref + bias
    if hr:
        ref = ref + res
    outs = []
    try:
        for impl in (0, 1):
            hook(impl)
            out = torch.full((M, N), 7.0, dtype=dt, device=DEV)
            ops.gemm(a, b, out=out, bias=bias, residual=res, alpha=alpha, variant=12)
            outs.append(out)
    finally:
        hook(-1)
    assert relerr(outs[1].float(), ref) < 6e-3        # bf16 output or bf16 split-K slabs; the claim here is the next line
    assert torch.equal(outs[0], outs[1])


def test_bf16_rounding_in_hardware_equals_the_integer_form():
    """common.h rounds fp32 -> bf16 with gfx950's v_cvt_pk_bf16_f32 (scalar and packed forms) instead of the integer
    add-0x7fff-plus-lsb sequence: the same bits for every high half (all 65536) combined with the low halves that decide a
    rounding (ties, one below / above, zero, all ones) -- finite values, denormals, the overflow to infinity -- and a quiet NaN
    for a NaN."""
    import ctypes
    from myriad_amd import _lib as L
    fn = L.load().mh_bf16_round_check
    hi = torch.arange(65536, dtype=torch.int64)
    lows = torch.tensor([0x0000, 0x0001, 0x7fff, 0x8000, 0x8001, 0xffff, 0x1234, 0xfedc], dtype=torch.int64)
    bits = ((hi[:, None] << 16) | lows[None, :]).reshape(-1)
    g = torch.Generator().manual_seed(5)
    bits = torch.cat([bits, torch.randint(0, 2 ** 32, (1 << 20,), generator=g, dtype=torch.int64)])
    x = bits.to(torch.int32, copy=True) if False else (bits & 0xffffffff)
    x = torch.where(x >= 2 ** 31, x - 2 ** 32, x).to(torch.int32).view(torch.float32).to(DEV)
    n = x.numel()
    outs = [torch.empty(n, dtype=torch.int16, device=DEV) for _ in range(3)]
    assert fn(x.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), n, torch.cuda.current_stream().cuda_stream) == 0
    hw, hw_pk, sw = (o.cpu().to(torch.int64) & 0xffff for o in outs)
    nan = torch.isnan(x.cpu())
    assert torch.equal(hw[~nan], sw[~nan]) and torch.equal(hw_pk[~nan], sw[~nan])
    for t in (hw, hw_pk, sw):                              # a NaN stays a (quiet) NaN, sign kept
        tn = t[nan]
        assert bool(((tn & 0x7f80) == 0x7f80).all()) and bool(((tn & 0x007f) != 0).all())
    assert int(nan.sum()) > 1000 and int((~nan).sum()) > 1_000_000


@pytest.mark.parametrize("M,N,K,pad", [(1184, 4160, 256, 0), (1184, 4160, 256, 64), (300, 520, 192, 8), (257, 256, 64, 0), (2056, 1408, 128, 0)])
def test_padding_rows_of_the_256_tile_read_zeros_and_change_nothing(M, N, K, pad):
    """The 256 x 256 kernel reads the rows past M / N as zeros (one row past the end lies outside the buffer descriptor's range)
    instead of copies of the last row -- the padding MFMAs then switch nothing on a power-limited chip.  Same bits as with the
    copies (option gemm_zero_pad = 0), for row strides wider than K and for operands that end exactly at their allocation."""
    import ctypes
    from myriad_amd import _lib as L
    hook = _opt_hook("gemm_zero_pad")
    ops.ensure_workspace(torch.device(DEV))
    a = bf(rnd(M, K + pad, seed=231)).to(DEV)[:, :K]                 # lda = K + pad; the last row ends `pad` elements before the allocation does
    b = bf(rnd(N, K + pad, seed=232) * 0.05).to(DEV)[:, :K]
    bias = rnd(N, seed=233).to(DEV)
    outs = []
    try:
        for zp in (1, 0):
            hook(zp)
            outs.append((ops.gemm(a, b, bias=bias, variant=12), ops.gemm(a, b, out_dtype=torch.float32, variant=12)))
    finally:
        hook(1)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert relerr(outs[0][1], a.float() @ b.float().T) < 6e-3
    assert not torch.isnan(outs[0][1]).any()


@pytest.mark.parametrize("M,N,K", [(1184, 4096, 4096), (1184, 22016, 4096), (2056, 4224, 1408), (1056, 512, 384), (1057, 512, 128),
                                   (160, 256, 64), (257, 256, 192), (1184, 4160, 12288), (300, 328, 192)])
def test_waves_past_the_last_row_skip_their_matrix_instructions_and_change_nothing(M, N, K):
    """256 x 256 kernel: a wave whose 128-row slab has at most two 16-row fragments below M (the second wave row of the LLaMA
    launches' fifth row tile, both wave rows of the ViT's ninth) runs the generated loop without the MFMAs of its other six
    fragments, a wave with no row below M or no column below N (N = 4160, 4224, 328) without any (option gemm_skip_pad;
    gen_gemm_x4.py ni_act).  Same requests / waits / barriers, so the SAME bits as the full loop:
    bf16 and f32 outputs with bias, K split and unsplit, one / two / three k-tiles (all three loop bodies), M that leaves exactly 32,
    33 (no skipping), 160 and 1 rows in the last tile."""
    hook = _opt_hook("gemm_skip_pad")
    ops.ensure_workspace(torch.device(DEV))
    a = bf(rnd(M, K, seed=241)).to(DEV)
    b = bf(rnd(N, K, seed=242) * 0.05).to(DEV)
    bias = rnd(N, seed=243).to(DEV)
    outs = []
    try:
        for on in (1, 0):
            hook(on)
            outs.append((ops.gemm(a, b, bias=bias, variant=12), ops.gemm(a, b, out_dtype=torch.float32, variant=12)))
    finally:
        hook(1)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert relerr(outs[0][1], a.float() @ b.float().T) < 6e-3
    assert not torch.isnan(outs[0][1]).any()


@pytest.mark.parametrize("M,N,K", [(1184, 4096, 4096), (1184, 4096, 22016), (1184, 4160, 12288), (2056, 1408, 6144), (1184, 4096, 11008),
                                   (648, 4096, 8192), (300, 768, 4096)])
def test_k_split_workgroups_grouped_by_split_per_xcd_change_nothing(M, N, K):
    """256 x 256 kernel, K-split launches (option gemm_split_xcd, round 6): which workgroup computes which (tile, split) is re-mapped
    so that one XCD's workgroups share a K range -- a permutation of the grid, the SAME bits in every slab and in the summed result:
    grids whose tile count is / is not a multiple of eight (80, 85 = the LoRA-bordered qkv dgrad, 54, 48, 6), bf16 and f32 outputs
    with bias, and the slab-consuming fused forms through ops.gemm's own split plan."""
    hook = _opt_hook("gemm_split_xcd")
    ops.ensure_workspace(torch.device(DEV))
    a = bf(rnd(M, K, seed=251)).to(DEV)
    b = bf(rnd(N, K, seed=252) * 0.05).to(DEV)
    bias = rnd(N, seed=253).to(DEV)
    k, sp = ops.gemm_plan(M, N, K)
    outs = []
    try:
        for on in (1, 0):
            hook(on)
            outs.append((ops.gemm(a, b, bias=bias), ops.gemm(a, b, out_dtype=torch.float32), ops.gemm(a, b, bias=bias, variant=12),
                         ops.gemm(a, b, out_dtype=torch.float32, variant=12)))
    finally:
        hook(1)
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y)
    assert relerr(outs[0][1], a.float() @ b.float().T) < 6e-3
    assert not torch.isnan(outs[0][1]).any()
    if (M, N, K) != (300, 768, 4096):
        assert k == 2 and sp > 1, "the shape was meant to exercise a K-split launch of plan kernel 2"


@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (8, 12288, 4160), (16, 1000, 11008), (3, 32000, 4096)])
def test_skinny_m_weight_streaming_gemm(M, N, K):
    """M <= 16 routes to the decode weight-streaming kernel (gemv.hip)."""
    a = bf(rnd(M, K, seed=15)).to(DEV)
    b = bf(rnd(N, K, seed=16) * 0.05 + torch.arange(N)[:, None] * 1e-4).to(DEV)
    bias = rnd(N, seed=17).to(DEV)
    res = rnd(M, N, seed=18).to(DEV)
    ref = a.float() @ b.float().T
    assert relerr(ops.gemm(a, b, out_dtype=torch.float32), ref) < 1e-4
    assert relerr(ops.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32, alpha=0.5), 0.5 * ref + bias + res) < 1e-4
    assert relerr(ops.gemm(a, b).float(), ref) < 6e-3
    assert relerr(ops.gemm(a, b, out_dtype=torch.float32), ops.gemm(a, b, out_dtype=torch.float32, variant=1)) < 1e-4


@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (8, 12288, 4160), (16, 1000, 11008), (3, 32000, 4096), (2, 40, 192)])
def test_decode_kernel_on_the_packed_weight_copy_is_bit_identical(M, N, K):
    """mh_gemv_pack + mh_gemv_packed: the decode kernel streaming a pre-permuted copy of the weight (contiguous KiB per
    wave-instruction) -- same k per lane and same reduction order, so the SAME bits as the row-major path, for ragged N
    (1000, 40: padded blocks), a K that does not divide among the waves (4160 = 65 steps), bias / residual / f32 out."""
    a = bf(rnd(M, K, seed=61)).to(DEV)
    b_wide = bf(rnd(N, K + 64, seed=62) * 0.05 + torch.arange(N)[:, None] * 1e-4).to(DEV)
    b = b_wide[:, :K]                                               # ldb > K
    bias = rnd(N, seed=63).to(DEV)
    res = rnd(M, N, seed=64).to(DEV)
    pw = ops.gemv_pack(b)
    assert pw.data.numel() >= N * K
    assert torch.equal(ops.gemv_packed(a, pw), ops.gemm(a, b))
    assert torch.equal(ops.gemv_packed(a, pw, out_dtype=torch.float32), ops.gemm(a, b, out_dtype=torch.float32))
    got = ops.gemv_packed(a, pw, bias=bias, residual=res, out_dtype=torch.float32, alpha=0.5)
    assert torch.equal(got, ops.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32, alpha=0.5))
    assert relerr(got, 0.5 * (a.float() @ b.float().T) + bias + res) < 1e-4
    # re-pack in place after the weight moved (the LoRA border of the qkv weight does between generate() calls)
    b2 = bf(rnd(N, K, seed=65) * 0.05).to(DEV)
    ops.gemv_pack(b2, out=pw)
    assert torch.equal(ops.gemv_packed(a, pw), ops.gemm(a, b2))
    with pytest.raises(RuntimeError):
        ops.gemv_packed(bf(rnd(17, K)).to(DEV), pw)


@pytest.mark.parametrize("M,N,K", [(1, 12288, 4096), (2, 4096, 4096), (2, 22016, 4096), (1, 32000, 4096), (1, 1000, 2048)])
def test_decode_gemv_with_the_rmsnorm_fused_in_is_bit_identical(M, N, K):
    """mh_gemv_packed_rmsnorm: every workgroup rebuilds the normalised rows (rmsnorm_fwd_kernel's summation order and
    expression) in LDS, then streams the packed weight -- the SAME bits as mh_rmsnorm_fwd + mh_gemv_packed, with / without the
    fp32 residual, bf16 and f32 outputs, both wave counts (N / 16 below and above 512 workgroups)."""
    h = (rnd(M, K, seed=71) * 1.7).to(DEV)
    w = (1 + 0.1 * rnd(K, seed=72)).to(DEV)
    b = bf(rnd(N, K, seed=73) * 0.05).to(DEV)
    res = rnd(M, N, seed=74).to(DEV)
    pw = ops.gemv_pack(b)
    xn = ops.rmsnorm_fwd(h, w, 1e-6)
    assert torch.equal(ops.gemv_packed_rmsnorm(h, w, 1e-6, pw), ops.gemv_packed(xn, pw))
    got = ops.gemv_packed_rmsnorm(h, w, 1e-6, pw, residual=res, out_dtype=torch.float32)
    assert torch.equal(got, ops.gemv_packed(xn, pw, residual=res, out_dtype=torch.float32))
    ref = (h.double() * torch.rsqrt(h.double().pow(2).mean(-1, keepdim=True) + 1e-6) * w.double()) @ b.double().T + res.double()
    assert relerr(got, ref) < 2e-2


def test_decode_gemv_fused_forms_refuse_more_than_two_rows():
    """Every workgroup rebuilds all M operand rows, which only pays for one or two (measured: batch-8 decode 6.5 vs 4.1 ms per
    token): beyond that the entry points return MH_ERR_UNSUPPORTED (None here) and the caller runs the two launches."""
    K, N = 4096, 4096
    pw = ops.gemv_pack(bf(rnd(N, K, seed=75) * 0.05).to(DEV))
    w = torch.ones(K, device=DEV)
    assert ops.gemv_packed_rmsnorm(rnd(16, K, seed=76).to(DEV), w, 1e-6, pw) is None
    assert ops.gemv_packed_rmsnorm(rnd(3, K, seed=77).to(DEV), w, 1e-6, pw) is None


@pytest.mark.parametrize("M", [1, 2])
def test_decode_gemv_with_the_silu_gate_fused_in_is_bit_identical(M):
    """mh_gemv_packed_silu on the 128-blocked gate|up layout == mh_silu_mul_fwd_blk + mh_gemv_packed (LLaMA-7B's down
    projection: K = 11008, N = 4096), fp32 residual."""
    I, N = 11008, 4096
    gu = bf(rnd(M, 2 * I, seed=81) * 2.0).to(DEV)
    b = bf(rnd(N, I, seed=82) * 0.05).to(DEV)
    res = rnd(M, N, seed=83).to(DEV)
    pw = ops.gemv_pack(b)
    act = ops.silu_mul_fwd_blk(gu)
    got = ops.gemv_packed_silu(gu, pw, residual=res, out_dtype=torch.float32)
    assert torch.equal(got, ops.gemv_packed(act, pw, residual=res, out_dtype=torch.float32))
    assert torch.equal(ops.gemv_packed_silu(gu, pw), ops.gemv_packed(act, pw))
    assert ops.gemv_packed_silu(bf(rnd(4, 2 * I, seed=84)).to(DEV), pw) is None           # 4 x 22 KiB: does not fit


@pytest.mark.parametrize("B", [1, 3])
def test_decode_attention_with_rotary_and_append_fused_in_is_bit_identical(B):
    """mh_attn_decode_rope == mh_rope_kv_append + mh_attn_fwd(Sq = 1): same rotated q (in place), same cache row, same output."""
    H, hd, T, pos = 32, 128, 192, 41
    W = H * hd
    qkv0 = bf(rnd(B, 3 * W, seed=91)).to(DEV)
    cache0 = bf(rnd(B, T, 2 * W, seed=92)).to(DEV)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    ang = torch.arange(T).float()[:, None] * inv[None]
    cos, sin = ang.cos().contiguous().to(DEV), ang.sin().contiguous().to(DEV)
    pos_t = torch.full((B,), pos, dtype=torch.int32, device=DEV)
    kvlen = torch.full((B,), pos + 1, dtype=torch.int32, device=DEV)
    scale = 1.0 / hd ** 0.5
    qa, ca = qkv0.clone(), cache0.clone()
    ops.rope_kv_append(qa, H, hd, pos_t, cos, sin, ca, pos_t)
    oa, _ = ops.attn_fwd(qa.view(B, 1, 3 * W)[:, :, :W], ca[:, :, :W], ca[:, :, W:], H, hd, scale, causal=False, kv_len=kvlen, need_lse=False)
    qb, cb = qkv0.clone(), cache0.clone()
    ob = ops.attn_decode_rope(qb, cb, pos_t, pos_t, kvlen, cos, sin, H, hd, scale)
    assert torch.equal(qa[:, :W], qb[:, :W]) and torch.equal(ca, cb)
    assert torch.equal(oa.view(B, W), ob)
    assert not torch.equal(cb[:, pos], cache0[:, pos])                                      # the row was really written


@pytest.mark.parametrize("M,N,K", [(1184, 4096, 4096), (1184, 4096, 11008), (300, 512, 256), (2056, 1408, 6144), (148, 4096, 11008)])
def test_gemm_residual_rmsnorm_is_bit_identical_to_two_launches(M, N, K):
    """Split-K reduce + residual add + RMSNorm in one kernel (split shapes) or GEMM then norm (the rest): same bits."""
    ops.ensure_workspace(DEV)
    a = bf(rnd(M, K, seed=31)).to(DEV)
    b = bf(rnd(N, K, seed=32) * 0.05).to(DEV)
    res = rnd(M, N, seed=33).to(DEV)
    w = (1 + 0.1 * rnd(N, seed=34)).to(DEV)
    h_ref = ops.gemm(a, b, residual=res, out_dtype=torch.float32)
    y_ref = ops.rmsnorm_fwd(h_ref, w, 1e-6)
    h, y = ops.gemm_residual_rmsnorm(a, b, res, w, 1e-6)
    assert torch.equal(h, h_ref) and torch.equal(y, y_ref)
    wide = torch.zeros(M, N + 64, dtype=torch.bfloat16, device=DEV)       # the bordered LoRA operand: strided y
    h2, y2 = ops.gemm_residual_rmsnorm(a, b, res, w, 1e-6, y_out=wide[:, :N])
    assert torch.equal(h2, h_ref) and torch.equal(wide[:, :N], y_ref) and wide[:, N:].abs().max() == 0
    assert relerr(h, a.float() @ b.float().T + res) < slab_tol(M, N, K, 1e-4)


@pytest.mark.parametrize("M,N,K", [(2056, 1408, 6144), (2056, 1408, 1408), (300, 512, 256)])
def test_gemm_residual_layernorm_is_bit_identical_to_two_launches(M, N, K):
    ops.ensure_workspace(DEV)
    a = bf(rnd(M, K, seed=41)).to(DEV)
    b = bf(rnd(N, K, seed=42) * 0.05).to(DEV)
    bias = rnd(N, seed=43).to(DEV)
    res = rnd(M, N, seed=44).to(DEV)
    w, nb = (1 + 0.1 * rnd(N, seed=45)).to(DEV), (0.1 * rnd(N, seed=46)).to(DEV)
    h_ref = ops.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32)
    y_ref, _ = ops.layernorm_fwd(h_ref, w, nb, 1e-6)
    h, y = ops.gemm_residual_layernorm(a, b, bias, res, w, nb, 1e-6)
    assert torch.equal(h, h_ref) and torch.equal(y, y_ref)
    ref = torch.nn.functional.layer_norm(a.float() @ b.float().T + bias + res, (N,), w, nb, 1e-6)
    assert relerr(y.float(), ref) < 1e-2


@pytest.mark.parametrize("M,N,K", [(1184, 4096, 22016), (1184, 4096, 4096), (300, 512, 256), (148, 4096, 11008), (37, 64, 128)])
def test_gemm_rmsnorm_bwd_is_bit_identical_to_two_launches(M, N, K):
    """A dgrad Linear and the RMSNorm backward that consumes it: the split-K slabs are summed inside the norm kernel (split
    shapes) or the GEMM writes dY and the norm kernel reads it (the rest) -- same bits either way; and against torch autograd."""
    ops.ensure_workspace(DEV)
    a = bf(rnd(M, K, seed=71)).to(DEV)
    b = bf(rnd(N, K, seed=72) * 0.05).to(DEV)
    x = rnd(M, N, seed=73).to(DEV)
    w = (1 + 0.1 * rnd(N, seed=74)).to(DEV)
    dres = rnd(M, N, seed=75).to(DEV)
    dy = ops.gemm(a, b, out_dtype=torch.float32)
    dx_ref, dxb_ref = ops.rmsnorm_bwd(dy, x, w, 1e-6, dres=dres, want_bf16=True)
    dx, dxb = ops.gemm_rmsnorm_bwd(a, b, x, w, 1e-6, dres=dres)
    assert torch.equal(dx, dx_ref) and torch.equal(dxb, dxb_ref)
    dx2, _ = ops.gemm_rmsnorm_bwd(a, b, x, w, 1e-6, want_bf16=False)
    assert torch.equal(dx2, ops.rmsnorm_bwd(dy, x, w, 1e-6)[0])
    xt = x.clone().requires_grad_(True)
    y = xt * torch.rsqrt(xt.pow(2).mean(-1, keepdim=True) + 1e-6) * w
    y.backward(a.float() @ b.float().T)
    assert relerr(dx, xt.grad + dres) < slab_tol(M, N, K, 2e-4)


def test_gemm_rejects_bad_k():
    a = bf(rnd(8, 40)).to(DEV)
    b = bf(rnd(8, 40)).to(DEV)
    with pytest.raises(RuntimeError):
        ops.gemm(a, b)


# ------------------------------------------------------------------------------------------------ attention
def attn_ref(q, k, v, scale, causal, bias, kv_len):
    # q [B,H,Sq,D] fp32 etc.
    s = (q @ k.transpose(-1, -2)) * scale
    Sq, Sk = s.shape[-2:]
    if bias is not None:
        s = s + bias[None]
    mask = torch.zeros(s.shape[0], 1, Sq, Sk, dtype=torch.bool, device=s.device)
    if causal:
        i = torch.arange(Sq, device=s.device)[:, None] + (Sk - Sq)
        j = torch.arange(Sk, device=s.device)[None]
        mask = mask | (j > i)[None, None]
    if kv_len is not None:
        j = torch.arange(Sk, device=s.device)[None, None, None]
        mask = mask | (j >= kv_len[:, None, None, None])
    s = s.masked_fill(mask, float("-inf"))
    p = s.softmax(-1)
    return p @ v


CASES = [
    # name, B, H, Sq, Sk, D, causal, bias, ragged
    ("vit", 2, 16, 257, 257, 88, False, False, False),
    ("vit_bias", 1, 4, 130, 130, 88, False, True, False),
    ("qf_self", 2, 12, 81, 81, 64, False, False, False),
    ("qf_cross", 2, 12, 81, 257, 64, False, False, False),
    ("qf_small", 1, 12, 32, 32, 64, False, False, False),
    ("llama", 2, 32, 148, 148, 128, True, False, True),
    ("llama_short", 3, 4, 12, 12, 16, True, False, True),
    ("decode", 2, 32, 1, 150, 128, True, False, False),
    ("decode_ragged", 3, 8, 1, 77, 128, False, False, True),     # KV cache with device-side valid lengths
    ("decode_d64", 2, 12, 1, 33, 64, True, False, False),
]


@pytest.mark.parametrize("name,B,H,Sq,Sk,D,causal,use_bias,ragged", CASES)
def test_attention_fwd_bwd(name, B, H, Sq, Sk, D, causal, use_bias, ragged):
    W = H * D
    # token-major fused qkv buffer like the projection GEMM writes: [B, S, 3W]
    qkv_q = bf(rnd(B, Sq, W, seed=11)).to(DEV)
    kv = bf(rnd(B, Sk, 2 * W, seed=12)).to(DEV)
    k, v = kv[:, :, :W], kv[:, :, W:]
    scale = D ** -0.5
    bias = rnd(H, Sq, Sk, seed=13).to(DEV) if use_bias else None
    kv_len = None
    if ragged:
        kv_len = torch.tensor([Sk - (3 * i) % max(1, Sk // 2) for i in range(B)], dtype=torch.int32, device=DEV)
    o, lse = ops.attn_fwd(qkv_q, k, v, H, D, scale, causal=causal, bias=bias, kv_len=kv_len)

    def heads(t, S):
        return t.float().view(B, S, H, D).transpose(1, 2).detach().requires_grad_(True)

    qf, kf, vf = heads(qkv_q, Sq), heads(k, Sk), heads(v, Sk)
    ref = attn_ref(qf, kf, vf, scale, causal, bias, None if kv_len is None else kv_len.long())
    ref_tok = ref.transpose(1, 2).reshape(B, Sq, W)
    # padded query rows (q >= kv_len under right padding) are defined but never consumed; compare valid rows
    valid = torch.ones(B, Sq, dtype=torch.bool, device=DEV)
    if kv_len is not None and Sq == Sk:
        valid = torch.arange(Sq, device=DEV)[None] < kv_len[:, None]
    assert relerr(o.float()[valid], ref_tok[valid]) < 1.5e-2, name
    if Sq == 1:
        return
    dout = bf(rnd(B, Sq, W, seed=14)).to(DEV)
    dout = dout * valid[..., None]
    dq, dk, dv = ops.attn_bwd(qkv_q, k, v, o, dout, lse, H, D, scale, causal=causal, bias=bias, kv_len=kv_len)
    (ref_tok * dout.float()).sum().backward()
    for got, want, nm in ((dq, qf.grad, "dq"), (dk, kf.grad, "dk"), (dv, vf.grad, "dv")):
        S = want.shape[2]
        want_tok = want.transpose(1, 2).reshape(B, S, W)
        m = valid if S == Sq else torch.ones(B, S, dtype=torch.bool, device=DEV)
        assert relerr(got.float()[m], want_tok[m]) < 2.5e-2, (name, nm)


@pytest.mark.parametrize("name,B,H,Sq,Sk,D", [("vit_g", 8, 16, 257, 257, 88), ("vit_b1", 1, 16, 257, 257, 88), ("qf_self", 8, 12, 81, 81, 64),
                                              ("qf_cross", 8, 12, 81, 257, 64), ("qf_self32", 2, 12, 32, 32, 64), ("ragged", 3, 5, 100, 288, 72),
                                              ("one_frag", 2, 4, 7, 19, 40)])
def test_whole_sequence_encoder_attention_vs_reference_and_the_tiled_kernel(name, B, H, Sq, Sk, D):
    """csrc/attn_full.hip (one workgroup stages ALL keys and values of an (image, head); the frozen encoders' unmasked attention:
    eva_vit.py:118-148 at head dim 88 -> 96 padded, Qformer.py:169-275 at 64) against an fp32 torch softmax(q k^T * scale) v and
    against the 64x64-tile kernel it replaces (option attn_full = 0): outputs within the bf16 class of each other, the
    log-sum-exp the Q-Former backward reads within 1e-3.  q / k / v are strided views of token-major projection outputs (head h
    at columns h * D, rows wider than the heads), as the models pass them."""
    W = H * D
    qkv = bf(rnd(B, Sq, 3 * W + 16, seed=301)).to(DEV)              # q inside a wider row (ld = 3W + 16)
    kv = bf(rnd(B, Sk, 2 * W, seed=302)).to(DEV)
    q, k, v = qkv[:, :, :W], kv[:, :, :W], kv[:, :, W:]
    scale = D ** -0.5
    hook = _opt_hook("attn_full")
    try:
        hook(1)
        o1, l1 = ops.attn_fwd(q, k, v, H, D, scale)
        hook(0)
        o0, l0 = ops.attn_fwd(q, k, v, H, D, scale)
    finally:
        hook(1)

    def heads(t, S):
        return t.float().view(B, S, H, D).transpose(1, 2)

    ref = attn_ref(heads(q, Sq), heads(k, Sk), heads(v, Sk), scale, False, None, None).transpose(1, 2).reshape(B, Sq, W)
    assert relerr(o1.float(), ref) < 1.5e-2, name
    assert relerr(o1.float(), o0.float()) < 1.5e-2 and float((l1 - l0).abs().max()) < 1e-3, name
    sc = torch.einsum("bhqd,bhkd->bhqk", heads(q, Sq), heads(k, Sk)) * scale
    assert float((l1 - torch.logsumexp(sc, -1)).abs().max()) < 1e-3
    if Sk > 64:                                                 # more than one key tile: the two kernels sum in different orders,
        assert not torch.equal(o1, o0)                          # so bit-equal outputs would mean the switch did not switch


def test_attention_online_softmax_spike():
    """Force a large running-max jump between KV tiles (guide rule 26): one key row spikes late."""
    B, H, S, D = 1, 2, 200, 64
    q = rnd(B, S, H * D, seed=21)
    k = rnd(B, S, H * D, seed=22)
    k[:, 170] = q[:, 5] * 6.0  # key in the 3rd tile strongly matches query 5
    v = rnd(B, S, H * D, seed=23)
    q, k, v = bf(q).to(DEV), bf(k).to(DEV), bf(v).to(DEV)
    o, _ = ops.attn_fwd(q, k, v, H, D, D ** -0.5)

    def heads(t):
        return t.float().view(B, S, H, D).transpose(1, 2)

    ref = attn_ref(heads(q), heads(k), heads(v), D ** -0.5, False, None, None).transpose(1, 2).reshape(B, S, H * D)
    assert relerr(o.float(), ref) < 1.5e-2


@pytest.mark.parametrize("M,I,K", [(1184, 11008, 4096), (300, 256, 128), (148, 1024, 512), (148, 11008, 4096), (257, 11008, 4096),
                                   (648, 11008, 4096)])
def test_swiglu_fused_into_the_mlp_gemms(M, I, K):
    """mh_gemm_swiglu_fwd/bwd (SiLU-gated product in the gate|up GEMM's epilogue, its backward in the down dgrad's) against
    an fp32 torch model of LlamaMLP (modeling_llama.py:139-140) and bit-for-bit against GEMM + silu kernels.  The fused
    epilogues are the default since the read-out rounds in hardware (MYRIAD_SWIGLU_FUSED=0: the separate launches); the option
    selects them explicitly.  At the batch-1 step's row counts (148, 257) the down dgrad is a K-split launch of the 160-row tiles and
    the gate backward sums its slabs itself: the same comparison covers that form."""
    import ctypes
    from myriad_amd import _lib as L
    hook = _opt_hook("swiglu_fused")
    ops.ensure_workspace(torch.device(DEV))
    hook(1)
    try:
        _swiglu_case(M, I, K)
    finally:
        hook(1)


def _swiglu_case(M, I, K):
    x = bf(rnd(M, K, seed=71, scale=0.5)).to(DEV)
    wg = bf(rnd(I, K, seed=72, scale=0.05)).to(DEV)
    wu = bf(rnd(I, K, seed=73, scale=0.05) + torch.arange(I)[:, None] * 1e-5).to(DEV)
    wgu = ops.interleave_gate_up(wg, wu)
    gu, act = ops.gemm_swiglu_fwd(x, wgu)
    g_ref, u_ref = x.float() @ wg.float().T, x.float() @ wu.float().T
    blk = ops.SWIGLU_BLK
    gu3 = gu.float().view(M, I // blk, 2, blk)
    assert relerr(gu3[:, :, 0].reshape(M, I), g_ref) < 6e-3 and relerr(gu3[:, :, 1].reshape(M, I), u_ref) < 6e-3
    assert relerr(act.float(), F.silu(g_ref) * u_ref) < 1.5e-2
    gu2 = ops.gemm(x, wgu)
    if (M, I) == (648, 11008):
        # 258 tiles on 256 CUs (the MiniGPT-4 arch): the fused launch takes the 85 column blocks that fill one round, the left-over
        # 256 columns (one gate | up pair of blocks) go through a K-split product on their slice: bit-equal where fused, split-K
        # rounding (fp32 partial sums, one bf16 rounding) on the slice, and the gate is applied to the pre-activations as stored
        n1 = 85 * 256
        assert torch.equal(gu[:, :n1], gu2[:, :n1]) and relerr(gu[:, n1:].float(), gu2[:, n1:].float()) < 4e-3
        assert torch.equal(act, ops.silu_mul_fwd_blk(gu))
    else:
        assert torch.equal(gu, gu2) and torch.equal(act, ops.silu_mul_fwd_blk(gu2))
    # backward: dh [M, D] against the down projection's transposed weight [I, D]
    D = 256 if K <= 512 else 4096
    dh = bf(rnd(M, D, seed=74, scale=0.1)).to(DEV)
    wdT = bf(rnd(I, D, seed=75, scale=0.05)).to(DEV)
    dgu = ops.gemm_swiglu_bwd(dh, wdT, gu)
    dact = ops.gemm(dh, wdT)
    want = torch.empty_like(gu)
    from myriad_amd import _lib
    _lib.check(_lib.load().mh_silu_mul_bwd_blk(dact.data_ptr(), gu.data_ptr(), want.data_ptr(), M, I, blk,
                                               torch.cuda.current_stream().cuda_stream), "mh_silu_mul_bwd_blk")
    assert torch.equal(dgu, want)
    g32 = gu3[:, :, 0].reshape(M, I).clone().requires_grad_(True)
    u32 = gu3[:, :, 1].reshape(M, I).clone().requires_grad_(True)
    (F.silu(g32) * u32 * (dh.float() @ wdT.float().T)).sum().backward()
    d3 = dgu.float().view(M, I // blk, 2, blk)
    assert relerr(d3[:, :, 0].reshape(M, I), g32.grad) < 2e-2 and relerr(d3[:, :, 1].reshape(M, I), u32.grad) < 2e-2


@pytest.mark.parametrize("M,N,K,D", [(648, 3072, 768, 768), (81, 3072, 768, 768), (257, 768, 256, 128), (50, 132, 64, 64),
                                     (2056, 6144, 1408, 1408), (16, 3072, 768, 768)])
def test_gelu_fused_into_the_mlp_gemms(M, N, K, D):
    """mh_gemm_gelu_fwd/bwd (erf-GELU in the first product's epilogue with the pre-activation kept, gelu' in the second dgrad's
    epilogue: Qformer.py:481-484) against an fp32 torch model and bit-for-bit against GEMM + mh_gelu_fwd / mh_gelu_bwd -- on the
    plan kernels that fuse (648 / 81 / 257 / 50 rows: 128x64, 64x64 and 128x128 tiles, ragged edges) and on shapes whose plan
    does not (the ViT's 256-tile product, a 16-row GEMV), where the entry points run the two launches themselves."""
    import ctypes
    from myriad_amd import _lib as L
    hook = _opt_hook("gelu_fused")
    ops.ensure_workspace(torch.device(DEV))
    x = bf(rnd(M, K, seed=171, scale=0.5)).to(DEV)
    w1 = bf(rnd(N, K, seed=172, scale=0.08)).to(DEV)
    b1 = rnd(N, seed=173, scale=0.3).to(DEV)
    dy = bf(rnd(M, D, seed=174, scale=0.1)).to(DEV)
    w2T = bf(rnd(N, D, seed=175, scale=0.05)).to(DEV)
    res = {}
    for fused in (1, 0):
        hook(fused)
        try:
            pre, act = ops.gemm_gelu_fwd(x, w1, b1)
            dpre = ops.gemm_gelu_bwd(dy, w2T, pre)
        finally:
            hook(1)
        res[fused] = (pre, act, dpre)
    for a, b in zip(res[1], res[0]):
        assert torch.equal(a, b)
    pre, act, dpre = res[1]
    pre2 = ops.gemm(x, w1, bias=b1)                        # and against the public separate launches
    assert torch.equal(pre, pre2) and torch.equal(act, ops.gelu_fwd(pre2))
    assert torch.equal(dpre, ops.gelu_bwd(ops.gemm(dy, w2T), pre2))
    pre_ref = x.float() @ w1.float().T + b1
    assert relerr(pre.float(), pre_ref) < 6e-3
    assert relerr(act.float(), F.gelu(pre_ref)) < 1.2e-2
    p32 = pre.float().clone().requires_grad_(True)
    (F.gelu(p32) * (dy.float() @ w2T.float().T)).sum().backward()
    assert relerr(dpre.float(), p32.grad) < 1.2e-2


def _rope_tables(D, max_pos=256):
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.einsum("i,j->ij", torch.arange(max_pos).float(), inv)
    return fr.cos().contiguous().to(DEV), fr.sin().contiguous().to(DEV)


def _rope_ref(x, pos, cos, sin):
    """reference modeling_llama.py:109-123: x [B,H,S,D] fp32, rotate-half form."""
    D = x.shape[-1]
    c = torch.cat([cos[pos], cos[pos]], -1)[:, None]      # [B,1,S,D]
    s_ = torch.cat([sin[pos], sin[pos]], -1)[:, None]
    rot = torch.cat([-x[..., D // 2:], x[..., :D // 2]], -1)
    return x * c + rot * s_


@pytest.mark.parametrize("B,H,S,ragged,border", [(2, 3, 148, False, 64), (2, 2, 160, True, 0), (1, 4, 81, False, 0),
                                                 (3, 2, 17, True, 64), (1, 1, 1, False, 0), (2, 2, 99, True, 0)])
def test_attention_rope_whole_sequence(B, H, S, ragged, border):
    """mh_attn_rope_fwd/bwd (rotary fused, one workgroup per (b, h)) vs an fp32 torch model of
    apply_rotary_pos_emb + LlamaAttention (modeling_llama.py:109-123, 168-231), and vs the tile kernels + mh_rope_inplace."""
    D = 128
    W = H * D
    assert ops.attn_rope_supported(S, D)
    ld = 3 * W + border                                  # the LoRA-bordered qkv GEMM output has a wider row
    qkv_full = bf(rnd(B, S, ld, seed=51)).to(DEV)
    qkv = qkv_full[:, :, :ld]
    cos, sin = _rope_tables(D)
    pos = (torch.arange(S, dtype=torch.int32)[None] + torch.arange(B, dtype=torch.int32)[:, None] * 3).contiguous().to(DEV)
    kv_len = None
    if ragged:
        kv_len = torch.tensor([max(1, S - (5 * i + 2) % max(1, S // 2)) for i in range(B)], dtype=torch.int32, device=DEV)
    scale = D ** -0.5
    o, lse = ops.attn_rope_fwd(qkv, H, D, scale, pos.view(-1), cos, sin, kv_len=kv_len)

    def heads(t):
        return t.float().reshape(B, S, H, D).transpose(1, 2).detach().requires_grad_(True)

    qf, kf, vf = heads(qkv[:, :, :W]), heads(qkv[:, :, W:2 * W]), heads(qkv[:, :, 2 * W:3 * W])
    pl = pos.long()
    ref = attn_ref(_rope_ref(qf, pl, cos, sin), _rope_ref(kf, pl, cos, sin), vf, scale, True, None,
                   None if kv_len is None else kv_len.long())
    ref_tok = ref.transpose(1, 2).reshape(B, S, W)
    valid = torch.ones(B, S, dtype=torch.bool, device=DEV)
    if kv_len is not None:
        valid = torch.arange(S, device=DEV)[None] < kv_len[:, None]
    assert relerr(o.float()[valid], ref_tok[valid]) < 1.5e-2
    # the tile kernels on the rotated copy give the same attention (both round q, k to bf16 after the rotation)
    rot = qkv.contiguous().clone()
    ops.rope_(rot.view(B * S, ld), 0, 2 * H, D, pos.view(-1), cos, sin, 1.0)
    o2, lse2 = ops.attn_fwd(rot[:, :, :W], rot[:, :, W:2 * W], rot[:, :, 2 * W:3 * W], H, D, scale, causal=True, kv_len=kv_len)
    assert relerr(o.float()[valid], o2.float()[valid]) < 8e-3
    assert relerr(lse.transpose(1, 2)[valid], lse2.transpose(1, 2)[valid]) < 2e-3
    if S == 1:
        return
    dout = bf(rnd(B, S, W, seed=52)).to(DEV) * valid[..., None]
    dqkv = ops.attn_rope_bwd(qkv, o, dout, lse, H, D, scale, pos.view(-1), cos, sin, kv_len=kv_len)
    (ref_tok * dout.float()).sum().backward()
    for i, (want, nm) in enumerate(((qf.grad, "dq"), (kf.grad, "dk"), (vf.grad, "dv"))):
        want_tok = want.transpose(1, 2).reshape(B, S, W)
        got = dqkv[:, :, i * W:(i + 1) * W].float()
        assert relerr(got[valid], want_tok[valid]) < 2.5e-2, nm


def test_attention_rope_bwd_two_workgroups_per_head_same_bits():
    """Small batches run the LLaMA attention backward as two workgroups per (batch, head) (dq | dk, dv): the same bits as one."""
    import ctypes
    from myriad_amd import _lib
    lib = _lib.load()
    B, H, S, D = 2, 4, 148, 128
    W = H * D
    qkv = bf(rnd(B, S, 3 * W, seed=61)).to(DEV)
    cos, sin = _rope_tables(D)
    pos = torch.arange(S, dtype=torch.int32).repeat(B).to(DEV)
    scale = D ** -0.5
    o, lse = ops.attn_rope_fwd(qkv, H, D, scale, pos, cos, sin)
    dout = bf(rnd(B, S, W, seed=62)).to(DEV)
    try:
        lib.mh_set_option(b"attn_bwd_split", 0)
        one = ops.attn_rope_bwd(qkv, o, dout, lse, H, D, scale, pos, cos, sin).clone()
        lib.mh_set_option(b"attn_bwd_split", 1)
        two = ops.attn_rope_bwd(qkv, o, dout, lse, H, D, scale, pos, cos, sin).clone()
    finally:
        lib.mh_set_option(b"attn_bwd_split", 1)
    assert torch.equal(one, two)


def test_gemm_attention_rope_bwd_reads_split_k_slabs():
    """mh_gemm_attn_rope_bwd: the o_proj dgrad's split-K slabs summed inside the attention backward == gemm + attn_rope_bwd."""
    ops.ensure_workspace(torch.device(DEV))
    B, H, S, D = 8, 32, 148, 128
    W = H * D
    qkv = bf(rnd(B, S, 3 * W + 64, seed=61, scale=0.5)).to(DEV)
    cos, sin = _rope_tables(D)
    pos = torch.arange(S, dtype=torch.int32).repeat(B).to(DEV)
    scale = D ** -0.5
    o, lse = ops.attn_rope_fwd(qkv, H, D, scale, pos, cos, sin)
    a = bf(rnd(B * S, W, seed=62, scale=0.1)).to(DEV)
    bw = bf(rnd(W, W, seed=63, scale=0.05)).to(DEV)
    assert ops.gemm_plan(B * S, W, W, out_f32=True)[1] > 1       # the policy splits K for this shape
    do = ops.gemm(a, bw)
    want = ops.attn_rope_bwd(qkv, o, do, lse, H, D, scale, pos, cos, sin)
    got = ops.gemm_attn_rope_bwd(a, bw, qkv, o, lse, H, D, scale, pos, cos, sin)
    assert torch.equal(got[:, :, :3 * W], want[:, :, :3 * W])


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("M,D", [(37, 4096), (5, 64), (300, 1408)])
def test_rmsnorm(M, D):
    x = rnd(M, D, seed=31).to(DEV).requires_grad_(True)
    w = (1 + 0.1 * rnd(D, seed=32)).to(DEV)
    y = ops.rmsnorm_fwd(x.detach(), w, 1e-6)
    ref = R.rms_norm(x, w, 1e-6)
    assert relerr(y.float(), ref) < 5e-3
    dy = rnd(M, D, seed=33).to(DEV)
    dres = rnd(M, D, seed=34).to(DEV)
    (ref * dy).sum().backward()
    dx, dxb = ops.rmsnorm_bwd(dy, x.detach(), w, 1e-6, dres=dres, want_bf16=True)
    assert relerr(dx, x.grad + dres) < 1e-5
    assert relerr(dxb.float(), x.grad + dres) < 5e-3


@pytest.mark.parametrize("M,D,eps", [(257, 1408, 1e-6), (81, 768, 1e-12), (6, 64, 1e-5)])
def test_layernorm(M, D, eps):
    x = rnd(M, D, seed=41).to(DEV).requires_grad_(True)
    w = (1 + 0.1 * rnd(D, seed=42)).to(DEV)
    b = (0.1 * rnd(D, seed=43)).to(DEV)
    yb, yf = ops.layernorm_fwd(x.detach(), w, b, eps, want_bf16=True, want_f32=True)
    ref = F.layer_norm(x, (D,), w, b, eps)
    assert relerr(yf, ref) < 1e-5
    assert relerr(yb.float(), ref) < 5e-3
    dy = rnd(M, D, seed=44).to(DEV)
    (ref * dy).sum().backward()
    dx, _ = ops.layernorm_bwd(dy, x.detach(), w, eps)
    assert relerr(dx, x.grad) < 2e-5


# ------------------------------------------------------------------------------------------------ elementwise
def test_rope_kv_append_equals_rope_then_append():
    """The fused decode-token kernel must be bit-identical to rope_ + kv_append (same fp32 arithmetic, one rounding)."""
    B, H, D, T = 3, 4, 128, 20
    W = H * D
    qkv = bf(rnd(B, 3 * W, seed=71)).to(DEV)
    cos, sin = R.rotary_tables(D, 64)
    cs, sn = cos[:, :D // 2].contiguous().to(DEV), sin[:, :D // 2].contiguous().to(DEV)
    pos = torch.full((B,), 7, dtype=torch.int32, device=DEV)
    pos_dev = torch.tensor([7], dtype=torch.int32, device=DEV)
    cache_a = bf(rnd(B, T, 2 * W, seed=72)).to(DEV)
    cache_b = cache_a.clone()
    a = qkv.clone()
    ops.rope_(a, 0, 2 * H, D, pos, cs, sn, 1.0)
    ops.kv_append(a[:, W:], cache_a, pos_dev)
    b = qkv.clone()
    ops.rope_kv_append(b, H, D, pos, cs, sn, cache_b, pos_dev)
    assert torch.equal(a[:, :W], b[:, :W])                       # q rotated in place
    assert torch.equal(cache_a, cache_b)                         # k (rotated) | v in row 7, every other row untouched
    assert torch.equal(b[:, W:], qkv[:, W:])                     # the fused kernel leaves the k|v source alone


def test_rope_fwd_bwd():
    B, S, H, D = 2, 40, 4, 128
    x = bf(rnd(B * S, 3 * H * D, seed=51)).to(DEV)
    cos, sin = R.rotary_tables(D, 256)
    pos = (torch.arange(S).repeat(B) + 3).int().to(DEV)
    cs, sn = cos[:, :D // 2].contiguous().to(DEV), sin[:, :D // 2].contiguous().to(DEV)
    y = x.clone()
    ops.rope_(y, 0, 2 * H, D, pos, cs, sn, 1.0)  # q and k heads, v untouched
    q = x[:, :H * D].float().view(B, S, H, D).transpose(1, 2)
    k = x[:, H * D:2 * H * D].float().view(B, S, H, D).transpose(1, 2)
    qr, kr = R.apply_rotary(q.cpu(), k.cpu(), cos, sin, pos.view(B, S).long().cpu())
    assert relerr(y[:, :H * D].float(), qr.transpose(1, 2).reshape(B * S, H * D)) < 5e-3
    assert relerr(y[:, H * D:2 * H * D].float(), kr.transpose(1, 2).reshape(B * S, H * D)) < 5e-3
    assert torch.equal(y[:, 2 * H * D:], x[:, 2 * H * D:])
    # backward == inverse rotation: rope(-1)(rope(+1)(x)) == x up to bf16 rounding
    ops.rope_(y, 0, 2 * H, D, pos, cs, sn, -1.0)
    assert relerr(y.float(), x.float()) < 1e-2


def test_gelu_keeps_sign_and_magnitude_in_the_negative_tail():
    """nn.GELU (erf form: eva_vit.py:54-61, Qformer.py FFN) for x in [-8, -3]: the values are 1e-3 .. 1e-15 and an erf
    approximation with 1.5e-7 ABSOLUTE error used as 1 + erf(u) loses them (wrong magnitude from x = -5 on, possibly a positive
    sign).  The library evaluates the erfc form: relative error <= 2 % over the whole tail, never positive."""
    x = torch.arange(-8.0, -2.999, 1.0 / 32)                               # exact in bf16
    n = x.numel()
    a = torch.zeros(n, 64)
    a[:, 0] = x
    b = torch.zeros(64, 64)
    b[0, 0] = 1.0
    y = ops.gemm(bf(a).to(DEV), bf(b).to(DEV), gelu=True, out_dtype=torch.float32)[:, 0].cpu().double()
    xd = x.double()
    want = xd * 0.5 * torch.special.erfc(-xd / 2 ** 0.5)
    assert bool((y <= 0).all()) and bool((y < 0)[x > -7.5].all())
    rel = ((y - want).abs() / want.abs())
    assert float(rel.max()) < 2e-2, float(rel.max())
    xp = torch.cat([x, x[-7:]])                                            # 168 values: the elementwise kernel takes multiples of 8
    g = ops.gelu_fwd(bf(xp.view(1, -1)).to(DEV)).float().cpu().double().view(-1)[:n]   # the elementwise form: the same function,
    assert float(((g - want).abs() / want.abs()).max()) < 1e-2 + 2 ** -8              # its output rounded to bf16


def test_silu_mul_and_gelu():
    M, I = 50, 11008
    gu = bf(rnd(M, 2 * I, seed=61)).to(DEV)
    g, u = gu[:, :I].float().requires_grad_(True), gu[:, I:].float().requires_grad_(True)
    h = ops.silu_mul_fwd(gu)
    ref = F.silu(g) * u
    assert relerr(h.float(), ref) < 5e-3
    dh = bf(rnd(M, I, seed=62)).to(DEV)
    (ref * dh.float()).sum().backward()
    dgu = ops.silu_mul_bwd(dh, gu)
    assert relerr(dgu[:, :I].float(), g.grad) < 6e-3
    assert relerr(dgu[:, I:].float(), u.grad) < 6e-3
    x = bf(rnd(64, 3072, seed=63)).to(DEV)
    xf = x.float().requires_grad_(True)
    y = ops.gelu_fwd(x)
    assert relerr(y.float(), F.gelu(xf)) < 5e-3
    dy = bf(rnd(64, 3072, seed=64)).to(DEV)
    (F.gelu(xf) * dy.float()).sum().backward()
    assert relerr(ops.gelu_bwd(dy, x).float(), xf.grad) < 6e-3


def test_data_movement():
    x = rnd(70, 200, seed=71).to(DEV)
    t = ops.transpose_to_bf16(x, 64)
    assert t.shape == (200, 128)
    assert relerr(t[:, :70].float(), x.T) < 5e-3 and t[:, 70:].abs().max() == 0
    tb = ops.transpose_to_bf16(bf(x), 64)
    assert torch.equal(tb[:, :70], bf(x).T)
    table = bf(rnd(500, 64, seed=72)).to(DEV)
    ids = torch.randint(0, 500, (33,), generator=torch.Generator().manual_seed(1)).to(DEV)
    rows = torch.randperm(40)[:33].int().to(DEV)
    out = torch.zeros(40, 96, device=DEV)
    ops.embed_gather(table, ids, out[:, 16:80], rows)
    assert torch.equal(out[rows.long(), 16:80], table[ids].float())
    src = rnd(4, 9, 32, seed=73).to(DEV)
    dst = torch.zeros(4, 20, 32, device=DEV)
    ops.copy3d(src, dst[:, 5:14])
    assert torch.equal(dst[:, 5:14], src) and dst[:, :5].abs().max() == 0
    ops.copy3d(src, dst[:, 5:14], accumulate=True)
    assert torch.equal(dst[:, 5:14], 2 * src)
    big = rnd(50, 64, seed=74).to(DEV)
    pick = torch.tensor([3, 7, 49, 0], dtype=torch.int32, device=DEV)
    g = ops.gather_rows_bf16(big, pick)
    assert torch.equal(g, bf(big[pick.long()]))
    back = torch.zeros(50, 64, device=DEV)
    ops.scatter_rows(big[pick.long()].contiguous(), pick, back)
    assert torch.equal(back[pick.long()], big[pick.long()])
    assert relerr(ops.colsum(big), big.sum(0)) < 1e-5
    assert relerr(ops.to_f32(ops.to_bf16(big)), big) < 5e-3


# ------------------------------------------------------------------------------------------------ adaptor / loss / optim
def test_lowrank_adaptor():
    M, D, r = 514, 1408, 4
    x = rnd(M, D, seed=81).to(DEV)
    A = (rnd(r, D, seed=82) * 0.02).to(DEV).requires_grad_(True)
    Bm = (rnd(D, r, seed=83) * 0.02).to(DEV).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    y, t = ops.lowrank_fwd(x, A.detach(), Bm.detach())
    ref = xr + F.linear(F.linear(xr, A), Bm)
    assert relerr(y, ref) < 1e-5
    dy = rnd(M, D, seed=84).to(DEV)
    (ref * dy).sum().backward()
    dA, dB = torch.empty_like(A), torch.empty_like(Bm)
    dx = ops.lowrank_bwd(dy, x, t, A.detach(), Bm.detach(), dA, dB, need_dx=True)
    assert relerr(dA, A.grad) < 1e-4
    assert relerr(dB, Bm.grad) < 1e-4
    assert relerr(dx, xr.grad) < 1e-5


def test_clamp_ce_matches_oracle_including_saturation():
    Rr, V = 12, 32000
    x = rnd(Rr, V, seed=91) * 2
    y = torch.randint(0, V, (Rr,), generator=torch.Generator().manual_seed(2))
    y[3] = -100
    x[0, y[0]] = x[0].min() - 30   # p_t < 1e-7  -> loss 16.118, zero row grad
    x[1, y[1]] = x[1].max() + 40   # p_t > 1-1e-7 -> zero row grad
    xr = x.clone().requires_grad_(True)
    loss_ref = R.clamp_ce_loss(xr, y)
    loss_ref.backward()
    n_valid = int((y != -100).sum())
    row_loss, dlog = ops.clamp_ce(x.to(DEV), y.to(DEV), 1.0 / n_valid)
    loss = ops.sum_f32(row_loss, 1.0 / n_valid)
    assert abs(loss.item() - loss_ref.item()) < 2e-5 * abs(loss_ref.item())
    assert dlog[0].abs().max() == 0 and dlog[1].abs().max() == 0 and dlog[3].abs().max() == 0
    assert relerr(dlog[:, :V].float(), xr.grad) < 6e-3
    am, margin = ops.argmax_rows(x.to(DEV), ban_id=int(x[5].argmax()), want_margin=True)
    x2 = x.clone()
    x2[:, int(x[5].argmax())] = -float("inf")
    assert torch.equal(am.cpu(), x2.argmax(-1))
    t2 = x2.topk(2, -1).values
    assert relerr(margin.cpu(), t2[:, 0] - t2[:, 1]) < 1e-5


@pytest.mark.parametrize("R,V,ban", [(1, 32000, -1), (8, 32000, 2), (3, 1000, 17), (2, 320, -1), (4, 32768, 5)])
def test_greedy_step_in_one_launch_equals_the_two_scans(R, V, ban):
    """mh_argmax_pmax_rows with the row in registers (one launch: arg-max, top-1 / top-2 margin, p_max) against the
    single-workgroup scan kernels it replaces (mh_argmax_rows) and a torch softmax: ids and margins equal, ties included (first
    index wins, a duplicated maximum gives margin 0, the banned id is -inf), p_max = max softmax over the allowed ids within 1e-6."""
    g = torch.Generator().manual_seed(900 + R + V)
    x = torch.randn(R, V, generator=g) * 3
    x[0, V // 3] = x[0].max() + 1.0
    x[0, V // 3 + 7] = x[0, V // 3]                         # a tie: the first index wins, margin 0
    if ban >= 0 and R > 1:
        x[1, ban] = x[1].max() + 5.0                         # the banned id would have won
    xd = x.to(DEV)
    ids0, m0 = ops.argmax_rows(xd, ban_id=ban, want_margin=True)
    ids = torch.empty(R, dtype=torch.long, device=DEV)
    mar = torch.empty(R, dtype=torch.float32, device=DEV)
    pm = torch.empty(R, dtype=torch.float32, device=DEV)
    for inv_temp in (1.0, 0.7):
        ops.argmax_pmax_rows(xd, ids, mar, pm, ban_id=ban, inv_temp=inv_temp)
        assert torch.equal(ids, ids0) and torch.equal(mar, m0)
        x2 = x.clone()
        if ban >= 0:
            x2[:, ban] = -float("inf")
        assert torch.equal(ids.cpu(), x2.argmax(-1))
        want = torch.softmax(x2.double() * inv_temp, -1).max(-1).values
        assert float((pm.cpu().double() - want).abs().max()) < 1e-6
    assert int(ids[0]) == V // 3 and float(mar[0]) == 0.0


def test_adamw_matches_oracle():
    n = 4096 * 3
    p = rnd(n, seed=101)
    g = rnd(n, seed=102)
    m, v = torch.zeros(n), torch.zeros(n)
    pd, md, vd = p.clone().to(DEV), m.clone().to(DEV), v.clone().to(DEV)
    shadow = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    for step in range(1, 4):
        gk = g * step
        R.adamw_step(p, gk, m, v, step, 1e-3, 0.05)
        ops.adamw_step(pd, gk.to(DEV), md, vd, 1e-3, 0.05, step, shadow=shadow)
    assert relerr(pd, p) < 1e-5 and relerr(md, m) < 1e-5 and relerr(vd, v) < 1e-5
    assert relerr(shadow.float(), p) < 5e-3


# ------------------------------------------------------------------------------------------------ conv stack
def test_conv_layer_fwd_bwd_vs_torch():
    B, H, W, Cin, Cout = 2, 28, 28, 16, 64
    x = bf(rnd(B, Cin, H, W, seed=111)).float()
    wt = bf(rnd(Cout, Cin, 3, 3, seed=112) * 0.1).float()
    bias = bf(rnd(Cout, seed=113) * 0.1).float()
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yref = F.conv2d(xr, wr, br, padding=1)
    pref = F.max_pool2d(F.relu(yref), 2)
    x_nhwc = bf(x.permute(0, 2, 3, 1).contiguous()).to(DEV)
    wm = wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(DEV)  # GEMM order (ky,kx,ci)
    wp = ops.conv_pack(wm, bias.to(DEV))
    col = ops.im2col(x_nhwc, 3, 3, 1)
    y = ops.gemm(col, wp, out_dtype=torch.float32)  # [B*H*W, Cout] fp32 pre-activation (+bias via the ones column)
    assert relerr(y.view(B, H, W, Cout).permute(0, 3, 1, 2), yref) < 1e-4
    p = ops.relu_pool_fwd(y, B, H, W, Cout)
    assert relerr(p.float().permute(0, 3, 1, 2), pref) < 8e-3
    dp = rnd(B, Cout, H // 2, W // 2, seed=114)
    (pref * dp).sum().backward()
    dy, _ = ops.relu_pool_bwd(dp.permute(0, 2, 3, 1).contiguous().to(DEV), y, B, H, W, Cout)
    # wgrad: dWp[Cout, Kpad] = dy^T . col
    dyT = ops.transpose_to_bf16(dy, 64)
    colT = ops.transpose_to_bf16(col, 64)
    dWp = ops.gemm(dyT, colT, out_dtype=torch.float32)
    dW = torch.empty(Cout, 9 * Cin, device=DEV)
    db = torch.empty(Cout, device=DEV)
    ops.conv_unpack_grad(dWp, dW, db)
    dW_ref = wr.grad.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)
    assert relerr(dW, dW_ref) < 2e-2
    assert relerr(db, br.grad) < 2e-2
    # dgrad: dcol = dy . Wp  (NT with Wp^T), then col2im
    wpT = ops.transpose_to_bf16(wp, 64)           # [Kpad, Cout]
    dcol = ops.gemm(dy, wpT)                       # [M, Kpad]
    dx = ops.col2im(dcol, B, H, W, Cin, 3, 3, 1)
    assert relerr(dx.permute(0, 3, 1, 2), xr.grad) < 2e-2


def test_context_object_owns_scratch_and_the_gradient_exchange():
    """the opaque mh_ctx.  (1) A split-K GEMM run while a context with its own scratch is current gives the bits of
    the process-default scratch, and leaves that one untouched; (2) mh_allreduce_start / _wait on a one-rank RCCL communicator
    created through the library (id -> init): the sum over one rank is the buffer itself, issued on the context's side stream
    and handed back to the producer stream without a host synchronisation."""
    import ctypes
    from myriad_amd import _lib
    from myriad_amd.runner import CtxCollective
    lib = _lib.load()
    ops.ensure_workspace(DEV)
    a = bf(rnd(1184, 4096, seed=101)).to(DEV)
    b = bf(rnd(4096, 4096, seed=102) * 0.05).to(DEV)
    assert ops.gemm_plan(1184, 4096, 4096)[1] > 1
    want = ops.gemm(a, b, out_dtype=torch.float32).clone()
    cc = CtxCollective(DEV, 0, 1)
    try:
        own = torch.zeros(64 << 20, dtype=torch.uint8, device=DEV)
        _lib.check(lib.mh_ctx_set_workspace(cc.h, None, own.data_ptr(), own.numel()), "mh_ctx_set_workspace")
        _lib.check(lib.mh_ctx_make_current(cc.h), "mh_ctx_make_current")
        got = ops.gemm(a, b, out_dtype=torch.float32)
        torch.cuda.synchronize()
        assert torch.equal(got, want) and int(own.view(torch.int32)[:1024].abs().sum()) != 0     # the context's scratch was used
        _lib.check(lib.mh_ctx_make_current(None), "mh_ctx_make_current")
        assert torch.equal(ops.gemm(a, b, out_dtype=torch.float32), want)
        g = rnd(1 << 20, seed=103).to(DEV)
        ref = g.clone()
        g.mul_(2.0)                                                                              # producer work queued before the exchange
        cc.start(g)
        cc.wait()
        g.add_(1.0)                                                                              # consumer work ordered behind it
        torch.cuda.synchronize()
        assert torch.equal(g, ref * 2.0 + 1.0)
        assert lib.mh_ctx_world(cc.h) == 1
        # round 4: the other two verbs and the bf16 wire, through RCCL's own entry points on the one-rank communicator, queued
        # back to back and covered by ONE wait -- the call sequence of runner.DataParallel mode 'rs_ag' (flags all-reduce,
        # gradient reduce-scatter; later the parameter all-gather)
        for dt in (torch.float32, torch.bfloat16):
            src = rnd(1 << 18, seed=104).to(DEV).to(dt)
            flags = rnd(8, seed=105).to(DEV)
            f0 = flags.clone()
            mine = torch.zeros_like(src)
            back = torch.zeros_like(src)
            cc.start(flags)
            cc.reduce_scatter(src, mine)
            cc.wait()
            cc.all_gather(mine, back)
            cc.wait()
            torch.cuda.synchronize()
            assert torch.equal(flags, f0) and torch.equal(mine, src) and torch.equal(back, src), dt
            h = src.clone()
            cc.start(h)                                                                          # all-reduce with this wire type
            cc.wait()
            torch.cuda.synchronize()
            assert torch.equal(h, src), dt
    finally:
        lib.mh_ctx_make_current(None)
        cc.close()

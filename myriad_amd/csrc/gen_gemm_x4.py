#!/usr/bin/env python3
"""Writes gemm_x4_loop.inc: the K loop of gemm_x4_kernel (gemm_x4.hip) as ONE inline-asm statement per schedule variant.

Why a generator: the loop is ~1200 hand-placed instructions (three copies of a 128-MFMA body whose ds_reads, LDS-DMA
requests, counted waits and barriers sit at fixed MFMA slots).  The schedule is the product; this file is where it is written
down.  Run `python gen_gemm_x4.py` after editing; build.py re-runs it when the .inc is older than this file.
`python gen_gemm_x4.py --sweep` also emits the tuning / knock-out variants that tools/gemm_x4_sweep.py times
(selected at run time through mhdbg_set_gemm_x4_variant; the shipped library carries variant 0 only).

Structure (one wave per SIMD, 4 waves = 2 x 2, wave tile 128 x 128 = 8 x 8 fragments of mfma_f32_16x16x32_bf16,
BK = 64, two 64-KiB LDS buffers [A 256 rows x 128 B | B 256 rows x 128 B]):

  iteration t (buffer cur = t & 1), fragments of k-half 0 already in registers:
    MFMA slot   0..63   A0 x B0      | ds_read k-half 1 of cur (16 reads, one per `rd` slots from slot 0)
                 mid                 | lgkmcnt(0), s_barrier: every wave has read all of cur -> cur is free
    from mid                         | 16 LDS-DMA requests of k-tile t+2 into cur, one per `dma` slots
    MFMA slot  64..127  A1 x B1      |
                 end                 | vmcnt(16): k-tile t+1 (requested one iteration ago) has landed; s_barrier
    from end                         | ds_read k-half 0 of the other buffer (16 reads), waited for at the loop edge

Register map (physical, pinned by the constraints in gemm_x4.hip):
  a[0:255]    accumulators, acc(i, j) = a[(i*8+j)*4 .. +3]   (i: A fragment = 16 rows, j: B fragment = 16 columns)
  v[0:31]  B0   v[32:63]  B1   v[64:95]  A0   v[96:127] A1   (fragments of k-half 0 / 1)
  v[128:135] per-lane global byte offsets of this wave's 8 A requests, v[136:143] the same for B
  v144 / v145  LDS byte address of the lane's A fragment row, k-half 0 / 1 (current buffer);  v146 / v147: B
  s[36:39] / s[40:43]  buffer descriptors of A / B;  s44 k byte offset of the next request;  s45 k-tiles left
  s46  LDS byte address of this wave's first A request slot in the buffer the next requests go to
"""
import os
import sys

BUF = 0x10000     # LDS bytes per buffer
BREG = 0x8000     # offset of the B region inside a buffer

# Schedule parameters of a variant (MFMA slots 0..127 of one iteration):
#   rd      one ds_read_b128 per this many slots          dma     one LDS-DMA request per this many slots
#   mid     slot behind which lgkmcnt(0) + barrier 1 sit  end     slot behind which vmcnt + barrier 2 sit
#   fine    1: the loop edge waits only for the fragments the first MFMA row needs (lgkmcnt(7)); row i of the next
#           iteration waits for its own A fragment with a counted lgkmcnt
#   ko      timing-only knock-outs (WRONG results): "dma", "rd", "bar", "mfma" in any combination
#   order   "snake": the B fragments of odd A rows are walked backwards, so consecutive MFMAs always share one operand (row changes
#           keep B, column changes keep A); "": every row walks B 0..nj-1.  Same sums per accumulator; on a chip whose clock under this
#           loop is set by the matrix pipes' power the operand that does not change is worth 0.6 % of the step (r05_experiments.md)
#   waves   4: 2 x 2 waves, 128 x 128 per wave, one wave per SIMD (256 accumulators, 128 MFMA slots per iteration)
#           8: 2 x 4 waves, 128 x 64 per wave, two waves per SIMD (128 accumulators, 64 MFMA slots per iteration per wave; each
#              wave issues half the requests: 4 + 4, m0 stride 8192) -- what one wave cannot hide under its own MFMAs (the issue
#              time of its LDS-DMA requests and fragment reads) the SIMD's other wave fills
X4 = dict(waves=4, rd=2, dma=3, mid=36, end=96, fine=1, ko="")
X8 = dict(waves=8, rd=1, dma=4, mid=14, end=50, fine=1, ko="", order=os.environ.get("X8_ORDER", "snake"))
DEFAULT = X8        # shipped: tools/gemm_x4_sweep.py, profiles/r04_gemm_x4.md
SWEEP = [
    dict(X8),
    dict(X8, ko="nowait"),
    dict(X8, ko="nowait bar"),
    dict(X8, ko="bar"),
    dict(X8, ko="dma"),
    dict(X8, ko="rd"),
    dict(X8, ko="mfma"),
    dict(X8, ko="mfma rd"),
    dict(X8, ko="mfma dma"),
    dict(X8, ko="mfma nowait"),
]

# register map per geometry (VGPR numbers): fragment bases of k-half 0 / 1, request offsets, fragment-row LDS addresses
GEO = {4: dict(B0=0, B1=32, A0=64, A1=96, VA=128, VB=136, RA0=144, RA1=145, RB0=146, RB1=147),
       8: dict(B0=0, B1=16, A0=32, A1=64, VA=96, VB=100, RA0=104, RA1=105, RB0=106, RB1=107)}   # 108 VGPRs + 128 accumulators: two waves per SIMD


def acc(i, j, nj=8):
    b = (i * nj + j) * 4
    return f"a[{b}:{b + 3}]"


def vreg(base, n):
    return f"v[{base + 4 * n}:{base + 4 * n + 3}]"


def mfma(i, j, ab, bb, nj=8):
    return f"v_mfma_f32_16x16x32_bf16 {acc(i, j, nj)}, {vreg(bb, j)}, {vreg(ab, i)}, {acc(i, j, nj)}"


def reads(half, nj=8):
    """nj + 8 ds_read_b128 of k-half `half`: B fragments first (the first MFMA row needs all of them), then A."""
    out = []
    g = GEO[4 if nj == 8 else 8]
    bb, ab = (g["B0"], g["A0"]) if half == 0 else (g["B1"], g["A1"])
    va, vb = (g["RA0"], g["RB0"]) if half == 0 else (g["RA1"], g["RB1"])
    for j in range(nj):
        out.append(f"ds_read_b128 {vreg(bb, j)}, v{vb} offset:{j * 2048}")
    for i in range(8):
        out.append(f"ds_read_b128 {vreg(ab, i)}, v{va} offset:{i * 2048}")
    return out


def dma_requests(waves=4):
    """The wave's requests of one k-tile: (m0 setup, load) pairs; m0 walks this wave's slots (one 1-KiB slot per `waves`) in the
    A then the B region.  4 waves: 8 + 8 requests, 8 waves: 4 + 4."""
    out = []
    n, stride, g = 32 // waves, 1024 * waves, GEO[waves]
    for i in range(n):
        pre = "s_mov_b32 m0, s46" if i == 0 else f"s_add_u32 m0, m0, {stride}"
        out.append((pre, f"buffer_load_dwordx4 v{g['VA'] + i}, s[36:39], s44 offen lds"))
    for i in range(n):
        pre = f"s_add_u32 m0, s46, {BREG}" if i == 0 else f"s_add_u32 m0, m0, {stride}"
        out.append((pre, f"buffer_load_dwordx4 v{g['VB'] + i}, s[40:43], s44 offen lds"))
    return out


def body(kind, V, ni_act=8):
    """kind: 'steady' (requests k-tile t+2, reads t+1), 'tail2' (no requests, reads t+1), 'last' (neither).
    ni_act < 8: the body of a wave whose A fragments ni_act.. are rows past M -- their MFMAs become `s_nop 0` (one issue cycle each,
    so every hazard spacing of the full body survives) while the reads, requests, waits and barriers stay where they are: the wave
    keeps step with the full-body waves of its workgroup and multiplies nothing that is never stored."""
    rd, dma, mid, end, fine, ko = V["rd"], V["dma"], V["mid"], V["end"], V["fine"], V["ko"].split()
    waves = V["waves"]
    nj = 8 if waves == 4 else 4                             # B fragments per wave
    nslot = 8 * nj * 2                                      # MFMA slots per iteration
    nrd, nreq = 8 + nj, 64 // waves                         # reads per k-half, requests per k-tile (per wave)
    extras = {s: [] for s in range(-1, nslot)}              # instructions that ride behind MFMA slot s
    if "rd" not in ko:
        for n, r in enumerate(reads(1, nj)):
            extras[n * rd].append(r)
    assert (nrd - 1) * rd < mid
    extras[mid].append("s_waitcnt lgkmcnt(0)")              # every wave has read all of cur (also fences the epilogue's LDS use)
    if "bar" not in ko:
        extras[mid].append("s_barrier")
    if fine:
        # A fragment i of k-half 0 (requested at the end of the previous iteration, after the B fragments) is first read by
        # slot nj * i: LDS operations return in order, so it has landed once at most (7 - i) older-still reads plus the k-half 1
        # reads issued since (one per `rd` slots) are outstanding.  lgkmcnt saturates at 15: waiting for more is only earlier.
        for i in range(1, 8):
            issued = min(nrd, (nj * i - 1) // rd + 1)
            extras[nj * i - 1].append(f"s_waitcnt lgkmcnt({min(15, 7 - i + issued)})")
    if kind == "steady" and "dma" not in ko:
        s = mid + 1
        for pre, ld in dma_requests(waves):
            extras[s].append(pre)                           # m0 one slot ahead of the request that reads it
            extras[s + 1].append(ld)
            s += dma
        assert s - dma + 1 < end, "all requests must be issued before the END wait"
    # the k-half 1 address registers move to the other buffer once their reads of cur are issued
    g = GEO[waves]
    extras[nslot // 2] += [f"v_xor_b32 v{r}, 0x10000, v{r}" for r in (g["RA0"], g["RA1"], g["RB0"], g["RB1"])]
    if kind != "last":
        if "nowait" in ko:       # timing probe: never wait for the k-tile that is about to be read (stale LDS, wrong results)
            extras[end].append(f"s_waitcnt vmcnt({2 * nreq})")
        else:
            extras[end].append(f"s_waitcnt vmcnt({nreq})" if (kind == "steady" and "dma" not in ko) else "s_waitcnt vmcnt(0)")
        if "bar" not in ko:
            extras[end].append("s_barrier")
        if "rd" not in ko:
            step = rd if end + 1 + (nrd - 1) * rd <= nslot - 1 else 1
            for n, r in enumerate(reads(0, nj)):
                extras[end + 1 + n * step].append(r)
            assert end + 1 + (nrd - 1) * step <= nslot - 1
    L = []
    slot = 0
    for ab, bb in ((g["A0"], g["B0"]), (g["A1"], g["B1"])):
        for i in range(8):
            for j in (range(nj) if (i % 2 == 0 or "snake" not in V.get("order", "")) else reversed(range(nj))):
                if "mfma32" in ko:
                    # timing probe: the same FLOPs as 32x32x16 instructions (half as many, twice as long); wrong results
                    if slot % 2 == 0:
                        q = (slot // 2) % 16
                        L.append(f"v_mfma_f32_32x32x16_bf16 a[{q * 16}:{q * 16 + 15}], {vreg(bb, j)}, {vreg(ab, i)}, a[{q * 16}:{q * 16 + 15}]")
                elif "mfma" not in ko:
                    L.append(mfma(i, j, ab, bb, nj) if i < ni_act else "s_nop 0")
                L += extras[slot]
                slot += 1
    if kind != "last":
        L.append("s_waitcnt lgkmcnt(7)" if fine else "s_waitcnt lgkmcnt(0)")
    if kind == "steady":
        L.append("s_add_u32 s44, s44, 128")
        L.append("s_xor_b32 s46, s46, 0x10000")
    return L


def program(V, ni_act=8):
    waves = V["waves"]
    nj = 8 if waves == 4 else 4
    nacc = 8 * nj * 4
    nreq = 64 // waves
    P = ["s_nop 4"]
    zero = [f"v_accvgpr_write_b32 a{n}, 0" for n in range(nacc)]
    per = nacc // nreq
    # ---- prologue: k-tile 0 -> buffer 0, k-tile 1 -> buffer 1; the accumulators are zeroed in the shadow of the requests' issue
    for pre, ld in dma_requests(waves):
        P += [pre, "s_nop 0", ld] + [zero.pop() for _ in range(per)]
    P.append("s_add_u32 s44, s44, 128")
    P.append("s_cmp_lt_u32 s45, 2")
    P.append("s_cbranch_scc1 X4_ONE_%=")
    P.append("s_xor_b32 s46, s46, 0x10000")
    for pre, ld in dma_requests(waves):
        P += [pre, "s_nop 0", ld]
    P.append("s_add_u32 s44, s44, 128")
    P.append("s_xor_b32 s46, s46, 0x10000")
    P.append(f"s_waitcnt vmcnt({nreq})")
    P.append("s_branch X4_GO_%=")
    P.append("X4_ONE_%=:")
    P.append("s_waitcnt vmcnt(0)")
    P.append("X4_GO_%=:")
    P.append("s_barrier")
    P += reads(0, nj)
    P.append("s_waitcnt lgkmcnt(0)")
    # ---- loop
    P.append("X4_LOOP_%=:")
    P.append("s_cmp_le_u32 s45, 2")
    P.append("s_cbranch_scc1 X4_TAIL_%=")
    P += body("steady", V, ni_act)
    P.append("s_sub_u32 s45, s45, 1")
    P.append("s_branch X4_LOOP_%=")
    P.append("X4_TAIL_%=:")
    P.append("s_cmp_eq_u32 s45, 1")
    P.append("s_cbranch_scc1 X4_LAST_%=")
    P += body("tail2", V, ni_act)
    P.append("X4_LAST_%=:")
    P += body("last", V, ni_act)
    P.append("s_waitcnt vmcnt(0) lgkmcnt(0)")               # (knock-out variants may leave something in flight)
    P.append("s_nop 7")
    P.append("s_nop 7")
    return P


def block_writer8(h):
    """8-wave form, epilogue block h (64 of the wave's 128 rows x its 64 columns): accumulators (i = h*4 + ii, j) -> slice."""
    return [f"ds_write_b128 %{jj}, {acc(h * 4 + ii, jj, 4)} offset:{ii * 4096}" for ii in range(4) for jj in range(4)]


def block_writer(blk):
    """Epilogue block blk = cb * 2 + h: accumulators (i = h*4 + ii, j = cb*4 + jj) -> the wave's LDS slice, straight from the
    AGPRs; %0..%3 = the lane's slice address for jj = 0..3 (the row's chunk swizzle is in it), ii * 16 rows in the offset."""
    cb, h = blk >> 1, blk & 1
    return [f"ds_write_b128 %{jj}, {acc(h * 4 + ii, cb * 4 + jj)} offset:{ii * 4096}" for ii in range(4) for jj in range(4)]


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    variants = SWEEP if "--sweep" in sys.argv else [DEFAULT]
    with open(os.path.join(here, "gemm_x4_loop.inc"), "w") as f:
        f.write("// GENERATED by gen_gemm_x4.py -- do not edit; the schedule is described there.\n")
        f.write(f"#define X4_NVARIANTS {len(variants)}\n")
        for vi, V in enumerate(variants):
            lines = program(V)
            f.write(f"// variant {vi}: {V}\n#define X4_WAVES_{vi} {V['waves']}\n#define X4_LOOP_{vi} \\\n")
            for ln in lines:
                f.write('  "' + ln + '\\n\\t" \\\n')
            f.write('  ""\n')
            n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
            if vi == 0 and V["waves"] == 8:
                # the shipped loop once more for waves that own at most two row fragments below M (gemm_x4.hip: `x8_part`)
                # ... and for waves with nothing below M / N at all
                for na in (2, 0):
                    f.write(f"#define X4_LOOP_{vi}_P{na} \\\n")
                    for ln in program(V, na):
                        f.write('  "' + ln + '\\n\\t" \\\n')
                    f.write('  ""\n')
            if vi == len(variants) - 1:
                for w, asm in ((4, "X4_ASM"), (8, "X8_ASM")):
                    idx = [i for i, v in enumerate(variants) if v["waves"] == w]

                    def use(i):
                        if w == 8 and i == 0:
                            return (f"if (x8_part == 2) {{ {asm}(X4_LOOP_0_P0); }} else if (x8_part == 1) {{ {asm}(X4_LOOP_0_P2); }} "
                                    f"else {{ {asm}(X4_LOOP_0); }}")
                        return f"{asm}(X4_LOOP_{i});"
                    chain = " else ".join(f"if constexpr (V == {i}) {{ {use(i)} }}" for i in idx) or "(void)0"
                    f.write(f"#define X4_DISPATCH{w}(V) {chain}\n")
                cases = " ".join(f"case {i}: X{v['waves']}_LAUNCH({i}) break;" for i, v in enumerate(variants))
                f.write(f"#define X4_LAUNCH_SWITCH {cases}\n")
            if vi == 0:
                for blk in range(4):
                    f.write(f"#define X4_WR_{blk} \\\n")
                    for ln in block_writer(blk):
                        f.write('  "' + ln + '\\n\\t" \\\n')
                    f.write('  ""\n')
                for h in range(2):
                    f.write(f"#define X8_WR_{h} \\\n")
                    for ln in block_writer8(h):
                        f.write('  "' + ln + '\\n\\t" \\\n')
                    f.write('  ""\n')
            print(f"variant {vi}: {len(lines)} instructions, {n_mfma} MFMAs  {V}")


if __name__ == "__main__":
    main()

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
DEFAULT = dict(rd=2, dma=3, mid=36, end=96, fine=1, ko="")
SWEEP = [
    dict(DEFAULT),
    dict(DEFAULT, fine=0, end=92, mid=42),
    dict(DEFAULT, mid=42),
    dict(DEFAULT, dma=2),
    dict(DEFAULT, fine=0, end=84, mid=34),
    dict(DEFAULT, ko="dma"),
    dict(DEFAULT, ko="rd"),
    dict(DEFAULT, ko="bar"),
    dict(DEFAULT, ko="dma rd"),
    dict(DEFAULT, ko="dma rd bar"),
    dict(DEFAULT, ko="mfma"),
    dict(DEFAULT, ko="dma rd bar mfma32"),
    dict(DEFAULT, ko="mfma32"),
]

B0, B1, A0, A1 = 0, 32, 64, 96


def acc(i, j):
    b = (i * 8 + j) * 4
    return f"a[{b}:{b + 3}]"


def vreg(base, n):
    return f"v[{base + 4 * n}:{base + 4 * n + 3}]"


def mfma(i, j, ab, bb):
    return f"v_mfma_f32_16x16x32_bf16 {acc(i, j)}, {vreg(bb, j)}, {vreg(ab, i)}, {acc(i, j)}"


def reads(half):
    """16 ds_read_b128 of k-half `half`: B fragments first (the first MFMA row needs all eight), then A."""
    out = []
    bb, ab = (B0, A0) if half == 0 else (B1, A1)
    va, vb = (144, 146) if half == 0 else (145, 147)
    for j in range(8):
        out.append(f"ds_read_b128 {vreg(bb, j)}, v{vb} offset:{j * 2048}")
    for i in range(8):
        out.append(f"ds_read_b128 {vreg(ab, i)}, v{va} offset:{i * 2048}")
    return out


def dma_requests():
    """16 requests: (m0 setup, load) pairs; m0 walks this wave's slots (stride 4096) in the A then the B region."""
    out = []
    for i in range(8):
        pre = "s_mov_b32 m0, s46" if i == 0 else "s_add_u32 m0, m0, 0x1000"
        out.append((pre, f"buffer_load_dwordx4 v{128 + i}, s[36:39], s44 offen lds"))
    for i in range(8):
        pre = f"s_add_u32 m0, s46, {BREG}" if i == 0 else "s_add_u32 m0, m0, 0x1000"
        out.append((pre, f"buffer_load_dwordx4 v{136 + i}, s[40:43], s44 offen lds"))
    return out


def body(kind, V):
    """kind: 'steady' (requests k-tile t+2, reads t+1), 'tail2' (no requests, reads t+1), 'last' (neither)."""
    rd, dma, mid, end, fine, ko = V["rd"], V["dma"], V["mid"], V["end"], V["fine"], V["ko"].split()
    extras = {s: [] for s in range(-1, 128)}                # instructions that ride behind MFMA slot s
    if "rd" not in ko:
        for n, r in enumerate(reads(1)):
            extras[n * rd].append(r)
    assert 15 * rd < mid
    extras[mid].append("s_waitcnt lgkmcnt(0)")              # every wave has read all of cur (also fences the epilogue's LDS use)
    if "bar" not in ko:
        extras[mid].append("s_barrier")
    if fine:
        # A fragment i of k-half 0 (requested at the end of the previous iteration, after the eight B fragments) is first read by
        # slot 8 i: LDS operations return in order, so it has landed once at most (7 - i) older-still reads plus the k-half 1
        # reads issued since (one per `rd` slots) are outstanding.  lgkmcnt saturates at 15: waiting for more is only earlier.
        for i in range(1, 8):
            issued = min(16, (8 * i - 1) // rd + 1)
            extras[8 * i - 1].append(f"s_waitcnt lgkmcnt({min(15, 7 - i + issued)})")
    if kind == "steady" and "dma" not in ko:
        s = mid + 1
        for pre, ld in dma_requests():
            extras[s].append(pre)                           # m0 one slot ahead of the request that reads it
            extras[s + 1].append(ld)
            s += dma
        assert s - dma + 1 < end, "all requests must be issued before the END wait"
    # the k-half 1 address registers move to the other buffer once their reads of cur are issued
    extras[64] += [f"v_xor_b32 v{r}, 0x10000, v{r}" for r in (144, 145, 146, 147)]
    if kind != "last":
        extras[end].append("s_waitcnt vmcnt(16)" if (kind == "steady" and "dma" not in ko) else "s_waitcnt vmcnt(0)")
        if "bar" not in ko:
            extras[end].append("s_barrier")
        if "rd" not in ko:
            for n, r in enumerate(reads(0)):
                extras[end + 1 + n * rd].append(r)
        assert end + 1 + 15 * rd <= 127
    L = []
    slot = 0
    for ab, bb in ((A0, B0), (A1, B1)):
        for i in range(8):
            for j in range(8):
                if "mfma32" in ko:
                    # timing probe: the same FLOPs as 32x32x16 instructions (half as many, twice as long); wrong results
                    if slot % 2 == 0:
                        q = (slot // 2) % 16
                        L.append(f"v_mfma_f32_32x32x16_bf16 a[{q * 16}:{q * 16 + 15}], {vreg(bb, j)}, {vreg(ab, i)}, a[{q * 16}:{q * 16 + 15}]")
                elif "mfma" not in ko:
                    L.append(mfma(i, j, ab, bb))
                L += extras[slot]
                slot += 1
    if kind != "last":
        L.append("s_waitcnt lgkmcnt(7)" if fine else "s_waitcnt lgkmcnt(0)")
    if kind == "steady":
        L.append("s_add_u32 s44, s44, 128")
        L.append("s_xor_b32 s46, s46, 0x10000")
    return L


def program(V):
    P = ["s_nop 4"]
    zero = [f"v_accvgpr_write_b32 a{n}, 0" for n in range(256)]
    # ---- prologue: k-tile 0 -> buffer 0, k-tile 1 -> buffer 1; the accumulators are zeroed in the shadow of the requests' issue
    for pre, ld in dma_requests():
        P += [pre, "s_nop 0", ld] + [zero.pop() for _ in range(8)]
    P.append("s_add_u32 s44, s44, 128")
    P.append("s_cmp_lt_u32 s45, 2")
    P.append("s_cbranch_scc1 X4_ONE_%=")
    P.append("s_xor_b32 s46, s46, 0x10000")
    z2 = list(zero)
    for pre, ld in dma_requests():
        P += [pre, "s_nop 0", ld] + [z2.pop() for _ in range(8)]
    P.append("s_add_u32 s44, s44, 128")
    P.append("s_xor_b32 s46, s46, 0x10000")
    P.append("s_waitcnt vmcnt(16)")
    P.append("s_branch X4_GO_%=")
    P.append("X4_ONE_%=:")
    P += zero
    P.append("s_waitcnt vmcnt(0)")
    P.append("X4_GO_%=:")
    P.append("s_barrier")
    P += reads(0)
    P.append("s_waitcnt lgkmcnt(0)")
    # ---- loop
    P.append("X4_LOOP_%=:")
    P.append("s_cmp_le_u32 s45, 2")
    P.append("s_cbranch_scc1 X4_TAIL_%=")
    P += body("steady", V)
    P.append("s_sub_u32 s45, s45, 1")
    P.append("s_branch X4_LOOP_%=")
    P.append("X4_TAIL_%=:")
    P.append("s_cmp_eq_u32 s45, 1")
    P.append("s_cbranch_scc1 X4_LAST_%=")
    P += body("tail2", V)
    P.append("X4_LAST_%=:")
    P += body("last", V)
    P.append("s_waitcnt vmcnt(0) lgkmcnt(0)")               # (knock-out variants may leave something in flight)
    P.append("s_nop 7")
    P.append("s_nop 7")
    return P


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
            f.write(f"// variant {vi}: {V}\n#define X4_LOOP_{vi} \\\n")
            for ln in lines:
                f.write('  "' + ln + '\\n\\t" \\\n')
            f.write('  ""\n')
            n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
            if vi == 0:
                for blk in range(4):
                    f.write(f"#define X4_WR_{blk} \\\n")
                    for ln in block_writer(blk):
                        f.write('  "' + ln + '\\n\\t" \\\n')
                    f.write('  ""\n')
            print(f"variant {vi}: {len(lines)} instructions, {n_mfma} MFMAs  {V}")


if __name__ == "__main__":
    main()

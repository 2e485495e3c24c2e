// K1, four-wave form of the 256 x 256 tile (round 4): BK = 64, two 64-KiB LDS buffers, ONE wave per SIMD owning a 128 x 128
// sub-tile (8 x 8 fragments of mfma_f32_16x16x32_bf16, 256 accumulator registers in the AGPR half of the unified file), the
// whole K loop one hand-scheduled instruction stream (gemm_x4_loop.inc, written by gen_gemm_x4.py: MFMA slots with the
// fragment reads, the LDS-DMA requests, two counted waits and two barriers per 128 MFMAs pinned behind them).
//
// Why (profiles/r04_gemm_x4.md): the eight-wave kernel (gemm_256.hip) reads 96 KiB of fragments out of LDS per 32-KiB k-tile
// and lands at 0.75 us per 32-deep k-tile whatever is done to its schedule (rounds 2-3); with 128 x 128 per wave every
// fragment is read twice instead of four / two times (64 KiB), a 64-deep k-tile makes every LDS-DMA request eight whole
// 128-byte lines, and requests are `buffer_load_dwordx4 ... lds` with one address VGPR each and a shared scalar k offset
// (no per-request 64-bit address arithmetic).  Round 3's four-wave attempt left the order to the compiler and lost; here the
// order is written down.
//
// LDS image of a buffer: [A: 256 rows x 128 B | B: 256 rows x 128 B]; row r keeps its eight 16-byte chunks at physical
// position c ^ ((r >> 1) & 7): a ds_read_b128 lane group (16 lanes: rows {0-3, 12-15} at chunk c and rows {4-11} at chunk
// c ^ 1, MI355X_MICROARCH.md LDS table) then touches 16 distinct 16-byte slots of the 256-byte bank row -- conflict-free --
// and the LDS-DMA, which writes lane-linearly, gets the same permutation on its per-lane SOURCE chunk (whole lines either way).
// Request i of wave w covers rows (w + 4 i) * 8 .. + 7: the swizzle term ((row >> 1) & 7) = 4 (w & 1) + (lane >> 4) does not
// depend on i, and m0 advances by 4096 per request.
//
// Same k order per output element as gemm_256.hip / gemm.hip (32-deep MFMA steps in ascending k): results are bit-identical.
#include "common.h"
#include <cstdlib>

#define X4_BM 256
#define X4_BN 256
#define X4_BUF 0x10000
#define X4_BREG 0x8000

#define MH_GEMM_OUT_F32 1
#define MH_GEMM_GELU 2
// launcher-internal flag bits (never part of the ABI's flags): debug hooks of the sweep / probe tools and the padding policy
#define X4_F_SAME_PANEL 0x40000000   // every workgroup stages tile (0, 0)'s panels (all-L2-hit timing probe; sweep builds)
#define X4_F_ZERO_PAD 0x20000000     // rows past M / N read as zeros (default on; option gemm_zero_pad)
#define X4_F_CLOCK_PROBE 0x10000000  // aux = int64 stamps of workgroups 0, 64, .. (mhdbg_set_gemm_x4_clock_probe)
#define X4_F_SKIP_PAD 0x04000000     // waves whose rows lie (almost) all past M skip those fragments' MFMAs (option gemm_skip_pad)
#define X4_F_SPLIT_XCD 0x02000000    // K-split launches: an XCD owns one split of a band of tile columns (option gemm_split_xcd)
#define X4_F_NO_STORES 0x08000000    // read-out without its global stores (mhdbg_set_gemm_x4_no_stores: timing only, wrong results)
#define MH_GEMM_SWIGLU_FWD 16
#define MH_GEMM_SWIGLU_BWD 32

typedef __attribute__((ext_vector_type(32))) float float32_t;
typedef __attribute__((ext_vector_type(8))) int int8v_t;
typedef __attribute__((ext_vector_type(4))) int int4v_t;

#define X4_CLOB8(b) "v" #b "0", "v" #b "1", "v" #b "2", "v" #b "3", "v" #b "4", "v" #b "5", "v" #b "6", "v" #b "7", "v" #b "8", "v" #b "9"

#include "gemm_x4_loop.inc"

// the K loop: outputs = the 8 x 32 accumulator registers; in/out = the registers the loop advances; inputs pinned where the
// instruction stream names them (gen_gemm_x4.py: register map)
#define X4_ASM(LOOP)                                                                                                            \
  asm volatile(LOOP                                                                                                             \
               : "={a[0:31]}"(c[0]), "={a[32:63]}"(c[1]), "={a[64:95]}"(c[2]), "={a[96:127]}"(c[3]), "={a[128:159]}"(c[4]),     \
                 "={a[160:191]}"(c[5]), "={a[192:223]}"(c[6]), "={a[224:255]}"(c[7]), "+{v144}"(ra0), "+{v145}"(ra1),           \
                 "+{v146}"(rb0), "+{v147}"(rb1), "+{s44}"(koff), "+{s45}"(cnt), "+{s46}"(wr)                                    \
               : "{v[128:135]}"(voa), "{v[136:143]}"(vob), "{s[36:39]}"(sa), "{s[40:43]}"(sb)                                   \
               /* m0 is written inside (24 times) but cannot be named here: it is a RESERVED register to the compiler ('inline asm
                  clobber list contains reserved registers: m0'), which never keeps a value live in it across a statement and
                  sets it up again in front of each of its own uses (LDS-DMA, movrel, sendmsg) -- ADVICE r4 */ \
               : "memory", "scc", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", X4_CLOB8(1), X4_CLOB8(2),         \
                 X4_CLOB8(3), X4_CLOB8(4), X4_CLOB8(5), X4_CLOB8(6), X4_CLOB8(7), X4_CLOB8(8), X4_CLOB8(9), X4_CLOB8(10),       \
                 X4_CLOB8(11), "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127")

// Read-out of one 64 x 64 block of a wave's tile from its LDS slice `ep` ([64 rows][16 chunks of 16 B], chunk ^= row & 15, fp32,
// un-scaled accumulators) to global memory, shared by the four- and eight-wave forms; the same expressions as gemm_256.hip's
// epilogue (alpha applied here, on the way out of LDS).  mbase: global row of the block's row 0; wc: the block's 64-column
// index inside the 256-column tile.  SwiGLU forward pairs the wave holding gate columns (slice g_slice) with the one holding
// the matching up columns (g_slice + u_step); each turns rows pair_half * 32 .. + 31 into act columns (n0 / 2) + act_cb * 64 ..
__device__ __forceinline__ void x_readout(char* smem, char* ep, int lane, int mbase, int wc, int n0, int g_slice, int u_step,
                                          int pair_half, int act_cb, void* Cv, const float* bias, const float* res, int M, int N,
                                          int ldc, int ldr, int flags, float alpha, void* aux, int ldaux) {
  const bool out_f32 = flags & MH_GEMM_OUT_F32;
  const bool do_gelu = flags & MH_GEMM_GELU;
  const bool sw_fwd = flags & MH_GEMM_SWIGLU_FWD, sw_bwd = flags & MH_GEMM_SWIGLU_BWD;
  const int er = lane >> 4, ec = lane & 15;
  const int ncol = n0 + wc * 64 + ec * 4;
  if (sw_bwd) {
    const int r8 = lane >> 3, c8 = lane & 7;
    const int nc = n0 + wc * 64 + c8 * 8;
    const long gcol = (long)(nc >> 7) * 256 + (nc & 127);
    const bf16_t* gu_in = reinterpret_cast<const bf16_t*>(aux);
    bf16_t* dgu = reinterpret_cast<bf16_t*>(Cv);
    // the gate / up values of four rows at a time, loaded before they are used (4 rows x 2 x 16 B per lane: the K loop's
    // registers are free here): the loop waits for one memory round trip per four rows instead of one per row
    for (int ph = 0; ph < 2; ++ph) {
    short8_t gq[4], uq[4];
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int m = mbase + (ph * 4 + pp) * 8 + r8;
      gq[pp] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
      uq[pp] = gq[pp];
      if (m < M && nc < N) {
        gq[pp] = *reinterpret_cast<const short8_t*>(gu_in + (size_t)m * ldaux + gcol);
        uq[pp] = *reinterpret_cast<const short8_t*>(gu_in + (size_t)m * ldaux + gcol + 128);
      }
    }
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int p = ph * 4 + pp;
      const int row = p * 8 + r8;
      const int m = mbase + row;
      const float4_t va = *reinterpret_cast<const float4_t*>(ep + row * 256 + (((2 * c8) ^ (row & 15)) << 4));
      const float4_t vb = *reinterpret_cast<const float4_t*>(ep + row * 256 + (((2 * c8 + 1) ^ (row & 15)) << 4));
      if (m >= M || nc >= N) continue;
      const float v[8] = {va[0] * alpha, va[1] * alpha, va[2] * alpha, va[3] * alpha,
                          vb[0] * alpha, vb[1] * alpha, vb[2] * alpha, vb[3] * alpha};
      const short8_t g8 = gq[pp];
      const short8_t u8 = uq[pp];
      short8_t og, ou;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gv = bf2f((bf16_t)g8[e]), uv = bf2f((bf16_t)u8[e]), dv = bf2f(f2bf(v[e]));
        const float sg = 1.f / (1.f + __expf(-gv));
        const float silu = gv * sg;
        og[e] = (short)f2bf(dv * uv * (sg + silu * (1.f - sg)));
        ou[e] = (short)f2bf(dv * silu);
      }
      *reinterpret_cast<short8_t*>(dgu + (size_t)m * ldc + gcol) = og;
      *reinterpret_cast<short8_t*>(dgu + (size_t)m * ldc + gcol + 128) = ou;
    }
    }
    return;
  }
  if (sw_fwd) {
    // wave (wm, 0) holds gate columns, (wm, 1) the matching up columns of this 64 x 64 block: each of the pair turns 32
    // of the 64 rows into act = silu(g) * u
    __syncthreads();
    const int r8 = lane >> 3, c8 = lane & 7;
    const char* sg_ = smem + g_slice * 16384;
    const char* su_ = smem + (g_slice + u_step) * 16384;
    bf16_t* act = reinterpret_cast<bf16_t*>(aux);
    const int ac = (n0 >> 1) + act_cb * 64 + c8 * 8;
#pragma unroll 4
    for (int p = 0; p < 4; ++p) {
      const int row = pair_half * 32 + p * 8 + r8;
      const int m = mbase + row;
      const int o0 = row * 256 + (((2 * c8) ^ (row & 15)) << 4), o1 = row * 256 + (((2 * c8 + 1) ^ (row & 15)) << 4);
      const float4_t ga = *reinterpret_cast<const float4_t*>(sg_ + o0), gb = *reinterpret_cast<const float4_t*>(sg_ + o1);
      const float4_t ua = *reinterpret_cast<const float4_t*>(su_ + o0), ub = *reinterpret_cast<const float4_t*>(su_ + o1);
      if (m >= M || ac >= (N >> 1)) continue;
      const float gg[8] = {ga[0], ga[1], ga[2], ga[3], gb[0], gb[1], gb[2], gb[3]};
      const float uu[8] = {ua[0], ua[1], ua[2], ua[3], ub[0], ub[1], ub[2], ub[3]};
      short8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gv = bf2f(f2bf(gg[e] * alpha)), uv = bf2f(f2bf(uu[e] * alpha));
        o[e] = (short)f2bf(gv / (1.f + __expf(-gv)) * uv);
      }
      *reinterpret_cast<short8_t*>(act + (size_t)m * ldaux + ac) = o;
    }
  }
  if (!out_f32 && (ldc & 7) == 0) {
    const int r8 = lane >> 3, c8 = lane & 7;
    const int nc = n0 + wc * 64 + c8 * 8;
    // Block whose 64 columns lie inside N (rows past M are skipped row by row): the options are tested once, the bias of the lane's eight columns is
    // loaded once, the row address advances by a constant -- per 8 rows two LDS reads, the epilogue arithmetic, four packed
    // conversions and one full-line store.  The generic loop below (edges) re-tests everything and redoes a 64-bit address per
    // row; with 128 outputs per lane and two waves per SIMD the read-out is instruction-bound (7.3 us of a workgroup's time with
    // the integer bf16 rounding, 2.5 us now: profiles/r04_gemm_x4.md).  Same expressions, same bits.
    if (!sw_fwd && (mbase >= M || n0 + wc * 64 >= N)) return;        // the block lies in the padding: nothing to store
    if (!sw_fwd && n0 + wc * 64 + 63 < N && !(flags & X4_F_NO_STORES)) {
      const int mrows = M - mbase - r8;                               // row p * 8 + r8 of the block is stored iff p * 8 < mrows
      bf16_t* dst = reinterpret_cast<bf16_t*>(Cv) + (size_t)(mbase + r8) * ldc + nc;
      const size_t step = (size_t)8 * ldc;
      const char* e0 = ep + r8 * 256;
      const bool plain = !bias && !do_gelu && !res && alpha == 1.0f;
      if (plain) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int sw = ((p & 1) * 8 + r8) & 15;          // row & 15 for row = p * 8 + r8
          const float4_t va = *reinterpret_cast<const float4_t*>(e0 + p * 2048 + (((2 * c8) ^ sw) << 4));
          const float4_t vb = *reinterpret_cast<const float4_t*>(e0 + p * 2048 + (((2 * c8 + 1) ^ sw) << 4));
          uint4 pk;
          pk.x = pack_bf2(va[0], va[1]);
          pk.y = pack_bf2(va[2], va[3]);
          pk.z = pack_bf2(vb[0], vb[1]);
          pk.w = pack_bf2(vb[2], vb[3]);
          if (p * 8 < mrows) *reinterpret_cast<uint4*>(dst) = pk;
          dst += step;
        }
        return;
      }
      float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (bias) {
        const float4_t b0 = *reinterpret_cast<const float4_t*>(bias + nc);
        const float4_t b1 = *reinterpret_cast<const float4_t*>(bias + nc + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bb[e] = b0[e]; bb[4 + e] = b1[e]; }
      }
      const float* rp = res ? res + (size_t)(mbase + r8) * ldr + nc : nullptr;
      const size_t rstep = (size_t)8 * ldr;
#pragma unroll 2
      for (int p = 0; p < 8; ++p) {
        const int sw = ((p & 1) * 8 + r8) & 15;
        const float4_t va = *reinterpret_cast<const float4_t*>(e0 + p * 2048 + (((2 * c8) ^ sw) << 4));
        const float4_t vb = *reinterpret_cast<const float4_t*>(e0 + p * 2048 + (((2 * c8 + 1) ^ sw) << 4));
        float v[8] = {va[0] * alpha, va[1] * alpha, va[2] * alpha, va[3] * alpha,
                      vb[0] * alpha, vb[1] * alpha, vb[2] * alpha, vb[3] * alpha};
        if (bias) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bb[e];
        }
        if (do_gelu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
        }
        if (rp) {
          if (p * 8 < mrows) {
            const float4_t q0 = *reinterpret_cast<const float4_t*>(rp);
            const float4_t q1 = *reinterpret_cast<const float4_t*>(rp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += q0[e]; v[4 + e] += q1[e]; }
          }
          rp += rstep;
        }
        uint4 pk;
        pk.x = pack_bf2(v[0], v[1]);
        pk.y = pack_bf2(v[2], v[3]);
        pk.z = pack_bf2(v[4], v[5]);
        pk.w = pack_bf2(v[6], v[7]);
        if (p * 8 < mrows) *reinterpret_cast<uint4*>(dst) = pk;
        dst += step;
      }
      return;
    }
#pragma unroll 4
    for (int p = 0; p < 8; ++p) {
      const int row = p * 8 + r8;
      const int m = mbase + row;
      const float4_t va = *reinterpret_cast<const float4_t*>(ep + row * 256 + (((2 * c8) ^ (row & 15)) << 4));
      const float4_t vb = *reinterpret_cast<const float4_t*>(ep + row * 256 + (((2 * c8 + 1) ^ (row & 15)) << 4));
      if (m >= M || nc >= N) continue;
      float v[8] = {va[0] * alpha, va[1] * alpha, va[2] * alpha, va[3] * alpha,
                    vb[0] * alpha, vb[1] * alpha, vb[2] * alpha, vb[3] * alpha};
      if (nc + 7 < N) {
        if (bias) {
          const float4_t b0 = *reinterpret_cast<const float4_t*>(bias + nc);
          const float4_t b1 = *reinterpret_cast<const float4_t*>(bias + nc + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
        }
        if (do_gelu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
        }
        if (res) {
          const float4_t q0 = *reinterpret_cast<const float4_t*>(res + (size_t)m * ldr + nc);
          const float4_t q1 = *reinterpret_cast<const float4_t*>(res + (size_t)m * ldr + nc + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] += q0[e]; v[4 + e] += q1[e]; }
        }
        uint4 pk;
        pk.x = pack_bf2(v[0], v[1]);
        pk.y = pack_bf2(v[2], v[3]);
        pk.z = pack_bf2(v[4], v[5]);
        pk.w = pack_bf2(v[6], v[7]);
        if (flags & X4_F_NO_STORES) { if (pk.x == 0x12345678u) reinterpret_cast<unsigned*>(Cv)[0] = pk.y ^ pk.z ^ pk.w; continue; }   // timing probe (X4_F_NO_STORES)
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(Cv) + (size_t)m * ldc + nc) = pk;
      } else {
        for (int e = 0; e < 8 && nc + e < N; ++e) {
          float x = v[e];
          if (bias) x += bias[nc + e];
          if (do_gelu) x = gelu_erf(x);
          if (res) x += res[(size_t)m * ldr + nc + e];
          reinterpret_cast<bf16_t*>(Cv)[(size_t)m * ldc + nc + e] = f2bf(x);
        }
      }
    }
    if (sw_fwd) __syncthreads();                     // the partner has read this slice before the next block overwrites it
    return;
  }
#pragma unroll 4
  for (int p = 0; p < 16; ++p) {
    const int row = p * 4 + er;
    const int m = mbase + row;
    const float4_t v4 = *reinterpret_cast<const float4_t*>(ep + row * 256 + ((ec ^ (row & 15)) << 4));
    if (m >= M || ncol >= N) continue;
    float v[4] = {v4[0] * alpha, v4[1] * alpha, v4[2] * alpha, v4[3] * alpha};
    if (ncol + 3 < N) {
      if (bias) {
        const float4_t b4 = *reinterpret_cast<const float4_t*>(bias + ncol);
        v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3];
      }
      if (do_gelu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
      }
      if (res) {
        const float4_t r4 = *reinterpret_cast<const float4_t*>(res + (size_t)m * ldr + ncol);
        v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3];
      }
      if (out_f32) {
        *reinterpret_cast<float4_t*>(reinterpret_cast<float*>(Cv) + (size_t)m * ldc + ncol) =
            (float4_t){v[0], v[1], v[2], v[3]};
      } else {
        uint2 pk;
        pk.x = pack_bf2(v[0], v[1]);
        pk.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + (size_t)m * ldc + ncol) = pk;
      }
    } else {
      for (int e = 0; e < 4 && ncol + e < N; ++e) {
        float x = v[e];
        if (bias) x += bias[ncol + e];
        if (do_gelu) x = gelu_erf(x);
        if (res) x += res[(size_t)m * ldr + ncol + e];
        if (out_f32) reinterpret_cast<float*>(Cv)[(size_t)m * ldc + ncol + e] = x;
        else reinterpret_cast<bf16_t*>(Cv)[(size_t)m * ldc + ncol + e] = f2bf(x);
      }
    }
  }
}

template <int V>
__global__ __launch_bounds__(256) void gemm_x4_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, void* Cv,
                                                      const float* __restrict__ bias, const float* res, int M, int N, int K,
                                                      int lda, int ldb, int ldc, int ldr, int flags, float alpha, int tiles_m,
                                                      int kt_per_split, long split_stride, void* aux, int ldaux) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][A 32K | B 32K]; reused by the epilogue
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lg = lane >> 4;

  // tile order of gemm_256.hip: an XCD's concurrent workgroups walk 8 tile-rows before the next tile-column
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tiles_n = nwg / tiles_m;
  const int per_group = 8 * tiles_n;
  const int first_m = (lid / per_group) * 8;
  const int gsz = (tiles_m - first_m) < 8 ? (tiles_m - first_m) : 8;
  const int tm = first_m + (lid % per_group) % gsz, tn = (lid % per_group) / gsz;
  const int m0 = tm * X4_BM, n0 = tn * X4_BN;

  const int nt_all = K >> 6;
  const int kt0 = blockIdx.y * kt_per_split;
  const int nt = (nt_all - kt0) < kt_per_split ? (nt_all - kt0) : kt_per_split;
  if (gridDim.y > 1)
    Cv = (flags & MH_GEMM_OUT_F32) ? (void*)(reinterpret_cast<float*>(Cv) + blockIdx.y * split_stride)
                                   : (void*)(reinterpret_cast<bf16_t*>(Cv) + blockIdx.y * split_stride);

  float32_t c[8];
  {
    // per-lane byte offsets of this wave's requests: request i stages rows (wave + 4 i) * 8 + (lane >> 3), the lane brings
    // the logical chunk that belongs at physical position lane & 7 (rows past the edge re-read the last row: never stored)
    int8v_t voa, vob;
    // Rows past M / N: one row past the end lies outside the descriptor's range (num_records = rows * ld * 2; the scalar k offset
    // is not part of the range check) and reads as ZERO instead of a copy of the last row.  Same results (those rows are never
    // stored), but the MFMAs of the padding multiply zeros: the chip is power-limited under this loop, and 96 of the 1280 padded
    // LLaMA rows doing no switching is worth 1.5-2.3 % on the M = 1184 launches (profiles/r04_gemm_x4.md).
    const int zp = (flags & X4_F_ZERO_PAD) ? 1 : 0;
    const int swz = (4 * (wave & 1) + (lane >> 4)) & 7;
    const int ch = ((lane & 7) ^ swz) << 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (wave + 4 * i) * 8 + (lane >> 3);
      int ra = m0 + row, rb = n0 + row;
      ra = ra < M ? ra : M - 1 + zp;               // zp: one row past the end = outside the descriptor's range: reads as zero
      rb = rb < N ? rb : N - 1 + zp;
      voa[i] = ra * lda * 2 + ch;
      vob[i] = rb * ldb * 2 + ch;
    }
    const int sw = (lr >> 1) & 7;
    const unsigned sbase = (unsigned)(uintptr_t)smem;
    unsigned ra0 = sbase + (wm * 128 + lr) * 128 + ((lg ^ sw) << 4);
    unsigned ra1 = sbase + (wm * 128 + lr) * 128 + (((4 + lg) ^ sw) << 4);
    unsigned rb0 = sbase + X4_BREG + (wn * 128 + lr) * 128 + ((lg ^ sw) << 4);
    unsigned rb1 = sbase + X4_BREG + (wn * 128 + lr) * 128 + (((4 + lg) ^ sw) << 4);
    int4v_t sa, sb;
    sa[0] = (int)(unsigned)(uintptr_t)A;
    sa[1] = (int)((unsigned)((uintptr_t)A >> 32) & 0xffffu);
    sa[2] = zp ? (int)((unsigned)M * (unsigned)lda * 2u) : -1;
    sa[3] = 0x00020000;
    sb[0] = (int)(unsigned)(uintptr_t)B;
    sb[1] = (int)((unsigned)((uintptr_t)B >> 32) & 0xffffu);
    sb[2] = zp ? (int)((unsigned)N * (unsigned)ldb * 2u) : -1;
    sb[3] = 0x00020000;
    unsigned koff = (unsigned)kt0 * 128u, cnt = (unsigned)nt, wr = sbase + wave * 1024;
    X4_DISPATCH4(V);
  }

  // ---- epilogue: the 128 x 128 wave tile leaves in four 64 x 64 blocks (column half cb, row half h), each transposed
  // through the wave's private 16-KiB LDS slice as gemm_256.hip does it (whole-line stores, bias / GELU / residual / SwiGLU
  // forms on the way out).  The accumulators go to LDS straight from the AGPRs (ds_write_b128 takes accumulator data: 16
  // instructions per block, X4_WR_n in gemm_x4_loop.inc) and the read-out is ONE runtime loop over the blocks -- the first
  // version indexed the accumulator vectors from C++, which unrolled everything four times (24 k instructions, 256
  // v_accvgpr_read, SGPR spills: 21 us of fixed cost per launch against the eight-wave kernel's 10.6).  alpha is applied on
  // the way out of LDS (the same fp32 product).  The loop's last barrier sits after every wave's last fragment read and no
  // LDS-DMA is in flight (the tail iterations request nothing), so the buffers are free.
  char* ep = smem + wave * 16384;                      // [64 rows][16 chunks of 16 B], chunk ^= row & 15
  // fragment (ii, jj) of a block: row ii * 16 + lr, 16-byte chunk (jj * 4 + lg) ^ lr -> one address per jj, ii in the offset
  unsigned wa0, wa1, wa2, wa3;
  {
    const unsigned b0 = (unsigned)(uintptr_t)ep + lr * 256 + ((lg ^ (lr & 3)) << 4);
    wa0 = b0 + ((0 ^ (lr >> 2)) << 6);
    wa1 = b0 + ((1 ^ (lr >> 2)) << 6);
    wa2 = b0 + ((2 ^ (lr >> 2)) << 6);
    wa3 = b0 + ((3 ^ (lr >> 2)) << 6);
  }
#define X4_WRITE(TXT)                                                                                                        \
  asm volatile(TXT ::"v"(wa0), "v"(wa1), "v"(wa2), "v"(wa3), "{a[0:31]}"(c[0]), "{a[32:63]}"(c[1]), "{a[64:95]}"(c[2]),      \
               "{a[96:127]}"(c[3]), "{a[128:159]}"(c[4]), "{a[160:191]}"(c[5]), "{a[192:223]}"(c[6]), "{a[224:255]}"(c[7])   \
               : "memory")
#pragma unroll 1
  for (int blk = 0; blk < 4; ++blk) {
    const int cb = blk >> 1, h = blk & 1;
    if (blk == 0) X4_WRITE(X4_WR_0);
    else if (blk == 1) X4_WRITE(X4_WR_1);
    else if (blk == 2) X4_WRITE(X4_WR_2);
    else X4_WRITE(X4_WR_3);
    // 64-column block wn * 2 + cb of the tile, as gemm_256.hip numbers them; SwiGLU: wave (wm, 0) holds gate, (wm, 1) up columns
    x_readout(smem, ep, lane, m0 + wm * 128 + h * 64, wn * 2 + cb, n0, wm * 2, 1, wn, cb, Cv, bias, res, M, N, ldc, ldr, flags,
              alpha, aux, ldaux);
  }
}

// ---- eight-wave form: 2 x 4 waves, 128 x 64 per wave (128 accumulators, 108 VGPRs: two waves per SIMD), the same buffers,
// LDS image, request pattern (each wave issues 4 + 4 of the 64 requests of a k-tile) and loop structure at half the slots per
// wave.  What one wave cannot hide under its own MFMAs -- the issue time of its LDS-DMA requests (~60 cycles each) and
// fragment reads -- the SIMD's other wave fills (profiles/r04_gemm_x4.md: knock-outs of the four-wave loop).
#define X8_ASM(LOOP)                                                                                                            \
  asm volatile(LOOP                                                                                                             \
               : "={a[0:31]}"(c[0]), "={a[32:63]}"(c[1]), "={a[64:95]}"(c[2]), "={a[96:127]}"(c[3]), "+{v104}"(ra0),            \
                 "+{v105}"(ra1), "+{v106}"(rb0), "+{v107}"(rb1), "+{s44}"(koff), "+{s45}"(cnt), "+{s46}"(wr)                    \
               : "{v[96:99]}"(voa), "{v[100:103]}"(vob), "{s[36:39]}"(sa), "{s[40:43]}"(sb)                                     \
               : "memory", "scc", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", X4_CLOB8(1), X4_CLOB8(2),         \
                 X4_CLOB8(3), X4_CLOB8(4), X4_CLOB8(5), X4_CLOB8(6), X4_CLOB8(7), X4_CLOB8(8), "v90", "v91", "v92", "v93",      \
                 "v94", "v95")

template <int V>
__global__ __launch_bounds__(512) void gemm_x8_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, void* Cv,
                                                      const float* __restrict__ bias, const float* res, int M, int N, int K,
                                                      int lda, int ldb, int ldc, int ldr, int flags, float alpha, int tiles_m,
                                                      int kt_per_split, long split_stride, void* aux, int ldaux) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][A 32K | B 32K]; reused by the epilogue
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // debug (mhdbg_set_gemm_x4_clock_probe): shader-clock and 100-MHz stamps around this workgroup -> the clock the loop really ran at
  const bool probe = flags & X4_F_CLOCK_PROBE;
  long long pc0 = 0, pr0 = 0, pr1 = 0, pr2 = 0;
  if (probe) { pc0 = (long long)__builtin_amdgcn_s_memtime(); pr0 = (long long)__builtin_amdgcn_s_memrealtime(); }
  const int lr = lane & 15, lg = lane >> 4;

  // Which (tile, K split) this workgroup owns.  Workgroups are handed to the eight XCDs round-robin in launch order (x fastest,
  // then y), each XCD has its own L2.  Unsplit launches: an XCD's workgroups are one contiguous run of the grouped tile order
  // (below).  K-split launches (round 6, option gemm_split_xcd): the run is taken from the list ordered (split, tile column,
  // tile row), so an XCD works on ONE split (two at a seam) of a band of tile columns -- it streams the A panel of that K range
  // once and each of its B panels once.  Before, every XCD ran all splits of a 5 x 2 tile block and each of the eight L2s read
  // the whole A matrix (profiles/r05_gemm256_traffic.json: the 80 x 3 grids 307 MB against 130 MB algorithmic).  Same
  // arithmetic per output element and per slab: the results do not change.
  const int nwg = gridDim.x;
  int bid = blockIdx.x, ksplit = blockIdx.y;
  if (gridDim.y > 1 && (flags & X4_F_SPLIT_XCD)) {
    const int total = nwg * gridDim.y, L = blockIdx.y * nwg + blockIdx.x;
    const int q8 = total >> 3, r8 = total & 7, x8 = L & 7;
    const int id = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + (L >> 3);
    ksplit = id / nwg;
    bid = -1 - (id - ksplit * nwg);                // a logical tile id, already in (column, row) order
  }
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int lid = bid < 0 ? -1 - bid : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tiles_n = nwg / tiles_m;
  const int per_group = 8 * tiles_n;
  const int first_m = (lid / per_group) * 8;
  const int gsz = (tiles_m - first_m) < 8 ? (tiles_m - first_m) : 8;
  const int tm = first_m + (lid % per_group) % gsz, tn = (lid % per_group) / gsz;
  const int m0 = tm * X4_BM, n0 = tn * X4_BN;

  const int nt_all = K >> 6;
  const int kt0 = ksplit * kt_per_split;
  const int nt = (nt_all - kt0) < kt_per_split ? (nt_all - kt0) : kt_per_split;
  if (gridDim.y > 1)
    Cv = (flags & MH_GEMM_OUT_F32) ? (void*)(reinterpret_cast<float*>(Cv) + ksplit * split_stride)
                                   : (void*)(reinterpret_cast<bf16_t*>(Cv) + ksplit * split_stride);

  float32_t c[4];
  {
    // request i of wave w stages rows (w + 8 i) * 8 + (lane >> 3); the swizzle term is again independent of i
    int4v_t voa, vob;
    // Rows past M / N: one row past the end lies outside the descriptor's range (num_records = rows * ld * 2; the scalar k offset
    // is not part of the range check) and reads as ZERO instead of a copy of the last row.  Same results (those rows are never
    // stored), but the MFMAs of the padding multiply zeros: the chip is power-limited under this loop, and 96 of the 1280 padded
    // LLaMA rows doing no switching is worth 1.5-2.3 % on the M = 1184 launches (profiles/r04_gemm_x4.md).
    const int zp = (flags & X4_F_ZERO_PAD) ? 1 : 0;
    const int swz = (4 * (wave & 1) + (lane >> 4)) & 7;
    const int ch = ((lane & 7) ^ swz) << 4;
#if X4_NVARIANTS > 1
    const int m0l = (flags & X4_F_SAME_PANEL) ? 0 : m0, n0l = (flags & X4_F_SAME_PANEL) ? 0 : n0;   // sweep builds: same-panel timing probe
#else
    const int m0l = m0, n0l = n0;
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave + 8 * i) * 8 + (lane >> 3);
      int ra = m0l + row, rb = n0l + row;
      ra = ra < M ? ra : M - 1 + zp;               // zp: one row past the end = outside the descriptor's range: reads as zero
      rb = rb < N ? rb : N - 1 + zp;
      voa[i] = ra * lda * 2 + ch;
      vob[i] = rb * ldb * 2 + ch;
    }
    const int sw = (lr >> 1) & 7;
    const unsigned sbase = (unsigned)(uintptr_t)smem;
    unsigned ra0 = sbase + (wm * 128 + lr) * 128 + ((lg ^ sw) << 4);
    unsigned ra1 = sbase + (wm * 128 + lr) * 128 + (((4 + lg) ^ sw) << 4);
    unsigned rb0 = sbase + X4_BREG + (wn * 64 + lr) * 128 + ((lg ^ sw) << 4);
    unsigned rb1 = sbase + X4_BREG + (wn * 64 + lr) * 128 + (((4 + lg) ^ sw) << 4);
    int4v_t sa, sb;
    sa[0] = (int)(unsigned)(uintptr_t)A;
    sa[1] = (int)((unsigned)((uintptr_t)A >> 32) & 0xffffu);
    sa[2] = zp ? (int)((unsigned)M * (unsigned)lda * 2u) : -1;
    sa[3] = 0x00020000;
    sb[0] = (int)(unsigned)(uintptr_t)B;
    sb[1] = (int)((unsigned)((uintptr_t)B >> 32) & 0xffffu);
    sb[2] = zp ? (int)((unsigned)N * (unsigned)ldb * 2u) : -1;
    sb[3] = 0x00020000;
    unsigned koff = (unsigned)kt0 * 128u, cnt = (unsigned)nt, wr = sbase + wave * 1024;
    if (probe) pr1 = (long long)__builtin_amdgcn_s_memrealtime();
    // This wave's 128-row slab reaches at most two fragments (32 rows) below M -- the second wave row of the last row tile of the
    // M = 1184 launches (160 of 256 rows), both wave rows of the ViT's (8 of 256): it runs the loop WITHOUT the MFMAs of fragments
    // 2..7 (gen_gemm_x4.py: `ni_act`), whose accumulators are never stored.  Same requests, reads, waits and barriers, same
    // results; 7 % of those launches' MFMA issues are not made at all on a chip whose clock is set by the matrix pipes' power.
    // A wave with no row below M or no column below N (the 17th column tile of the N = 4160 / 4224 launches has 64 / 128 of its
    // 256 columns) issues none.
    const int rows_left = M - (m0 + wm * 128), cols_left = N - (n0 + wn * 64);
    const int x8_part = !(flags & X4_F_SKIP_PAD) ? 0 : ((rows_left <= 0 || cols_left <= 0) ? 2 : (rows_left <= 32 ? 1 : 0));
    X4_DISPATCH8(V);
    if (probe) pr2 = (long long)__builtin_amdgcn_s_memrealtime();
  }

  char* ep = smem + wave * 16384;
  unsigned wa0, wa1, wa2, wa3;
  {
    const unsigned b0 = (unsigned)(uintptr_t)ep + lr * 256 + ((lg ^ (lr & 3)) << 4);
    wa0 = b0 + ((0 ^ (lr >> 2)) << 6);
    wa1 = b0 + ((1 ^ (lr >> 2)) << 6);
    wa2 = b0 + ((2 ^ (lr >> 2)) << 6);
    wa3 = b0 + ((3 ^ (lr >> 2)) << 6);
  }
#define X8_WRITE(TXT)                                                                                                        \
  asm volatile(TXT ::"v"(wa0), "v"(wa1), "v"(wa2), "v"(wa3), "{a[0:31]}"(c[0]), "{a[32:63]}"(c[1]), "{a[64:95]}"(c[2]),      \
               "{a[96:127]}"(c[3])                                                                                           \
               : "memory")
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    if (h == 0) X8_WRITE(X8_WR_0);
    else X8_WRITE(X8_WR_1);
    // SwiGLU: wave (wm, wn) holds gate columns for wn < 2 and pairs with (wm, wn + 2), as in gemm_256.hip
    x_readout(smem, ep, lane, m0 + wm * 128 + h * 64, wn, n0, wm * 4 + (wn & 1), 2, wn >> 1, wn & 1, Cv, bias, res, M, N, ldc, ldr,
              flags, alpha, probe ? nullptr : aux, ldaux);
  }
  if (probe && tid == 0 && (blockIdx.x & 63) == 0 && blockIdx.y == 0) {
    long long* o = reinterpret_cast<long long*>(aux) + (blockIdx.x >> 6) * 4;
    o[0] = (long long)__builtin_amdgcn_s_memtime() - pc0;
    o[1] = (long long)__builtin_amdgcn_s_memrealtime() - pr0;
    o[2] = pr1 - pr0;                                  // entry -> first request (index math, descriptors)
    o[3] = pr2 - pr0;                                  // entry -> end of the K loop
  }
}

static int x4_variant = 0, x4_same_panel = 0;
static void* x4_clock_probe = nullptr;
static int x4_no_stores = 0;
#ifdef MH_DEBUG_HOOKS   // timing probes and sweep switches: libmyriad_hip_dbg.so only (tools/gemm_x8_clock.py, gemm_x4_sweep.py)
extern "C" void mhdbg_set_gemm_x4_no_stores(int on) { x4_no_stores = on; }   // the read-out without its global stores: WRONG RESULTS
extern "C" void mhdbg_set_gemm_x4_clock_probe(void* p) { x4_clock_probe = p; }   // [8][4] int64 (shader cycles, 100-MHz ticks total / to the loop / to the loop's end) of workgroups 0, 64, ..
extern "C" void mhdbg_set_gemm_x4_same_panel(int on) { x4_same_panel = on; }
extern "C" void mhdbg_set_gemm_x4_variant(int v) { x4_variant = (v >= 0 && v < X4_NVARIANTS) ? v : 0; }
extern "C" int mhdbg_gemm_x4_nvariants() { return X4_NVARIANTS; }
#endif

// Same contract as mh_launch_gemm_256 (gemm_256.hip), which routes here; MH_ERR_UNSUPPORTED sends the caller back to the
// eight-wave kernel (operands whose row offsets do not fit the 32-bit buffer offset).
int mh_launch_gemm_x4(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias,
                      const float* residual, int ldr, int flags, float alpha, int splits, int tps, long split_stride,
                      hipStream_t stream, void* aux, int ldaux) {
  if ((size_t)M * lda * 2 >= (1ull << 31) || (size_t)N * ldb * 2 >= (1ull << 31) || (K & 63)) return MH_ERR_UNSUPPORTED;
  if (flags & (MH_GEMM_SWIGLU_FWD | MH_GEMM_SWIGLU_BWD)) {
    if (!aux || splits != 1 || bias || residual || (flags & (MH_GEMM_OUT_F32 | MH_GEMM_GELU)) || (ldc & 7) || (ldaux & 7) || (N & 127))
      return MH_ERR_ARG;
  }
  const int tiles_m = (M + X4_BM - 1) / X4_BM, tiles_n = (N + X4_BN - 1) / X4_BN;
  const size_t shmem = 2 * X4_BUF;   // 128 KiB -> one 4-wave workgroup per CU
  const dim3 grid(tiles_m * tiles_n, splits);
  if (x4_same_panel) flags |= X4_F_SAME_PANEL;
  if (mh_opt(MH_OPT_GEMM_ZERO_PAD)) flags |= X4_F_ZERO_PAD;
  if (mh_opt(MH_OPT_GEMM_SKIP_PAD)) flags |= X4_F_SKIP_PAD;
  if (splits > 1 && mh_opt(MH_OPT_GEMM_SPLIT_XCD)) flags |= X4_F_SPLIT_XCD;
  if (x4_no_stores) flags |= X4_F_NO_STORES;
  if (x4_clock_probe && !aux && !(flags & (MH_GEMM_SWIGLU_FWD | MH_GEMM_SWIGLU_BWD))) { flags |= X4_F_CLOCK_PROBE; aux = x4_clock_probe; }
  if (g_mh_prof_on) mh_prof_pre(stream, 2, M, N, K, splits, flags);
#define X4_LAUNCH(V)                                                                                                           \
  {                                                                                                                            \
    static bool attr_set = false;                                                                                              \
    if (!attr_set) {                                                                                                           \
      (void)hipFuncSetAttribute((const void*)gemm_x4_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);       \
      attr_set = true;                                                                                                         \
    }                                                                                                                          \
    hipLaunchKernelGGL(gemm_x4_kernel<V>, grid, dim3(256), shmem, stream, (const bf16_t*)A, (const bf16_t*)B, C, bias, residual,   \
                       M, N, K, lda, ldb, ldc, ldr, flags, alpha, tiles_m, tps, split_stride, aux, ldaux);                     \
  }
#define X8_LAUNCH(V)                                                                                                           \
  {                                                                                                                            \
    static bool attr_set = false;                                                                                              \
    if (!attr_set) {                                                                                                           \
      (void)hipFuncSetAttribute((const void*)gemm_x8_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);       \
      attr_set = true;                                                                                                         \
    }                                                                                                                          \
    hipLaunchKernelGGL(gemm_x8_kernel<V>, grid, dim3(512), shmem, stream, (const bf16_t*)A, (const bf16_t*)B, C, bias,         \
                       residual, M, N, K, lda, ldb, ldc, ldr, flags, alpha, tiles_m, tps, split_stride, aux, ldaux);           \
  }
  switch (x4_variant) {
    X4_LAUNCH_SWITCH
    default: return MH_ERR_ARG;
  }
  if (g_mh_prof_on) mh_prof_post(stream);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

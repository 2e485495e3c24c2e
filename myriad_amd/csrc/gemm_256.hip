// K1 (large-tile variant): 256 x 256 x 32 block tile, 8 waves in two software-skewed groups, 4-deep LDS-DMA ring.
//
// Measured reasons for this shape (phase timelines from tools/gemm_pp_trace.py, knock-outs and steady-state rates in
// profiles/r01_gemm_256.md):
//   * the CU's vector-memory path moves 64 B/clk, i.e. one 1-KiB global_load_lds per 16 cycles.  A 128x128x64 tile
//     stages 32 KiB per 512 MFMA-cycles/SIMD (100 % of that path at matrix peak), 256x128 needs 75 %, 256x256 50 %:
//     only the 256x256 tile leaves the DMA path slack at the rates we are after;
//   * a wave that issues LDS-DMA instructions in front of its MFMAs sits ~50 cycles per instruction on that path with
//     the matrix pipe idle, so the prefetch is issued by the group that is in its fragment-read phase, never by
//     the group that is issuing MFMAs;
//   * LDS-DMA issue -> landed is 2-3k cycles under load, so the ring holds tiles t .. t+3 (three 32-deep tiles, i.e.
//     ~3 x 1000 cycles of MFMA work, in flight).  BK = 32 is what lets four stages fit: 4 x 32 KiB = 128 KiB;
//   * steady state is 1.45 PF/s on 4096^2 outputs (the vendor library's rate); what a K = 4096 launch loses is the
//     fixed cost per workgroup (one workgroup per CU: nothing overlaps its prologue and store tail), hence the
//     one-tile prologue wait and the LDS-transposed, full-line epilogue stores below.
//
// Waves: grp = wave >> 2 owns rows grp*128 .. +127, wc = wave & 3 owns columns wc*64 .. +63 (wave tile 128 x 64 =
// 8 x 4 MFMA fragments, 128 accumulator registers; 12 ds_read_b128 feed 32 MFMAs).  One barrier per tile; inside
// interval t
//   group A:  reads(t)  prefetch(t+3)  MFMA(t)
//   group B:  MFMA(t-1) [s_setprio 3]  reads(t)  prefetch(t+3)
// Each SIMD holds one wave of each group, so its matrix pipe sees B's 32 MFMAs, then A's, back to back, with the
// other wave's LDS reads / DMA issue underneath.  B's MFMAs outrank A's (priority, then age) so B drains first and
// its reads hide under A's MFMAs.  Tile t+3 goes into the stage that was read during interval t-1; a share issued in
// interval t is retired by a counted vmcnt (two younger shares stay in flight) before barrier t+2, and first read
// after it.
//
// LDS image: rows of 64 B (32 bf16), four 16-B chunks per row, physical chunk = logical ^ g(row>>2 & 3) with
// g = {0,2,3,1}.  ds_read_b128 is served in lane groups {0-3,12-15,20-27}, ...: rows {0-3,12-15} at chunk c and
// rows {4-11} at chunk c^1 must fall on 16 distinct 16-B slots of the 256-B bank row; slot = (row&3)*4 + physical
// chunk, which needs {g0, g3, 1^g1, 1^g2} and {g1, g2, 1^g0, 1^g3} each distinct -- the identity map (chunk ^
// row>>2) is 2-way conflicted, {0,2,3,1} is conflict-free (SQ_LDS_BANK_CONFLICT = 0 measured).  LDS-DMA writes are
// lane-linear, so the same permutation is applied to the per-lane global source address.
#include "common.h"
#include <cstdlib>

#define G2_BM 256
#define G2_BN 256
#define G2_BK 32
#define G2_NST 4
#define G2_A_BYTES (G2_BM * G2_BK * 2)   // 16 KiB
#define G2_B_BYTES (G2_BN * G2_BK * 2)   // 16 KiB
#define G2_STAGE (G2_A_BYTES + G2_B_BYTES)

#define MH_GEMM_OUT_F32 1
#define MH_GEMM_GELU 2
// SiLU-gated MLP fused into the epilogue (modeling_llama.py:139-140).  The gate|up weight rows are interleaved in blocks of
// 128 ([g 0..127 | u 0..127 | g 128..255 | ...]) so one 256-column tile holds 128 gate columns and their 128 up columns.
//   SWIGLU_FWD: C = gu (bf16, kept for the backward) and aux[M, N/2] = silu(g) * u; waves wc and wc + 2 hold g and u of the
//               same (rows, columns), exchanged through the epilogue's LDS slices.
//   SWIGLU_BWD: the tile is dact = dh . W_down; aux = gu [M, 2N] (interleaved), C = dgu [M, 2N]:
//               dg = dact * u * (s + g s (1 - s)), du = dact * g s, s = sigmoid(g); dact itself is never stored.
// Both round g, u and dact to bf16 before the elementwise math, exactly as the separate silu_mul kernels see them.
#define MH_GEMM_SWIGLU_FWD 16
#define MH_GEMM_SWIGLU_BWD 32

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__device__ __forceinline__ int g2_perm(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // {0,2,3,1}
__device__ __forceinline__ int g2_off(int row, int chunk) { return row * 64 + ((chunk ^ g2_perm(row)) << 4); }

// TRACE build: wave 0 of each group in workgroup 0 stamps s_memtime at the phase boundaries (slot 7: 100 MHz counter)
#define G2_STAMP(slot)                                                                        \
  if (TRACE) {                                                                                \
    if (trace && blockIdx.x == 0 && blockIdx.y == 0 && wc == 0 && lane == 0 && t < 64)        \
      trace[(grp * 64 + t) * 8 + (slot)] = (slot) == 7 ? (long long)__builtin_amdgcn_s_memrealtime() \
                                                       : (long long)__builtin_amdgcn_s_memtime();   \
  }

// TRACE build: every workgroup's life cycle on the 100 MHz counter (entry, first tile landed, main loop done, stores
// issued) at trace[1024 + 4*workgroup + slot] -- tools/gemm_life.py turns it into the fixed-cost breakdown
#define G2_LIFE(slot)                                                                                    \
  if (TRACE) {                                                                                           \
    if (trace && blockIdx.y == 0 && threadIdx.x == 0)                                                    \
      trace[1024 + 4 * blockIdx.x + (slot)] = (long long)__builtin_amdgcn_s_memrealtime();               \
  }

template <int N>
__device__ __forceinline__ void g2_wait_vm() {
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}
__device__ __forceinline__ void g2_wait_younger(int younger_tiles) {   // 4 LDS-DMA instructions per tile share
  if (younger_tiles >= 3) g2_wait_vm<12>();
  else if (younger_tiles == 2) g2_wait_vm<8>();
  else if (younger_tiles == 1) g2_wait_vm<4>();
  else g2_wait_vm<0>();
}

template <bool TRACE>
__global__ __launch_bounds__(512, 2) void gemm_256_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                          void* Cv, const float* __restrict__ bias, const float* res,
                                                          int M, int N, int K, int lda, int ldb, int ldc, int ldr,
                                                          int flags, float alpha, int tiles_m, int kt_per_split,
                                                          long split_stride, long long* trace, void* aux, int ldaux) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [4 stages][A 16K | B 16K]; reused by the epilogue
  G2_LIFE(0)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int lr = lane & 15, lg = lane >> 4;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  // consecutive tiles (one XCD's concurrent 32 workgroups) walk 8 tile-rows before the next tile-column: an 8 x 4
  // patch re-uses each A panel 4x and each B panel 8x out of that XCD's L2 instead of 32 A panels against one B panel
  const int tiles_n = nwg / tiles_m;
  const int per_group = 8 * tiles_n;
  const int first_m = (lid / per_group) * 8;
  const int gsz = (tiles_m - first_m) < 8 ? (tiles_m - first_m) : 8;
  const int tm = first_m + (lid % per_group) % gsz, tn = (lid % per_group) / gsz;
  const int m0 = tm * G2_BM, n0 = tn * G2_BN;

  // staging shares: 1024 chunks of 16 B per operand tile -> 2 + 2 LDS-DMA instructions per thread per tile
  const bf16_t* gA[2];
  const bf16_t* gB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * 512 + tid;
    const int row = c >> 2, lc = (c & 3) ^ g2_perm(row);
    int ra = m0 + row, rb = n0 + row;
    ra = ra < M ? ra : M - 1;
    rb = rb < N ? rb : N - 1;
    gA[i] = A + (size_t)ra * lda + lc * 8;
    gB[i] = B + (size_t)rb * ldb + lc * 8;
  }

  float4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  const int nt_all = K / G2_BK;
  const int kt0 = blockIdx.y * kt_per_split;
  const int nt = (nt_all - kt0) < kt_per_split ? (nt_all - kt0) : kt_per_split;
  if (gridDim.y > 1)                                   // split_stride in elements of the slab type (fp32, or bf16 without OUT_F32)
    Cv = (flags & MH_GEMM_OUT_F32) ? (void*)(reinterpret_cast<float*>(Cv) + blockIdx.y * split_stride)
                                   : (void*)(reinterpret_cast<bf16_t*>(Cv) + blockIdx.y * split_stride);

  auto issue = [&](int t) {
    char* sA = smem + (t & 3) * G2_STAGE;
    char* sB = sA + G2_A_BYTES;
    const int k0 = (kt0 + t) * G2_BK;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(gA[i] + k0), (lds_void_t*)(sA + (i * 512 + wave * 64) * 16), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(gB[i] + k0), (lds_void_t*)(sB + (i * 512 + wave * 64) * 16), 16, 0, 0);
  };

  // prologue: put four tiles in flight but start on the first one (every CU is in its prologue at the same time, so
  // waiting for all four costs the whole 32 MiB burst at HBM rate before any MFMA issues)
#pragma unroll
  for (int t = 0; t < G2_NST; ++t)
    if (t < nt) issue(t);
  g2_wait_younger(nt - 1);
  __builtin_amdgcn_s_barrier();                       // tile 0 visible
  G2_LIFE(1)

  // per-lane fragment byte offsets inside a stage (row & chunk permutation are tile-invariant)
  int offA[8], offB[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) offA[i] = g2_off(grp * 128 + i * 16 + lr, lg);
#pragma unroll
  for (int j = 0; j < 4; ++j) offB[j] = G2_A_BYTES + g2_off(wc * 64 + j * 16 + lr, lg);

  short8_t af[8], bfr[4];
  auto mfma_block = [&]() {
    if (grp == 1) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  for (int t = 0; t < nt; ++t) {
    G2_STAMP(0)
    G2_STAMP(7)
    if (grp == 1 && t > 0) mfma_block();               // fragments of tile t-1, read in the previous interval
    G2_STAMP(1)
    __builtin_amdgcn_sched_barrier(0);
    const char* st = smem + (t & 3) * G2_STAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const short8_t*>(st + offB[j]);
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const short8_t*>(st + offA[i]);
    if (t >= 1 && t + 3 < nt) issue(t + 3);            // behind the reads: DMA issue overlaps the LDS latency
    G2_STAMP(2)
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 0) mfma_block();
    G2_STAMP(3)
    g2_wait_younger(nt - t - 2 < 2 ? nt - t - 2 : 2);  // tile t+1's share has landed (tiles t+2, t+3 may be in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // my reads of tile t are done (WAR on its stage)
    G2_STAMP(4)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    G2_STAMP(5)
  }
  if (grp == 1 && nt > 0) mfma_block();
  G2_LIFE(2)

  // ---- epilogue.  Every LDS-DMA has been retired and every fragment read is done (last barrier), so the ring is
  // free: each wave transposes its 128 x 64 fp32 tile through a private 16-KiB slice, 64 rows at a time, so that the
  // global stores are whole rows -- 16 lanes x 8 B (bf16) or x 16 B (fp32) = one or two full 128-B lines per row --
  // instead of the MFMA layout's 32-B fragments (the store tail was ~1/3 of a K = 4096 launch).  Bias / GELU /
  // residual are applied on the way out, where a lane holds 4 consecutive columns of one row.
  const bool out_f32 = flags & MH_GEMM_OUT_F32;
  const bool do_gelu = flags & MH_GEMM_GELU;
  const bool sw_fwd = flags & MH_GEMM_SWIGLU_FWD, sw_bwd = flags & MH_GEMM_SWIGLU_BWD;
  char* ep = smem + wave * 16384;                      // [64 rows][16 chunks of 16 B], chunk ^= row & 15
  const int er = lane >> 4, ec = lane & 15;            // write-out: 4 rows per pass, lane owns columns ec*4 .. +3
  const int ncol = n0 + wc * 64 + ec * 4;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int i = h * 4 + ii;
      const int row = ii * 16 + lr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = j * 4 + lg;
        *reinterpret_cast<float4_t*>(ep + row * 256 + ((c ^ (row & 15)) << 4)) =
            (float4_t){acc[i][j][0] * alpha, acc[i][j][1] * alpha, acc[i][j][2] * alpha, acc[i][j][3] * alpha};
      }
    }
    // the slice is private to the wave: program order + the compiler's lgkmcnt wait are the only ordering needed
    if (sw_bwd) {
      // dact (this wave: 64 rows x 64 columns of the [M, N] product) -> dg, du at the interleaved positions of gu / dgu
      const int r8 = lane >> 3, c8 = lane & 7;
      const int nc = n0 + wc * 64 + c8 * 8;                       // dact column
      const long gcol = (long)(nc >> 7) * 256 + (nc & 127);       // gate column in the interleaved [M, 2N] layout (up: + 128)
      const bf16_t* gu_in = reinterpret_cast<const bf16_t*>(aux);
      bf16_t* dgu = reinterpret_cast<bf16_t*>(Cv);
#pragma unroll 4
      for (int p = 0; p < 8; ++p) {
        const int row = p * 8 + r8;
        const int m = m0 + grp * 128 + h * 64 + row;
        const float4_t va = *reinterpret_cast<const float4_t*>(ep + row * 256 + (((2 * c8) ^ (row & 15)) << 4));
        const float4_t vb = *reinterpret_cast<const float4_t*>(ep + row * 256 + (((2 * c8 + 1) ^ (row & 15)) << 4));
        if (m >= M || nc >= N) continue;
        const float v[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
        const short8_t g8 = *reinterpret_cast<const short8_t*>(gu_in + (size_t)m * ldaux + gcol);
        const short8_t u8 = *reinterpret_cast<const short8_t*>(gu_in + (size_t)m * ldaux + gcol + 128);
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
      continue;
    }
    if (sw_fwd) {
      // every wave of the group has written its slice; wave (grp, wc) pairs with (grp, wc ^ 2): columns wc & 1, rows
      // 0..31 (wc < 2) or 32..63 (wc >= 2) of this 64-row half -> act = silu(g) * u, 32 rows x 64 columns per wave
      __syncthreads();
      const int r8 = lane >> 3, c8 = lane & 7;
      const char* sg_ = smem + (grp * 4 + (wc & 1)) * 16384;       // gate slice
      const char* su_ = smem + (grp * 4 + (wc & 1) + 2) * 16384;   // up slice
      bf16_t* act = reinterpret_cast<bf16_t*>(aux);
      const int ac = (n0 >> 1) + (wc & 1) * 64 + c8 * 8;           // act column (natural order)
#pragma unroll 4
      for (int p = 0; p < 4; ++p) {
        const int row = (wc >> 1) * 32 + p * 8 + r8;
        const int m = m0 + grp * 128 + h * 64 + row;
        const int o0 = row * 256 + (((2 * c8) ^ (row & 15)) << 4), o1 = row * 256 + (((2 * c8 + 1) ^ (row & 15)) << 4);
        const float4_t ga = *reinterpret_cast<const float4_t*>(sg_ + o0), gb = *reinterpret_cast<const float4_t*>(sg_ + o1);
        const float4_t ua = *reinterpret_cast<const float4_t*>(su_ + o0), ub = *reinterpret_cast<const float4_t*>(su_ + o1);
        if (m >= M || ac >= (N >> 1)) continue;
        const float gg[8] = {ga[0], ga[1], ga[2], ga[3], gb[0], gb[1], gb[2], gb[3]};
        const float uu[8] = {ua[0], ua[1], ua[2], ua[3], ub[0], ub[1], ub[2], ub[3]};
        short8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float gv = bf2f(f2bf(gg[e])), uv = bf2f(f2bf(uu[e]));
          o[e] = (short)f2bf(gv / (1.f + __expf(-gv)) * uv);
        }
        *reinterpret_cast<short8_t*>(act + (size_t)m * ldaux + ac) = o;
      }
      // fall through: the bf16 store path below writes this wave's own gu columns
    }
    if (!out_f32 && (ldc & 7) == 0) {
      // bf16 output: 16-B stores, 8 lanes cover one 128-B row, 8 rows per pass (half the store instructions of the
      // 8-B form; the store tail is issue-bound, not bandwidth-bound)
      const int r8 = lane >> 3, c8 = lane & 7;
      const int nc = n0 + wc * 64 + c8 * 8;
#pragma unroll 4
      for (int p = 0; p < 8; ++p) {
        const int row = p * 8 + r8;
        const int m = m0 + grp * 128 + h * 64 + row;
        const float4_t va = *reinterpret_cast<const float4_t*>(ep + row * 256 + (((2 * c8) ^ (row & 15)) << 4));
        const float4_t vb = *reinterpret_cast<const float4_t*>(ep + row * 256 + (((2 * c8 + 1) ^ (row & 15)) << 4));
        if (m >= M || nc >= N) continue;
        float v[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
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
      if (sw_fwd) __syncthreads();                     // the partner has read this slice before the next half overwrites it
      continue;
    }
#pragma unroll 4
    for (int p = 0; p < 16; ++p) {
      const int row = p * 4 + er;
      const int m = m0 + grp * 128 + h * 64 + row;
      const float4_t v4 = *reinterpret_cast<const float4_t*>(ep + row * 256 + ((ec ^ (row & 15)) << 4));
      if (m >= M || ncol >= N) continue;
      float v[4] = {v4[0], v4[1], v4[2], v4[3]};
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
  G2_LIFE(3)
}

static long long* g2_trace = nullptr;
#ifdef MH_DEBUG_HOOKS
extern "C" void mhdbg_set_gemm256_trace(void* p) { g2_trace = (long long*)p; }   // phase stamps (libmyriad_hip_dbg.so only)
#endif

// Round 4: the same tile as a four-wave, 64-deep, hand-scheduled instruction stream (gemm_x4.hip) for launches with long K
// per workgroup; both kernels produce the same bits.  Debug hook: 0 eight-wave, 1 four-wave, 2 / -1 the policy below.
int mh_launch_gemm_x4(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias,
                      const float* residual, int ldr, int flags, float alpha, int splits, int tps, long split_stride,
                      hipStream_t stream, void* aux, int ldaux);

// tps / split_stride as in gemm.hip (tps in 64-deep K tiles)
int mh_launch_gemm_256(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                       const float* bias, const float* residual, int ldr, int flags, float alpha, int splits, int tps,
                       long split_stride, hipStream_t stream, void* aux, int ldaux) {
  if (flags & (MH_GEMM_SWIGLU_FWD | MH_GEMM_SWIGLU_BWD)) {
    // one 256-column tile = whole 128-column blocks; 16-byte rows; the fused forms never split K or take bias / residual
    if (!aux || splits != 1 || bias || residual || (flags & (MH_GEMM_OUT_F32 | MH_GEMM_GELU)) || (ldc & 7) || (ldaux & 7) || (N & 127))
      return MH_ERR_ARG;
  }
  // policy (tools/gemm_x4_sweep.py, profiles/r04_gemm_x4.md): the hand-scheduled 64-deep loop (gemm_x4.hip, eight-wave form)
  // is ahead of the kernel above on every shape of the step -- 1.31 against 1.49 us per 64-deep k-tile at the same ~13 us
  // outside the loop -- so it takes every launch of plan kernel 2 it supports.  MYRIAD_GEMM256_IMPL=0 keeps the kernel above.
  if (mh_opt(MH_OPT_GEMM256_IMPL) && !g2_trace) {
    const int rc = mh_launch_gemm_x4(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, flags, alpha, splits, tps, split_stride,
                                     stream, aux, ldaux);
    if (rc != MH_ERR_UNSUPPORTED) return rc;
  }
  const int tiles_m = (M + G2_BM - 1) / G2_BM, tiles_n = (N + G2_BN - 1) / G2_BN;
  const size_t shmem = G2_NST * G2_STAGE;   // 128 KiB -> one 8-wave workgroup per CU
  // What the shipped library keeps of this file is the plain kernel: the fallback for operands whose byte offsets pass 2 GiB
  // (the hand-scheduled loop addresses rows through 32-bit buffer offsets) and the bit-identity partner behind option
  // gemm256_impl = 0.  The phase-stamping instance exists only in libmyriad_hip_dbg.so.
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
#ifdef MH_DEBUG_HOOKS
    (void)hipFuncSetAttribute((const void*)gemm_256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
#endif
    attr_set = true;
  }
  const dim3 grid(tiles_m * tiles_n, splits), block(512);
  if (g_mh_prof_on) mh_prof_pre(stream, 2, M, N, K, splits, flags);
#ifdef MH_DEBUG_HOOKS
  if (g2_trace)
    hipLaunchKernelGGL((gemm_256_kernel<true>), grid, block, shmem, stream, (const bf16_t*)A, (const bf16_t*)B, C, bias,
                       residual, M, N, K, lda, ldb, ldc, ldr, flags, alpha, tiles_m, tps * 2, split_stride, g2_trace, aux, ldaux);
  else
#endif
    hipLaunchKernelGGL((gemm_256_kernel<false>), grid, block, shmem, stream, (const bf16_t*)A, (const bf16_t*)B, C, bias,
                       residual, M, N, K, lda, ldb, ldc, ldr, flags, alpha, tiles_m, tps * 2, split_stride, nullptr, aux, ldaux);
  if (g_mh_prof_on) mh_prof_post(stream);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

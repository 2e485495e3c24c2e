// K1, experimental mid-M weight-streaming variant (variant 13; NOT selected by the automatic policy unless
// MYRIAD_GEMM_STREAM=1): C[M, N] = alpha * A[M, K] . B[N, K]^T for 16 < M <= 288 -- the batch-1 fine-tune step (148 LLaMA
// rows, 257 ViT rows; BASELINE configs[1]) and generate()'s prefill.
//
// At these M a launch reads its weight matrix once (100-180 MB) and does little arithmetic per byte: an HBM-streaming
// problem.  The idea is the decode gemv's, with the activation rows shared:
//   * a workgroup of 8 waves owns 256 weight rows (32 per wave, two MFMA fragments) and a K slice;
//   * every wave loads ITS weight rows straight from global memory into registers in MFMA operand layout, 32 contiguous
//     bytes per lane per 64-deep step (whole 128-B lines, nontemporal), D-1 steps ahead -- no LDS round trip, never shared;
//   * the activation tile [16*MF rows x 64] of a step is shared by the 8 waves through a D-deep LDS ring filled with LDS-DMA
//     by all waves, D-1 steps ahead (D = 4, or 3 above 192 rows).  vmcnt retires in order, so one counted wait per step
//     covers both the tile and the weights of the next step and leaves D-2 steps of both in flight;
//   * v_mfma_f32_16x16x32_bf16 with the weight fragment first, so a lane ends up with 4 consecutive n of one row (16-byte
//     stores); the k permutation (lane group lg holds k = 16*lg .. +15 of the step) is applied to both operands; an
//     activation fragment read from LDS feeds both weight fragments;
//   * one s_barrier per step; split-K over grid.y with the usual fixed-order reduce (splitk_reduce_kernel in gemm.hip).
// Status (profiles/r01_gemm_stream.md): bit-for-bit tested (tests/test_kernels_gpu.py::test_mid_m_weight_streaming_gemm) and
// at parity with the tile kernels, not ahead -- 148 x 22016 x 4096 in 73 us vs 68 us, 148 x 12352 x 4096 48 vs 43 us.
// Knock-outs show the loop is hidden behind the weight stream (no MFMA, no LDS reads, no barrier: same time); what is left
// is (a) the weight request order itself -- 4.9 TB/s against 6.5 TB/s for a linear read (tools/micro/stream_pattern.hip),
// (b) CU fill: 256-row workgroups give 86 x 2 = 172 of them for N = 22016, and (c) per-launch fixed cost that no kernel
// body removes: the split-K slabs of a 148-row output (5 x 7.3 MB written and re-read for N = 12352, as much as two thirds
// of the weight bytes), prologue, launch.  Kept as the starting point for a no-split version with uneven row ownership.
#include "common.h"

#define GS_BN 256

#define MH_GEMM_OUT_F32 1
#define MH_GEMM_GELU 2

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int N>
__device__ __forceinline__ void gs_wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt immediate");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));   // vmcnt(N); expcnt / lgkmcnt fields at their maximum
}

template <int MF>
struct GsCfg {
  static constexpr int XL = (MF + 3) / 4;             // LDS-DMA instructions per thread per stage (512 chunks of 16 B each)
  static constexpr int XSTAGE = XL * 8192;            // stage stride: the instruction grid, not the row count
  static constexpr int D = MF > 12 ? 3 : 4;           // ring depth
  static constexpr int YOUNGER = (D - 3) * (XL + 4) + XL;   // at a step's barrier: the loads of steps t+2 .. t+D-2 and the tile
                                                            // of step t+D-1 (its weights are issued after the barrier) may fly
};

template <int MF>
__global__ __launch_bounds__(512) void gemm_stream_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, void* Cv,
                                                          const float* __restrict__ bias, const float* res, int M, int N,
                                                          int K, int lda, int ldb, int ldc, int ldr, int flags, float alpha,
                                                          int steps_per_split, long split_stride) {
  using C = GsCfg<MF>;
  constexpr int XL = C::XL, XSTAGE = C::XSTAGE, D = C::D, NS = D - 1;
  constexpr int G = 2;                               // fragments per LDS read group (divides MF)
  extern __shared__ __attribute__((aligned(16))) char xs[];   // [D][XL * 512 chunks of 16 B]: row r at r*128, chunk ^= (r >> 1) & 7
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.x * GS_BN + wave * 32;
  const int nsteps_all = K / 64;
  const int s_begin = blockIdx.y * steps_per_split;
  const int nsteps = (nsteps_all - s_begin) < steps_per_split ? (nsteps_all - s_begin) : steps_per_split;
  if (gridDim.y > 1) Cv = reinterpret_cast<float*>(Cv) + blockIdx.y * split_stride;

  // activation tile: chunk c = i*512 + tid of a stage <- A[row = c >> 3][k0 + 8 * ((c & 7) ^ ((row >> 1) & 7))]; rows past M repeat
  // the last row (their products are never stored)
  const bf16_t* xsrc[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int c = i * 512 + tid;
    const int row = c >> 3, lc = (c & 7) ^ ((row >> 1) & 7);
    xsrc[i] = A + (size_t)(row < M ? row : M - 1) * lda + (size_t)s_begin * 64 + lc * 8;
  }
  auto issue_x = [&](int t) {                         // t may run past the slice: clamped, lands in a stage nobody reads again
    const int tc = t < nsteps ? t : nsteps - 1;
    char* st = xs + (t % D) * XSTAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < XL; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(xsrc[i] + tc * 64), (lds_void_t*)(st + i * 8192), 16, 0, 0);
  };
  const bf16_t* wp[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    int nrow = n0 + f * 16 + lr;
    nrow = nrow < N ? nrow : N - 1;
    wp[f] = B + (size_t)nrow * ldb + (size_t)s_begin * 64 + lg * 16;
  }
  short8_t w[NS][2][2];
  auto load_w = [&](int set, int t) {
    const int tc = t < nsteps ? t : nsteps - 1;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      w[set][f][0] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp[f] + tc * 64));
      w[set][f][1] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp[f] + tc * 64 + 8));
    }
  };
  float4_t acc[2][MF];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[f][i] = (float4_t){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int j = 0; j < NS; ++j) { issue_x(j); load_w(j, j); }
  gs_wait_vm<(D - 2) * (XL + 4)>();                   // step 0's tile and weights are in
  __builtin_amdgcn_s_barrier();

  // per-lane activation fragment offsets: row i*16 + lr, logical chunks 2*lg and 2*lg+1 (k = 16*lg .. 16*lg+15)
  // (the swizzle key is (row >> 1) & 7: a 16-lane ds_read_b128 phase covers 16 rows x 16 B = all 64 banks exactly once)
  const int xoff0 = lr * 128 + (((2 * lg) ^ ((lr >> 1) & 7)) << 4);
  const int xoff1 = lr * 128 + (((2 * lg + 1) ^ ((lr >> 1) & 7)) << 4);
  // Activation fragments are read in groups of G, one group ahead of the MFMAs that use them (two waves per SIMD cannot
  // hide a ds_read -> MFMA round trip per fragment).  The step's barrier sits in front of the LAST group's MFMAs and the
  // first group of the NEXT tile is read right behind it, so the LDS phase of a step overlaps the MFMA tail of the previous
  // one instead of every wave reading, then every wave multiplying, in lockstep.
  constexpr int NG = MF / G;
  short8_t xa[2][G][2], xn[G][2];
  auto read_group = [&](short8_t (*dst)[2], const char* st, int g) {
#pragma unroll
    for (int j = 0; j < G; ++j) {
      dst[j][0] = *reinterpret_cast<const short8_t*>(st + (g * G + j) * 2048 + xoff0);
      dst[j][1] = *reinterpret_cast<const short8_t*>(st + (g * G + j) * 2048 + xoff1);
    }
  };
  read_group(xn, xs, 0);
  auto step = [&](int set, int t) {                   // set = t % NS: a compile-time constant after unrolling
    issue_x(t + NS);                                  // its stage was last read in step t-1 (all of it consumed by then)
    const char* st = xs + (t % D) * XSTAGE;
#pragma unroll
    for (int j = 0; j < G; ++j) { xa[0][j][0] = xn[j][0]; xa[0][j][1] = xn[j][1]; }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      // With an LDS-DMA load pending the compiler only ever emits lgkmcnt(0) (it files global_load_lds under "flat"), so
      // a counted read-ahead is not available: wait for group g HERE, then issue group g+1, then multiply group g.
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0)
      __builtin_amdgcn_sched_barrier(0);              // keep that order: the scheduler would sink the reads back
      if (g + 1 < NG) {
        read_group(xa[(g + 1) & 1], st, g + 1);
      } else {
        gs_wait_vm<C::YOUNGER>();                     // step t+1's tile and weights have landed (this wave's share)
        __builtin_amdgcn_s_barrier();                 // ... everybody's; and everybody has issued its reads of tile t
        read_group(xn, xs + ((t + 1) % D) * XSTAGE, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < G; ++j) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          acc[f][g * G + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[set][f][0], xa[g & 1][j][0], acc[f][g * G + j], 0, 0, 0);
          acc[f][g * G + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[set][f][1], xa[g & 1][j][1], acc[f][g * G + j], 0, 0, 0);
        }
      }
    }
    load_w(set, t + NS);                              // refill the set just consumed
  };
  int t0 = 0;
  if (nsteps >= NS) {                                 // first group peeled: the loop header then merges two identical states
#pragma unroll
    for (int u = 0; u < NS; ++u) step(u, u);
    t0 = NS;
  }
  for (; t0 + NS <= nsteps; t0 += NS) {               // branch-free steady state
#pragma unroll
    for (int u = 0; u < NS; ++u) step(u, t0 + u);
  }
#pragma unroll
  for (int u = 0; u < NS - 1; ++u)
    if (t0 + u < nsteps) step(u, t0 + u);

  // epilogue: lane owns C[m = i*16 + lr][n .. n+3], n = n0 + f*16 + 4*lg
  const bool out_f32 = flags & MH_GEMM_OUT_F32;
  const bool do_gelu = flags & MH_GEMM_GELU;
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int n = n0 + f * 16 + 4 * lg;
    if (n >= N) continue;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      const int m = i * 16 + lr;
      if (m >= M) continue;
      float v[4] = {acc[f][i][0] * alpha, acc[f][i][1] * alpha, acc[f][i][2] * alpha, acc[f][i][3] * alpha};
      if (n + 3 < N) {
        if (bias) {
          const float4_t b4 = *reinterpret_cast<const float4_t*>(bias + n);
          v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3];
        }
        if (do_gelu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
        if (res) {
          const float4_t r4 = *reinterpret_cast<const float4_t*>(res + (size_t)m * ldr + n);
          v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3];
        }
        if (out_f32) {
          *reinterpret_cast<float4_t*>(reinterpret_cast<float*>(Cv) + (size_t)m * ldc + n) = (float4_t){v[0], v[1], v[2], v[3]};
        } else {
          uint2 pk;
          pk.x = pack_bf2(v[0], v[1]);
          pk.y = pack_bf2(v[2], v[3]);
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + (size_t)m * ldc + n) = pk;
        }
      } else {
        for (int e = 0; e < 4 && n + e < N; ++e) {
          float x = v[e];
          if (bias) x += bias[n + e];
          if (do_gelu) x = gelu_erf(x);
          if (res) x += res[(size_t)m * ldr + n + e];
          if (out_f32) reinterpret_cast<float*>(Cv)[(size_t)m * ldc + n + e] = x;
          else reinterpret_cast<bf16_t*>(Cv)[(size_t)m * ldc + n + e] = f2bf(x);
        }
      }
    }
  }
}

template <int MF>
static int gs_launch(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias,
                     const float* residual, int ldr, int flags, float alpha, int splits, int steps_per_split, long split_stride,
                     hipStream_t stream) {
  const size_t shmem = (size_t)GsCfg<MF>::D * GsCfg<MF>::XSTAGE;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_stream_kernel<MF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_stream_kernel<MF>, dim3((N + GS_BN - 1) / GS_BN, splits), dim3(512), shmem, stream, (const bf16_t*)A,
                     (const bf16_t*)B, C, bias, residual, M, N, K, lda, ldb, ldc, ldr, flags, alpha, steps_per_split,
                     split_stride);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// steps_per_split in 64-deep K steps; 16 < M <= 288
int mh_launch_gemm_stream(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                          const float* bias, const float* residual, int ldr, int flags, float alpha, int splits,
                          int steps_per_split, long split_stride, hipStream_t stream) {
#define GS_CASE(MF)                                                                                                       \
  if (M <= MF * 16)                                                                                                       \
    return gs_launch<MF>(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, flags, alpha, splits, steps_per_split,      \
                         split_stride, stream);
  GS_CASE(4) GS_CASE(8) GS_CASE(10) GS_CASE(12) GS_CASE(16) GS_CASE(18)
#undef GS_CASE
  return MH_ERR_UNSUPPORTED;
}

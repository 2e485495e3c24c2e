// K1, experimental mid-M weight-streaming variant (variant 13; NOT selected by the automatic policy unless
// MYRIAD_GEMM_STREAM=1): C[M, N] = alpha * A[M, K] . B[N, K]^T for 16 < M <= 288 -- the batch-1 fine-tune step (148 LLaMA
// rows; BASELINE configs[1]) and generate()'s prefill.
//
// At these M a launch reads its weight matrix once (100-180 MB) and does little arithmetic per byte: an HBM-streaming
// problem whose fixed costs (launch, prologue, split-K slabs of a 148-row output) weigh as much as the stream itself.
// Third design, built on what the first two measured (profiles/r01_gemm_stream.md):
//   * a workgroup = 6 compute waves + 2 loader waves.  Compute wave c owns ONE 16-row weight fragment and loads it straight
//     from global memory into registers in MFMA operand layout (32 contiguous bytes per lane per 64-deep step, nontemporal),
//     U = 8 steps ahead: no LDS round trip, never shared, and -- the loads being the only vector-memory traffic of the wave
//     -- the compiler's own vmcnt bookkeeping stays exact;
//   * the two loader waves fill a 4-deep LDS ring with the activation tile [16*MF rows x 64] of each step by LDS-DMA, three
//     steps ahead, on their own vmcnt counters (mixed into the compute waves, the in-order counter would tie the weight
//     prefetch depth to the ring depth -- the second design's limit);
//   * a workgroup owns `nfb` <= 6 consecutive fragments, chosen per shape so that ceil(N/16/nfb) workgroups fill the chip in
//     ONE round WITHOUT splitting K wherever N allows (N = 22016: 6 x 230; N = 12352: 4 x 193), because at M = 148 every
//     extra K slice costs a 7 MB fp32 slab written and read back.  Compute waves beyond nfb only keep the barrier count;
//   * v_mfma_f32_16x16x32_bf16 with the weight fragment first (a lane ends up with 4 consecutive n of one row: 16-byte
//     stores), the k permutation (lane group lg holds k = 16*lg .. +15) applied to both operands, one s_barrier per step.
// Status: bit-tested (tests/test_kernels_gpu.py::test_mid_m_weight_streaming_gemm); 148 x 22016 x 4096 in 60-69 us against
// 68-70 us for the 256x256 kernel, every other shape of the batch-1 step equal or behind -- so the policy does not select it.
// Knock-outs name the limiter: without the activation-tile DMA the same kernel takes 47 us (the weight stream at 3.8 TB/s),
// without MFMA and LDS reads it still takes 64 us.  Every workgroup re-reads all 148 activation rows (230 x 1.2 MB = 295 MB
// through L2 -> LDS per launch, 1.6x the weight bytes) and that traffic does not hide under the weight stream; fewer, fatter
// workgroups trade it for split-K slabs of the same size.  A 148-row GEMM wants the activation slice RESIDENT in LDS
// (148 x 512 x 2 B = 151 KB) and an order-preserving cross-workgroup reduction -- round 2.
#include "common.h"

#define GS_NC 6                  // compute waves
#define GS_NL 2                  // loader waves
#define GS_U 8                   // weight steps in flight per compute wave
#define GS_D 4                   // activation ring depth

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
__global__ __launch_bounds__(512) void gemm_stream_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, void* Cv,
                                                          const float* __restrict__ bias, const float* res, int M, int N,
                                                          int K, int lda, int ldb, int ldc, int ldr, int flags, float alpha,
                                                          int steps_per_split, long split_stride, int nfb) {
  constexpr int XSTAGE = MF * 2048;                   // [16*MF rows][128 B], chunk ^= (row >> 1) & 5
  extern __shared__ __attribute__((aligned(16))) char xs[];   // [GS_D][XSTAGE]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int nsteps_all = K / 64;
  const int s_begin = blockIdx.y * steps_per_split;
  const int nsteps = (nsteps_all - s_begin) < steps_per_split ? (nsteps_all - s_begin) : steps_per_split;
  if (gridDim.y > 1) Cv = reinterpret_cast<float*>(Cv) + blockIdx.y * split_stride;

  if (wave >= GS_NC) {
    // ---------------------------------------------------------------- loader waves: MF wave-instructions (1 KiB) per tile each
    const int lw = wave - GS_NC;
    const bf16_t* src[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      const int c = (i * GS_NL + lw) * 64 + lane;     // 16-B chunk of the stage; rows past M repeat the last row
      const int row = c >> 3, lc = (c & 7) ^ ((row >> 1) & 5);
      src[i] = A + (size_t)(row < M ? row : M - 1) * lda + (size_t)s_begin * 64 + lc * 8;
    }
    auto issue = [&](int t) {                         // t may run past the slice: clamped, lands in a stage nobody reads again
      const int tc = t < nsteps ? t : nsteps - 1;
      char* st = xs + (t % GS_D) * XSTAGE + lw * 1024;
#pragma unroll
      for (int i = 0; i < MF; ++i)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(src[i] + tc * 64), (lds_void_t*)(st + i * GS_NL * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int j = 0; j < GS_D - 1; ++j) issue(j);
    gs_wait_vm<(GS_D - 2) * MF>();                    // tile 0 has landed (this wave's half)
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < nsteps; ++t) {
      issue(t + GS_D - 1);                            // its stage was read during step t-1, before the last barrier
      gs_wait_vm<(GS_D - 2) * MF>();                  // tile t+1 has landed
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // ------------------------------------------------------------------ compute waves: one 16-row weight fragment each
  const int frag = blockIdx.x * nfb + wave;
  const int n0 = frag * 16;
  if (wave >= nfb || n0 >= N) {                       // no fragment: keep the barrier count (1 + nsteps) and leave
    for (int t = 0; t <= nsteps; ++t) __builtin_amdgcn_s_barrier();
    return;
  }
  int nrow = n0 + lr;
  nrow = nrow < N ? nrow : N - 1;
  const bf16_t* wp = B + (size_t)nrow * ldb + (size_t)s_begin * 64 + lg * 16;
  float4_t acc[MF];
#pragma unroll
  for (int i = 0; i < MF; ++i) acc[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
  short8_t w0[GS_U], w1[GS_U];
#pragma unroll
  for (int u = 0; u < GS_U; ++u) {
    const int t = u < nsteps ? u : nsteps - 1;        // short slices re-read their last step; those registers go unused
    w0[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + t * 64));
    w1[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + t * 64 + 8));
  }
  // per-lane activation fragment offsets: row i*16 + lr, logical chunks 2*lg and 2*lg+1 (k = 16*lg .. 16*lg+15).  Swizzle
  // key (row >> 1) & 5: a ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (rows 0-3 and
  // 12-15 of one k group with rows 4-11 of the next), and with this key each group touches all 64 banks exactly once
  // (the obvious (row >> 1) & 7 is 2-way conflicted under that grouping)
  const int xoff0 = lr * 128 + (((2 * lg) ^ ((lr >> 1) & 5)) << 4);
  const int xoff1 = lr * 128 + (((2 * lg + 1) ^ ((lr >> 1) & 5)) << 4);
  __builtin_amdgcn_s_barrier();                       // tile 0 visible (pairs with the loaders' first barrier)
  auto step = [&](int u, int t) {                     // u: compile-time register set after unrolling
    const char* st = xs + (t % GS_D) * XSTAGE;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      const short8_t x0 = *reinterpret_cast<const short8_t*>(st + i * 2048 + xoff0);
      const short8_t x1 = *reinterpret_cast<const short8_t*>(st + i * 2048 + xoff1);
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[u], x0, acc[i], 0, 0, 0);   // D[n = 4*lg + r][m = lr]
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[u], x1, acc[i], 0, 0, 0);
    }
    const int tn = (t + GS_U) < nsteps ? (t + GS_U) : nsteps - 1;   // refill this set: 2*(U-1) loads stay younger
    w0[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + tn * 64));
    w1[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + tn * 64 + 8));
    __builtin_amdgcn_s_barrier();                     // pairs with the loaders: tile t+1 visible, tile t's stage reusable
  };
  int t0 = 0;
  if (nsteps >= GS_U) {                               // first group peeled: the loop header then merges two identical states
#pragma unroll
    for (int u = 0; u < GS_U; ++u) step(u, u);
    t0 = GS_U;
  }
  for (; t0 + GS_U <= nsteps; t0 += GS_U) {           // branch-free steady state: the compiler's vmcnt stays exact
#pragma unroll
    for (int u = 0; u < GS_U; ++u) step(u, t0 + u);
  }
#pragma unroll
  for (int u = 0; u < GS_U - 1; ++u)
    if (t0 + u < nsteps) step(u, t0 + u);

  // epilogue: lane owns C[m = i*16 + lr][n .. n+3], n = n0 + 4*lg
  const bool out_f32 = flags & MH_GEMM_OUT_F32;
  const bool do_gelu = flags & MH_GEMM_GELU;
  {
    const int n = n0 + 4 * lg;
    if (n >= N) return;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      const int m = i * 16 + lr;
      if (m >= M) continue;
      float v[4] = {acc[i][0] * alpha, acc[i][1] * alpha, acc[i][2] * alpha, acc[i][3] * alpha};
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
                     int nfb, hipStream_t stream) {
  const size_t shmem = (size_t)GS_D * MF * 2048;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_stream_kernel<MF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    attr_set = true;
  }
  const int frags = (N + 15) / 16;
  hipLaunchKernelGGL(gemm_stream_kernel<MF>, dim3((frags + nfb - 1) / nfb, splits), dim3(512), shmem, stream, (const bf16_t*)A,
                     (const bf16_t*)B, C, bias, residual, M, N, K, lda, ldb, ldc, ldr, flags, alpha, steps_per_split,
                     split_stride, nfb);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// Fragments per workgroup and K slices: fill the 256 CUs in one round, prefer no split (slab traffic) and many fragments per
// workgroup (every workgroup re-reads the activation rows), at least 8 steps of 64 per slice.
void mh_gemm_stream_plan(int M, int N, int K, int can_split, int* nfb, int* splits) {
  (void)M;
  const int frags = (N + 15) / 16, steps = K / 64;
  double best = -1.0;
  *nfb = GS_NC; *splits = 1;
  for (int f = GS_NC; f >= 1; --f) {
    const int bn = (frags + f - 1) / f;
    for (int s = 1; s <= (can_split ? 16 : 1); ++s) {
      if (s > 1 && steps / s < 8) break;
      const long blocks = (long)bn * s;
      if (blocks > 256) continue;                     // N > 24576: nothing fits one round -> 6 fragments, no split, several rounds
      const double score = (double)blocks / 256.0 - 0.03 * (s - 1) - 0.04 * (GS_NC - f);
      if (score > best + 1e-9) { best = score; *nfb = f; *splits = s; }
    }
  }
}

// steps_per_split in 64-deep K steps; 16 < M <= 288
int mh_launch_gemm_stream(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                          const float* bias, const float* residual, int ldr, int flags, float alpha, int splits,
                          int steps_per_split, long split_stride, int nfb, hipStream_t stream) {
  if (nfb < 1 || nfb > GS_NC) return MH_ERR_ARG;
#define GS_CASE(MF)                                                                                                       \
  if (M <= MF * 16)                                                                                                       \
    return gs_launch<MF>(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, flags, alpha, splits, steps_per_split,      \
                         split_stride, nfb, stream);
  GS_CASE(4) GS_CASE(8) GS_CASE(10) GS_CASE(12) GS_CASE(16) GS_CASE(18)
#undef GS_CASE
  return MH_ERR_UNSUPPORTED;
}

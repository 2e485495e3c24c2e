// K1 (ping-pong variant): 256 x 128 x 64 block tile, 8 waves = two 4-wave groups that alternate roles every phase.
//
// Why: PMC on the 128x128 kernel (profiles/r01_gemm_pmc.md) shows 46 % MFMA utilisation with zero LDS bank
// conflicts -- the two co-resident workgroups of a CU tend to wait (vmcnt + barrier) at the same time and to
// contend for the matrix pipe at the same time.  Here the overlap is made deterministic instead of statistical
// (MI355X_MICROARCH "Two waves per SIMD"): group A (waves 0-3) and group B (waves 4-7) put one wave each on every
// SIMD; all 8 waves pass one s_barrier per phase and B runs one phase behind A, so while one group issues its 32
// MFMAs (operands already in VGPRs, s_setprio 1) the other does its 16 ds_read_b128 + its share of the LDS-DMA
// prefetch.  Each group owns a 128 x 128 sub-tile (wave tile 64 x 64, as in the 4-wave kernel) and both share
// the 128-row B tile, so LDS-DMA / L2 traffic per flop drops to 0.75x.
//
//   barrier #     0    1        2        3        4
//   group A      | R(0) | M(0)   | R(1)   | M(1)   | ...          R = fragment reads of tile t, M = 32 MFMAs
//   group B      | --   | R(0)   | M(0)   | R(1)   | ...          (B idles one phase at the start, A at the end)
//   prefetch: each wave issues ITS share of tile t+2 right after barrier #(2t+2) (A: start of R(t+1), B: start of
//   M(t)) into the stage both groups finished reading, and waits for it before barrier #(2t+4).
#include "common.h"

#define PP_BM 256
#define PP_BN 128
#define PP_BK 64
#define PP_A_BYTES (PP_BM * PP_BK * 2)   // 32 KiB
#define PP_B_BYTES (PP_BN * PP_BK * 2)   // 16 KiB
#define PP_STAGE (PP_A_BYTES + PP_B_BYTES)

#define MH_GEMM_OUT_F32 1
#define MH_GEMM_GELU 2

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__device__ __forceinline__ int pp_swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// TRACE: wave 0 (group A) and wave 4 (group B) of workgroup 0 stamp s_memtime at every phase boundary (tools/gemm_pp_trace.py)
#define PP_STAMP(slot)                                                                        \
  if (TRACE) {                                                                                \
    if (trace && blockIdx.x == 0 && blockIdx.y == 0 && wl == 0 && lane == 0 && t < 64)        \
      trace[(grp * 64 + t) * 8 + (slot)] = (long long)__builtin_amdgcn_s_memtime();           \
  }

template <int NST, bool TRACE, int SCHED>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                         void* Cv, const float* __restrict__ bias, const float* res,
                                                         int M, int N, int K, int lda, int ldb, int ldc, int ldr,
                                                         int flags, float alpha, int tiles_m, int kt_per_split,
                                                         long split_stride, long long* trace) {
  static_assert(NST == 3, "the in-MFMA prefetch schedule (A: tile t+2, B: tile t+3) needs exactly 3 stages");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [NST stages][A 32K | B 16K]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wl = wave & 3;
  const int wm = wl >> 1, wn = wl & 1;
  const int lr = lane & 15, lg = lane >> 4;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = lid % tiles_m, tn = lid / tiles_m;
  const int m0 = tm * PP_BM, n0 = tn * PP_BN;

  // staging shares: A tile 2048 chunks -> 4 per thread, B tile 1024 chunks -> 2 per thread
  const bf16_t* gA[4];
  const bf16_t* gB[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 512 + tid;
    const int row = c >> 3, lc = (c & 7) ^ (row & 7);
    int ra = m0 + row;
    ra = ra < M ? ra : M - 1;
    gA[i] = A + (size_t)ra * lda + lc * 8;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * 512 + tid;
    const int row = c >> 3, lc = (c & 7) ^ (row & 7);
    int rb = n0 + row;
    rb = rb < N ? rb : N - 1;
    gB[i] = B + (size_t)rb * ldb + lc * 8;
  }

  float4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  const int nt_all = K / PP_BK;
  const int kt0 = blockIdx.y * kt_per_split;
  const int nt = (nt_all - kt0) < kt_per_split ? (nt_all - kt0) : kt_per_split;
  if (gridDim.y > 1) Cv = reinterpret_cast<float*>(Cv) + blockIdx.y * split_stride;

  // one 1-KiB LDS-DMA instruction of this wave's share of tile t: idx 0-3 = A chunks, 4-5 = B chunks
  auto issue_one = [&](int t, int idx) {
    char* sA = smem + (t % NST) * PP_STAGE;
    char* sB = sA + PP_A_BYTES;
    const int k0 = (kt0 + t) * PP_BK;
    if (idx < 4)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(gA[idx] + k0), (lds_void_t*)(sA + (idx * 512 + wave * 64) * 16), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(gB[idx - 4] + k0), (lds_void_t*)(sB + ((idx - 4) * 512 + wave * 64) * 16), 16, 0, 0);
  };
  auto issue = [&](int t) {
#pragma unroll
    for (int idx = 0; idx < 6; ++idx) issue_one(t, idx);
  };

#pragma unroll
  for (int t = 0; t < NST; ++t)
    if (t < nt) issue(t);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                       // #0: tiles 0 .. NST-1 visible
  if (grp == 1) __builtin_amdgcn_s_barrier();         // B starts one phase late

  const int arow = grp * 128 + wm * 64;               // this wave's first A row inside the 256-row tile
  short8_t af[2][4], bfr[2][4];
  for (int t = 0; t < nt; ++t) {
    // ------------------------------------------------------------------ READ(t)
    PP_STAMP(0)
    // both groups need tile t+1 at their next READ; the share of tile t+2 (6 loads) may stay in flight
    const bool younger = (t + 2 < nt);
    const bool pf_in_read = (SCHED == 1) && t >= 1 && younger;   // SCHED 1: both groups prefetch tile t+2 from R(t)
    {
      const char* sA = smem + (t % NST) * PP_STAGE;
      const char* sB = sA + PP_A_BYTES;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          bfr[kk][j] = *reinterpret_cast<const short8_t*>(sB + pp_swz(wn * 64 + j * 16 + lr, kk * 4 + lg));
#pragma unroll
        for (int i = 0; i < 4; ++i)
          af[kk][i] = *reinterpret_cast<const short8_t*>(sA + pp_swz(arow + i * 16 + lr, kk * 4 + lg));
      }
    }
    if (pf_in_read) issue(t + 2);   // behind the fragment reads: the DMA issue overlaps the LDS read latency
    PP_STAMP(1)
    if (grp == 1) {                                                  // B's share of tile t+1
      if (younger) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP_STAMP(2)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    PP_STAMP(3)
    // ------------------------------------------------------------------ MFMA(t)
    // The prefetch is issued from INSIDE the MFMA phase, one LDS-DMA instruction per 5-6 MFMAs: a burst of 6 in
    // front of the MFMAs stalls the wave ~700 cycles on the CU's 64 B/clk vector-memory path with the matrix pipe
    // idle (measured, tools/gemm_pp_trace.py).  A prefetches tile t+2 (its stage was last read in B's R(t-1), which
    // ended at the barrier before this phase), B prefetches tile t+3 (stage last read in B's own R(t)).
    const int pf = grp == 0 ? t + 2 : t + 3;
    const bool do_pf = (SCHED == 0) && (pf >= NST) && (pf < nt);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
        const int q = kk * 4 + i;                       // 8 groups of 4 MFMAs; loads after groups 0..5
        if (SCHED == 0 && q < 6) {
          __builtin_amdgcn_sched_barrier(0);
          if (do_pf) issue_one(pf, q);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    __builtin_amdgcn_s_setprio(0);
    PP_STAMP(4)
    if (grp == 0) {                                                  // A's share of tile t+1
      if (younger) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PP_STAMP(5)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    PP_STAMP(6)
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();         // balance B's extra start barrier

  // epilogue (same lane ownership as the 4-wave kernel): C[m][n..n+3], m = m0+arow+i*16+lr, n = n0+wn*64+j*16+lg*4
  const bool out_f32 = flags & MH_GEMM_OUT_F32;
  const bool do_gelu = flags & MH_GEMM_GELU;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + arow + i * 16 + lr;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + lg * 4;
      if (n >= N) continue;
      float v[4] = {acc[i][j][0] * alpha, acc[i][j][1] * alpha, acc[i][j][2] * alpha, acc[i][j][3] * alpha};
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
          *reinterpret_cast<float4_t*>(reinterpret_cast<float*>(Cv) + (size_t)m * ldc + n) =
              (float4_t){v[0], v[1], v[2], v[3]};
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

static int g_sched = 1;
extern "C" void mhdbg_set_gemm_sched(int v) { g_sched = v; }                  // debug hook: A/B of the prefetch placement
static long long* g_trace = nullptr;
extern "C" void mhdbg_set_gemm_trace(void* p) { g_trace = (long long*)p; }   // debug hook, not part of the ABI

int mh_launch_gemm_pp(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                      const float* bias, const float* residual, int ldr, int flags, float alpha, int splits, int tps,
                      long split_stride, hipStream_t stream) {
  constexpr int NST = 3;
  const int tiles_m = (M + PP_BM - 1) / PP_BM, tiles_n = (N + PP_BN - 1) / PP_BN;
  const size_t shmem = NST * PP_STAGE;   // 144 KiB of the CU's 160 -> one workgroup (8 waves) per CU
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<NST, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<NST, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<NST, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<NST, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    attr_set = true;
  }
  const dim3 grid(tiles_m * tiles_n, splits), block(512);
#define PP_LAUNCH(TR, SC)                                                                                              \
  hipLaunchKernelGGL((gemm_pp_kernel<NST, TR, SC>), grid, block, shmem, stream, (const bf16_t*)A, (const bf16_t*)B, C, \
                     bias, residual, M, N, K, lda, ldb, ldc, ldr, flags, alpha, tiles_m, tps, split_stride, g_trace)
  if (g_trace) {
    if (g_sched == 0) PP_LAUNCH(true, 0); else PP_LAUNCH(true, 1);
  } else {
    if (g_sched == 0) PP_LAUNCH(false, 0); else PP_LAUNCH(false, 1);
  }
#undef PP_LAUNCH
  MH_CHECK_LAUNCH();
  return MH_OK;
}

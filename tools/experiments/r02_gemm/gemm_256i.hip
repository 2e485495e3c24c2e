// K1 (large-tile variant, interleaved schedule): the 256 x 256 x 32 block tile, 8 waves and 4-deep LDS-DMA ring of
// gemm_256.hip with a different main loop.
//
// Why (round-2 measurement, profiles/r02_gemm_256i.md): in gemm_256.hip a wave alternates a fragment-read phase
// (12 ds_read_b128 + 4 LDS-DMA issues + the wait for the last read, ~650 cycles) with a 32-MFMA phase (~550 cycles),
// and the two waves of a SIMD run those phases in anti-phase.  The interval is max(reads_A, mfma_B) + max(mfma_A,
// reads_B) ~ 1400 cycles for 64 MFMAs of 16-17 cycles each: the matrix pipe idles while the longer read phase
// finishes (shortening group B's MFMA block by 1/8 changed nothing).  Here every wave runs ONE stream per tile:
//   interval t:   32 MFMAs on the fragments of tile t (register set t & 1)
//                 with, in their issue shadows, the 12 ds_read_b128 of tile t+1 (into set (t+1) & 1) and the four
//                 LDS-DMA instructions of tile t+4 (into the stage tile t was read from one interval earlier)
//                 s_waitcnt lgkmcnt(0), vmcnt(8) [tile t+2 landed; t+3, t+4 in flight]; s_barrier
// A v_mfma_f32_16x16x32_bf16 occupies the SIMD's matrix pipe for ~16 cycles and the wave's issue port for 4: the
// reads and DMA issues are single-issue fillers between MFMAs (MI355X_MICROARCH.md: <= 5 fillers hide per 32-cycle
// gap), so a wave's stream is MFMA-paced and the two waves of a SIMD simply share the pipe.  Fragments are double
// buffered in registers (2 x 48 VGPRs; 128 accumulators + 96 + addressing < 256).
//
// Ring discipline (4 stages of 32 KiB, tile t lives in stage t & 3):
//   RAW  tile t+1 is read in interval t; its DMA shares were waited for (counted vmcnt) before the barrier that ends
//        interval t-1, by every wave, so every share has landed when any wave passes that barrier;
//   WAR  tile t+4 is written into stage t & 3 during interval t; the reads of tile t (interval t-1) were retired by
//        lgkmcnt(0) before the same barrier.
#include "gemm_256_common.h"

template <int MB>
__global__ __launch_bounds__(512, 2) void gemm_256i_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                           void* Cv, const float* __restrict__ bias, const float* res,
                                                           int M, int N, int K, int lda, int ldb, int ldc, int ldr,
                                                           int flags, float alpha, int tiles_m, int kt_per_split,
                                                           long split_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [4 stages][A 16K | B 16K]; reused by the epilogue
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int lr = lane & 15, lg = lane >> 4;

  // tile order: as gemm_256.hip (XCD-contiguous, 8 x 4 patches per XCD)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tiles_n = nwg / tiles_m;
  const int per_group = 8 * tiles_n;
  const int first_m = (lid / per_group) * 8;
  const int gsz = (tiles_m - first_m) < 8 ? (tiles_m - first_m) : 8;
  const int tm = first_m + (lid % per_group) % gsz, tn = (lid % per_group) / gsz;
  constexpr int BMT = 128 + 16 * MB;
  const int m0 = tm * BMT, n0 = tn * G2_BN;
  const int m_end = (m0 + BMT) < M ? (m0 + BMT) : M;

  // staging shares: 1024 chunks of 16 B per operand tile -> 2 + 2 LDS-DMA instructions per thread per tile.  Per-lane
  // 32-bit byte offsets against the wave-uniform operand bases (the K advance is a scalar add): 4 VGPRs, not 8
  unsigned vA[2], vB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * 512 + tid;
    const int row = c >> 2, lc = (c & 3) ^ g2_perm(row);
    int ra = m0 + row, rb = n0 + row;
    ra = ra < M ? ra : M - 1;
    rb = rb < N ? rb : N - 1;
    vA[i] = ((unsigned)ra * (unsigned)lda + lc * 8) * 2u;
    vB[i] = ((unsigned)rb * (unsigned)ldb + lc * 8) * 2u;
  }

  float4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  const int nt_all = K / G2_BK;
  const int kt0 = blockIdx.y * kt_per_split;
  const int nt = (nt_all - kt0) < kt_per_split ? (nt_all - kt0) : kt_per_split;
  if (gridDim.y > 1) Cv = reinterpret_cast<float*>(Cv) + blockIdx.y * split_stride;

  auto issue = [&](int t) {
    char* sA = smem + (t & 3) * G2_STAGE;
    char* sB = sA + G2_A_BYTES;
    const char* bA = reinterpret_cast<const char*>(A + (kt0 + t) * G2_BK);   // wave-uniform
    const char* bB = reinterpret_cast<const char*>(B + (kt0 + t) * G2_BK);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(bA + vA[i]), (lds_void_t*)(sA + (i * 512 + wave * 64) * 16), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(bB + vB[i]), (lds_void_t*)(sB + (i * 512 + wave * 64) * 16), 16, 0, 0);
  };

  // prologue: four tiles in flight; tiles 0 and 1 must be visible before the loop (tile 0 is read here, tile 1 in interval 0)
#pragma unroll
  for (int t = 0; t < G2_NST; ++t)
    if (t < nt) issue(t);
  {
    const int last = (nt - 1) < 3 ? (nt - 1) : 3;      // youngest tile issued
    g2_wait_younger(last - 1 > 0 ? last - 1 : 0);      // everything up to tile 1 has landed
  }
  __builtin_amdgcn_s_barrier();

  int offA[8], offB[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) offA[i] = g2_off(grp * 128 + i * 16 + lr, lg);
#pragma unroll
  for (int j = 0; j < 4; ++j) offB[j] = G2_A_BYTES + g2_off(wc * 64 + j * 16 + lr, lg);

  short8_t fa0[8], fb0[4], fa1[8], fb1[4];

  // NI = this wave's row fragments (8 for group A, MB for group B): a compile-time count per code path keeps each
  // interval a single basic block, which the instruction-group directives below need
  auto run = [&](auto ni_tag) {
    constexpr int NI = decltype(ni_tag)::value;
    auto reads = [&](short8_t (&fa)[8], short8_t (&fb)[4], int t) {
      const char* st = smem + (t & 3) * G2_STAGE;
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const short8_t*>(st + offB[j]);
#pragma unroll
      for (int i = 0; i < NI; ++i) fa[i] = *reinterpret_cast<const short8_t*>(st + offA[i]);
    };
    auto mfmas = [&](short8_t (&fa)[8], short8_t (&fb)[4]) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    };
    // steady state: t + 4 < nt.  Source order reads, DMA, MFMAs; the group directives interleave them one filler per MFMA
    auto steady = [&](short8_t (&fac)[8], short8_t (&fbc)[4], short8_t (&fan)[8], short8_t (&fbn)[4], int t) {
      reads(fan, fbn, t + 1);
      issue(t + 4);
      mfmas(fac, fbc);
#pragma unroll
      for (int k = 0; k < 4 + NI; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one ds_read
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // two MFMAs
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one LDS-DMA (VMEM read)
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * NI - (4 + NI) - 8, 0);
      __builtin_amdgcn_sched_barrier(0);                   // the waits stay behind the last MFMA
      g2_wait_vm<8>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    };
    // last four intervals: no DMA left (t + 4 >= nt); the last one has nothing to read either
    auto tail = [&](short8_t (&fac)[8], short8_t (&fbc)[4], short8_t (&fan)[8], short8_t (&fbn)[4], int t) {
      if (t + 1 < nt) reads(fan, fbn, t + 1);
      if (t + 4 < nt) issue(t + 4);
      mfmas(fac, fbc);
      const int youngest = (t + 4) < (nt - 1) ? (t + 4) : (nt - 1);
      g2_wait_younger(youngest - (t + 2) > 0 ? youngest - (t + 2) : 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    };

    reads(fa0, fb0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                      // every wave has read tile 0: interval 0 may overwrite stage 0
    int t = 0;
    for (; t + 5 < nt; t += 2) {
      steady(fa0, fb0, fa1, fb1, t);
      steady(fa1, fb1, fa0, fb0, t + 1);
    }
    for (; t < nt; t += 2) {
      tail(fa0, fb0, fa1, fb1, t);
      if (t + 1 < nt) tail(fa1, fb1, fa0, fb0, t + 1);
    }
  };
  if (MB == 8 || grp == 0) run(std::integral_constant<int, 8>{});
  else run(std::integral_constant<int, MB>{});

  g2_epilogue(smem, acc, Cv, bias, res, M, N, ldc, ldr, flags, alpha, m0, n0, m_end, wave, lane);
}

template <int MB>
static int launch_256i(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                       const float* bias, const float* residual, int ldr, int flags, float alpha, int splits, int tps,
                       long split_stride, hipStream_t stream) {
  constexpr int BMT = 128 + 16 * MB;
  const int tiles_m = (M + BMT - 1) / BMT, tiles_n = (N + G2_BN - 1) / G2_BN;
  const size_t shmem = G2_NST * G2_STAGE;   // 128 KiB -> one 8-wave workgroup per CU
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_256i_kernel<MB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    attr_set = true;
  }
  const dim3 grid(tiles_m * tiles_n, splits), block(512);
  hipLaunchKernelGGL((gemm_256i_kernel<MB>), grid, block, shmem, stream, (const bf16_t*)A, (const bf16_t*)B, C, bias,
                     residual, M, N, K, lda, ldb, ldc, ldr, flags, alpha, tiles_m, tps * 2, split_stride);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

int mh_launch_gemm_256i(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                        const float* bias, const float* residual, int ldr, int flags, float alpha, int splits, int tps,
                        long split_stride, int mb, hipStream_t stream) {
  switch (mb) {
    case 5: return launch_256i<5>(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, flags, alpha, splits, tps, split_stride, stream);
    case 6: return launch_256i<6>(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, flags, alpha, splits, tps, split_stride, stream);
    case 7: return launch_256i<7>(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, flags, alpha, splits, tps, split_stride, stream);
    default: return launch_256i<8>(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, flags, alpha, splits, tps, split_stride, stream);
  }
}

// K1/K2: bf16 GEMM  C[M,N] = alpha * A[M,K] . B[N,K]^T (+bias[N]) (GELU) (+residual[M,N] fp32)
//
// Both operands are K-contiguous ("NT" form) -- the layout of every nn.Linear on the hot path
// (y = x W^T, W stored [out,in]; reference eva_vit.py:124,146,55-59, Qformer.py:127-133,281,352,367,
// modeling_llama.py:134-136,159-162,604).  dgrad uses the same kernel on pre-transposed frozen weights
// (W^T stored [in,out]); wgrad uses it on transposed activations.  One kernel, one layout.
//
// gfx950 design: 128 x TBN x 64 block tile (TBN = 128 or 64), 4 waves (2x2), each wave a 64 x TBN/2 sub-tile of
// v_mfma_f32_16x16x32_bf16 fragments (fp32 accumulate).  A/B tiles are staged into LDS by
// global_load_lds_dwordx4 (LDS-DMA, 16 B/lane, no VGPR round trip), double-buffered; the LDS image is
// lane-linear, so the bank-conflict XOR swizzle (16-B chunk ^= row&7) is applied on the per-lane global
// SOURCE address and again on the ds_read_b128 fragment reads.  The MFMA is issued as (B-frag, A-frag) so
// each lane ends up with 4 consecutive N for one M row -> 8/16-byte vector epilogue stores with
// vector bias/residual loads.  Workgroup ids are remapped so each XCD (private L2) owns a contiguous
// run of tiles that share the same weight panel.
//
// Tile-width choice (measured, profiles/): the fine-tune step has M = B*S ~ 1.2-2k rows, so N = 4096 / 1408
// outputs give only 187-320 128x128 tiles for 256 CUs x 2 resident blocks; those shapes run at 360-650 TF/s while
// >= 512-tile shapes reach 780-840.  TBN = 64 doubles the tile count (3 resident blocks per CU at 48 KiB LDS) so
// the per-k-step vmcnt(0)+barrier drain of one block is covered by its neighbours.  Row fragments that lie
// entirely beyond M (the ragged last M tile) skip their ds_reads and MFMAs.
#include "common.h"

#define BM 128
#define BK 64
#define A_STAGE_BYTES (BM * BK * 2)  // 16 KiB

#define MH_GEMM_OUT_F32 1
#define MH_GEMM_GELU 2
#define MH_GEMM_REGSTAGE 4

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__device__ __forceinline__ int swz_off(int row, int chunk) { return row * (BK * 2) + ((chunk ^ (row & 7)) << 4); }

template <bool GLDS, int TBN>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                         void* Cv, const float* __restrict__ bias, const float* res,
                                                         int M, int N, int K, int lda, int ldb, int ldc, int ldr,
                                                         int flags, float alpha, int tiles_m, int kt_per_split,
                                                         long split_stride) {
  constexpr int NJ = TBN / 32;                 // 16-wide N fragments per wave
  constexpr int NB = TBN / 32;                 // B staging chunks per thread (TBN rows x 8 chunks / 256 threads)
  constexpr int B_STAGE_BYTES = TBN * BK * 2;
  constexpr int STAGE = A_STAGE_BYTES + B_STAGE_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A | B]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lg = lane >> 4;

  // XCD-aware bijective remap: blocks b, b+8, b+16.. land on one XCD -> give them consecutive tiles.
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = lid % tiles_m, tn = lid / tiles_m;
  const int m0 = tm * BM, n0 = tn * TBN;

  const bf16_t* gA[4];
  const bf16_t* gB[NB];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid;
    const int row = c >> 3, lc = (c & 7) ^ (row & 7);
    int ra = m0 + row;
    ra = ra < M ? ra : M - 1;
    gA[i] = A + (size_t)ra * lda + lc * 8;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int c = i * 256 + tid;
    const int row = c >> 3, lc = (c & 7) ^ (row & 7);
    int rb = n0 + row;
    rb = rb < N ? rb : N - 1;
    gB[i] = B + (size_t)rb * ldb + lc * 8;
  }

  float4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  // split-K: blockIdx.y owns K tiles [kt0, kt0+nt) and writes its fp32 partial slab at Cv + y*split_stride
  const int nt_all = K / BK;
  const int kt0 = blockIdx.y * kt_per_split;
  const int nt = (nt_all - kt0) < kt_per_split ? (nt_all - kt0) : kt_per_split;
  if (gridDim.y > 1) Cv = reinterpret_cast<float*>(Cv) + blockIdx.y * split_stride;
  // number of 16-row fragments of this wave that contain at least one valid row (ragged last M tile)
  int vrows = M - (m0 + wm * 64);
  const int ni = vrows >= 64 ? 4 : (vrows <= 0 ? 0 : (vrows + 15) >> 4);

  short8_t ra_[4], rb_[NB];

  auto issue = [&](int t, int buf) {
    char* sA = smem + buf * STAGE;
    char* sB = sA + A_STAGE_BYTES;
    const int k0 = (kt0 + t) * BK;
    if (GLDS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int base = (i * 256 + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(gA[i] + k0), (lds_void_t*)(sA + base), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int base = (i * 256 + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(gB[i] + k0), (lds_void_t*)(sB + base), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) ra_[i] = *reinterpret_cast<const short8_t*>(gA[i] + k0);
#pragma unroll
      for (int i = 0; i < NB; ++i) rb_[i] = *reinterpret_cast<const short8_t*>(gB[i] + k0);
    }
  };
  auto commit = [&](int buf) {  // register-staged path only
    char* sA = smem + buf * STAGE;
    char* sB = sA + A_STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<short8_t*>(sA + (i * 256 + tid) * 16) = ra_[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<short8_t*>(sB + (i * 256 + tid) * 16) = rb_[i];
  };

  issue(0, 0);
  if (GLDS) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    commit(0);
  }
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) issue(t + 1, buf ^ 1);
    const char* sA = smem + buf * STAGE;
    const char* sB = sA + A_STAGE_BYTES;
    if (ni == 4) {   // common case: straight-line, fully unrolled (no predication in the MFMA stream)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        short8_t af[4], bfr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          bfr[j] = *reinterpret_cast<const short8_t*>(sB + swz_off(wn * (TBN / 2) + j * 16 + lr, kk * 4 + lg));
#pragma unroll
        for (int i = 0; i < 4; ++i)
          af[i] = *reinterpret_cast<const short8_t*>(sA + swz_off(wm * 64 + i * 16 + lr, kk * 4 + lg));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
    } else if (ni > 0) {   // ragged last M tile: only the fragments that hold valid rows
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        short8_t af[4], bfr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          bfr[j] = *reinterpret_cast<const short8_t*>(sB + swz_off(wn * (TBN / 2) + j * 16 + lr, kk * 4 + lg));
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < ni) af[i] = *reinterpret_cast<const short8_t*>(sA + swz_off(wm * 64 + i * 16 + lr, kk * 4 + lg));
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < ni) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
          }
      }
    }
    if (GLDS) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (t + 1 < nt) commit(buf ^ 1);
    }
    __syncthreads();
  }

  // epilogue: lane owns C[m][n .. n+3] for m = m0+wm*64+i*16+lr, n = n0+wn*(TBN/2)+j*16+lg*4
  const bool out_f32 = flags & MH_GEMM_OUT_F32;
  const bool do_gelu = flags & MH_GEMM_GELU;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + lr;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n0 + wn * (TBN / 2) + j * 16 + lg * 4;
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

template <bool GLDS, int TBN>
static int launch_gemm(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                       const float* bias, const float* residual, int ldr, int flags, float alpha, int splits, int tps,
                       long split_stride, hipStream_t stream) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + TBN - 1) / TBN;
  const dim3 grid(tiles_m * tiles_n, splits), block(256);
  const size_t shmem = 2 * (A_STAGE_BYTES + TBN * BK * 2);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<GLDS, TBN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)shmem);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_nt_kernel<GLDS, TBN>), grid, block, shmem, stream, (const bf16_t*)A, (const bf16_t*)B, C,
                     bias, residual, M, N, K, lda, ldb, ldc, ldr, flags, alpha, tiles_m, tps, split_stride);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// 128-wide N tiles when they already give >= ~1.75 resident blocks per CU, else 64-wide
static inline bool use_narrow(int M, int N) {
  const long tiles128 = (long)((M + BM - 1) / BM) * ((N + 127) / 128);
  return tiles128 < 448;
}

extern "C" int mh_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                               const float* bias, const float* residual, int ldr, int flags, float alpha,
                               hipStream_t stream) {
  if (M <= 0 || N <= 0) return MH_OK;
  if (K <= 0 || (K % BK) != 0 || (lda % 8) != 0 || (ldb % 8) != 0) return MH_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15)) return MH_ERR_ARG;
  if ((ldc % 4) != 0 || (residual && (ldr % 4) != 0)) return MH_ERR_ARG;
  const int nt = K / BK;
  const bool narrow = use_narrow(M, N);
  if (flags & MH_GEMM_REGSTAGE) {
    return narrow ? launch_gemm<false, 64>(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, flags, alpha, 1, nt, 0L, stream)
                  : launch_gemm<false, 128>(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, flags, alpha, 1, nt, 0L, stream);
  }
  return narrow ? launch_gemm<true, 64>(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, flags, alpha, 1, nt, 0L, stream)
                : launch_gemm<true, 128>(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, flags, alpha, 1, nt, 0L, stream);
}

// ---- split-K variant for skinny outputs with a long reduction (wgrad of the conv stem, M,N small, K huge) ----
// partial slabs ws[splits][M][N] fp32, then a fixed-order reduction -> deterministic.
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int M, int N, int ldc,
                                     int splits) {
  const long total = (long)M * N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[(long)k * total + i];
    out[(i / N) * ldc + (i % N)] = s;
  }
}

extern "C" long mh_gemm_splitk_ws_floats(int M, int N, int splits) { return (long)M * N * splits; }

extern "C" int mh_gemm_bf16_nt_splitk(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N,
                                      int K, int splits, float* ws, hipStream_t stream) {
  if (M <= 0 || N <= 0) return MH_OK;
  if (K <= 0 || (K % BK) != 0 || (lda % 8) != 0 || (ldb % 8) != 0 || (N % 4) != 0 || splits < 1) return MH_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)ws & 15)) return MH_ERR_ARG;
  const int nt = K / BK;
  if (splits > nt) splits = nt;
  const int tps = (nt + splits - 1) / splits;
  splits = (nt + tps - 1) / tps;
  int rc = launch_gemm<true, 128>(A, lda, B, ldb, (void*)ws, N, M, N, K, nullptr, nullptr, 0, MH_GEMM_OUT_F32, 1.0f, splits,
                                  tps, (long)M * N, stream);
  if (rc) return rc;
  long g = ((long)M * N + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)g), dim3(256), 0, stream, ws, C, M, N, ldc, splits);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

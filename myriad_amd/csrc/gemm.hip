// K1/K2: bf16 GEMM  C[M,N] = alpha * A[M,K] . B[N,K]^T (+bias[N]) (GELU) (+residual[M,N] fp32)
//
// Both operands are K-contiguous ("NT" form) -- the layout of every nn.Linear on the hot path
// (y = x W^T, W stored [out,in]; reference eva_vit.py:124,146,55-59, Qformer.py:127-133,281,352,367,
// modeling_llama.py:134-136,159-162,604).  dgrad uses the same kernel on pre-transposed frozen weights
// (W^T stored [in,out]); wgrad uses it on transposed activations.  One kernel, one layout.
//
// gfx950 design: 128 x TBN x 64 block tile (TBN = 128 or 64), 4 waves (2x2), each wave a 64 x TBN/2 sub-tile of
// v_mfma_f32_16x16x32_bf16 fragments (fp32 accumulate).  A/B tiles are staged into LDS by
// global_load_lds_dwordx4 (LDS-DMA, 16 B/lane, no VGPR round trip) through an NST-deep LDS ring with COUNTED
// s_waitcnt vmcnt(N) and a raw s_barrier, so NST-1 K-tiles of loads stay in flight across barriers (the 2-deep
// vmcnt(0)+__syncthreads form measured 560-840 TF/s on the step's shapes: one K-step of MFMA work, ~512
// cycles, cannot cover a ~2k-cycle loaded L2/HBM round trip).  The LDS image is lane-linear, so the bank-conflict
// XOR swizzle (16-B chunk ^= row&7) is applied on the per-lane global SOURCE address and again on the
// ds_read_b128 fragment reads.  The MFMA is issued as (B-frag, A-frag) so each lane ends up with 4 consecutive N
// for one M row -> 8/16-byte vector epilogue stores with vector bias/residual loads.  Workgroup ids are remapped so
// each XCD (private L2) owns a contiguous run of tiles that share the same weight panel.
#include <cstdlib>

#include "common.h"

#define BM 128
#define BKMIN 32   // K must be a multiple of 64 at the ABI; kernels use 64- or 32-deep K tiles

#define MH_GEMM_OUT_F32 1
#define MH_GEMM_GELU 2
#define MH_GEMM_REGSTAGE 4
#define MH_GEMM_GELU_PRE 64    // internal (mh_gemm_gelu_fwd): aux <- bf16 pre-activation, C <- gelu of it
#define MH_GEMM_GELU_BWD 128   // internal (mh_gemm_gelu_bwd): C <- bf16(product) * gelu'(aux)
#define MH_GEMM_PACKED_B (1 << 20)   // internal: B is tile-packed, [N / TBN][K / 64][TBN rows][64] (one 16-KiB block per k-tile of a column tile)
#define MH_GEMM_VARIANT_SHIFT 8  // bits 8..11: 0 = auto, 1 = 2-stage/128, 2 = 4-stage/128, 3 = 3-stage/64, 4 = 2-stage/64

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// byte offset of 16-B chunk `chunk` of tile row `row` in the swizzled LDS image (rows are TBK*2 bytes)
template <int TBK>
__device__ __forceinline__ int swz_off(int row, int chunk) {
  if (TBK == 64) return row * 128 + ((chunk ^ (row & 7)) << 4);
  return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);   // 64-B rows: rows r, r+4, r+8, r+12 share banks
}
template <int TBK>
__device__ __forceinline__ int swz_src_chunk(int row, int phys) {
  return TBK == 64 ? (phys ^ (row & 7)) : (phys ^ ((row >> 2) & 3));
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  // s_waitcnt vmcnt(N), other counters untouched: gfx9 encoding vmcnt = simm16[15:14] : simm16[3:0], expcnt [6:4], lgkmcnt [11:8]
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  __builtin_amdgcn_s_waitcnt((N & 15) | 0x70 | 0xF00 | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}

// tile t may be read once at most `behind` younger tiles' loads (LPT LDS-DMA instructions each) are still in flight
template <int LPT, int MAXB>
__device__ __forceinline__ void wait_behind(int behind) {
  if constexpr (MAXB <= 0) {
    wait_vmcnt<0>();
  } else {
    if (behind >= MAXB) wait_vmcnt<MAXB * LPT>();
    else wait_behind<LPT, MAXB - 1>(behind);
  }
}

// STAGING: 0 = LDS-DMA ring (NST stages), 1 = register-staged double buffer (NST must be 2; A/B-test reference)
template <int STAGING, int NST, int TBN, int MINB, int TBK, int NW, int TBM>
__global__ __launch_bounds__(NW * 64, MINB * NW / 4) void gemm_nt_kernel(const bf16_t* __restrict__ A,
                                                            const bf16_t* __restrict__ B, void* Cv,
                                                            const float* __restrict__ bias, const float* res, int M,
                                                            int N, int K, int lda, int ldb, int ldc, int ldr, int flags,
                                                            float alpha, int tiles_m, int kt_per_split,
                                                            long split_stride, void* aux, int ldaux) {
  constexpr int BK = TBK;
  constexpr int CPR = TBK / 8;                 // 16-B chunks per tile row
  constexpr int NJ = TBN / 32;                 // 16-wide N fragments per wave
  constexpr int NT = NW * 64;                  // threads per workgroup
  constexpr int WROWS = NW / 2;                // waves along M (x 2 along N)
  constexpr int MI = TBM / WROWS / 16;         // 16-row M fragments per wave (4 for 4 waves, 2 for 8 waves; 5 for the 160-row tile)
  constexpr int NA = TBM * CPR / NT;           // A staging chunks per thread
  constexpr int NB = TBN * CPR / NT;           // B staging chunks per thread
  constexpr int LPT = NA + NB;                 // LDS-DMA instructions per thread per K tile
  constexpr int A_STAGE_BYTES = TBM * BK * 2;
  constexpr int B_STAGE_BYTES = TBN * BK * 2;
  constexpr int STAGE = A_STAGE_BYTES + B_STAGE_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [NST stages][A | B]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;   // wave grid WROWS x 2
  const int lr = lane & 15, lg = lane >> 4;

  // XCD-aware bijective remap: blocks b, b+8, b+16.. land on one XCD -> give them consecutive tiles.
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = lid % tiles_m, tn = lid / tiles_m;
  const int m0 = tm * TBM, n0 = tn * TBN;

  const bf16_t* gA[NA];
  const bf16_t* gB[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int c = i * NT + tid;
    const int row = c / CPR, lc = swz_src_chunk<TBK>(row, c % CPR);
    int ra = m0 + row;
    ra = ra < M ? ra : M - 1;
    gA[i] = A + (size_t)ra * lda + lc * 8;
  }
  const bool packed_b = flags & MH_GEMM_PACKED_B;
  const int kmul_b = packed_b ? TBN : 1;         // element step of B per k element: a packed k-tile is TBN * 64 elements away
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int c = i * NT + tid;
    const int row = c / CPR, lc = swz_src_chunk<TBK>(row, c % CPR);
    int rb = n0 + row;
    rb = rb < N ? rb : N - 1;
    gB[i] = packed_b ? B + (size_t)tn * K * TBN + (size_t)row * BK + lc * 8 : B + (size_t)rb * ldb + lc * 8;
  }

  float4_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  // split-K: blockIdx.y owns K tiles [kt0, kt0+nt) and writes its fp32 partial slab at Cv + y*split_stride
  const int nt_all = K / BK;
  const int kt0 = blockIdx.y * kt_per_split;
  const int nt = (nt_all - kt0) < kt_per_split ? (nt_all - kt0) : kt_per_split;
  if (gridDim.y > 1)   // the partial slab of this split: fp32, or bf16 when the launch asks for bf16 output (run_splitk)
    Cv = reinterpret_cast<char*>(Cv) + blockIdx.y * split_stride * ((flags & MH_GEMM_OUT_F32) ? 4 : 2);

  auto issue_dma = [&](int t, int stage) {
    char* sA = smem + stage * STAGE;
    char* sB = sA + A_STAGE_BYTES;
    const int k0 = (kt0 + t) * BK;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(gA[i] + k0), (lds_void_t*)(sA + (i * NT + wave * 64) * 16), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NB; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(gB[i] + (size_t)k0 * kmul_b), (lds_void_t*)(sB + (i * NT + wave * 64) * 16), 16, 0, 0);
  };
  auto compute = [&](int stage) {
    const char* sA = smem + stage * STAGE;
    const char* sB = sA + A_STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < TBK / 32; ++kk) {
      short8_t af[MI], bfr[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        bfr[j] = *reinterpret_cast<const short8_t*>(sB + swz_off<TBK>(wn * (TBN / 2) + j * 16 + lr, kk * 4 + lg));
#pragma unroll
      for (int i = 0; i < MI; ++i)
        af[i] = *reinterpret_cast<const short8_t*>(sA + swz_off<TBK>(wm * (MI * 16) + i * 16 + lr, kk * 4 + lg));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  };

  if (STAGING == 0) {
    // ---- LDS-DMA ring: tiles t+1 .. t+NST-1 are in flight while tile t is multiplied ----
    const int pre = (NST - 1) < nt ? (NST - 1) : nt;
    for (int t = 0; t < pre; ++t) issue_dma(t, t);
    int stage = 0;
    for (int t = 0; t < nt; ++t) {
      // loads issued after tile t's own: LPT * (tiles in flight behind it)
      const int issued = (t + NST - 1) < nt ? (t + NST - 1) : nt;
      const int behind = issued - (t + 1);
      wait_behind<LPT, NST - 2>(behind);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // tile t visible to all waves; everyone is done reading stage (t-1)%NST
      if (t + NST - 1 < nt) {
        int ps = stage + NST - 1;
        ps = ps >= NST ? ps - NST : ps;
        issue_dma(t + NST - 1, ps);
      }
      compute(stage);
      stage = (stage + 1 == NST) ? 0 : stage + 1;
    }
  } else {
    // ---- register-staged double buffer (reference structure for A/B tests) ----
    short8_t ra_[NA], rb_[NB];
    auto load_regs = [&](int t) {
      const int k0 = (kt0 + t) * BK;
#pragma unroll
      for (int i = 0; i < NA; ++i) ra_[i] = *reinterpret_cast<const short8_t*>(gA[i] + k0);
#pragma unroll
      for (int i = 0; i < NB; ++i) rb_[i] = *reinterpret_cast<const short8_t*>(gB[i] + k0);
    };
    auto commit = [&](int buf) {
      char* sA = smem + buf * STAGE;
      char* sB = sA + A_STAGE_BYTES;
#pragma unroll
      for (int i = 0; i < NA; ++i) *reinterpret_cast<short8_t*>(sA + (i * NT + tid) * 16) = ra_[i];
#pragma unroll
      for (int i = 0; i < NB; ++i) *reinterpret_cast<short8_t*>(sB + (i * NT + tid) * 16) = rb_[i];
    };
    load_regs(0);
    commit(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
      const int buf = t & 1;
      if (t + 1 < nt) load_regs(t + 1);
      compute(buf);
      if (t + 1 < nt) commit(buf ^ 1);
      __syncthreads();
    }
  }

  // epilogue: lane owns C[m][n .. n+3] for m = m0+wm*64+i*16+lr, n = n0+wn*(TBN/2)+j*16+lg*4
  const bool out_f32 = flags & MH_GEMM_OUT_F32;
  const bool do_gelu = flags & MH_GEMM_GELU;
  // GELU pair around an MLP (mh_gemm_gelu_fwd / _bwd): the pre-activation is rounded to bf16 and kept in `aux` (forward), the
  // incoming gradient is rounded to bf16 and multiplied by gelu'(aux) (backward) -- the bits of GEMM -> gelu_fwd / gelu_bwd
  const bool gelu_pre = flags & MH_GEMM_GELU_PRE, gelu_bwd = flags & MH_GEMM_GELU_BWD;
  bf16_t* auxp = reinterpret_cast<bf16_t*>(aux);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * (MI * 16) + i * 16 + lr;
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
        if (gelu_pre) {
          uint2 pk;
          pk.x = pack_bf2(v[0], v[1]);
          pk.y = pack_bf2(v[2], v[3]);
          *reinterpret_cast<uint2*>(auxp + (size_t)m * ldaux + n) = pk;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(bf2f(f2bf(v[e])));
        }
        if (gelu_bwd) {
          const short4_t x4 = *reinterpret_cast<const short4_t*>(auxp + (size_t)m * ldaux + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = bf2f(f2bf(v[e])) * gelu_erf_grad(bf2f((bf16_t)x4[e]));
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
          if (gelu_pre) {
            auxp[(size_t)m * ldaux + n + e] = f2bf(x);
            x = gelu_erf(bf2f(f2bf(x)));
          }
          if (gelu_bwd) x = bf2f(f2bf(x)) * gelu_erf_grad(bf2f(auxp[(size_t)m * ldaux + n + e]));
          if (do_gelu) x = gelu_erf(x);
          if (res) x += res[(size_t)m * ldr + n + e];
          if (out_f32) reinterpret_cast<float*>(Cv)[(size_t)m * ldc + n + e] = x;
          else reinterpret_cast<bf16_t*>(Cv)[(size_t)m * ldc + n + e] = f2bf(x);
        }
      }
    }
  }
}

struct GemmArgs {
  const void* A; int lda; const void* B; int ldb; void* C; int ldc; int M, N, K;
  const float* bias; const float* residual; int ldr; int flags; float alpha; int splits, tps; long split_stride;
  void* aux; int ldaux;        // MH_GEMM_GELU_PRE / _BWD operand (NULL otherwise)
};

template <int STAGING, int NST, int TBN, int MINB, int TBK = 64, int NW = 4, int TBM = BM>
static int launch_gemm(const GemmArgs& g, hipStream_t stream) {
  const int tiles_m = (g.M + TBM - 1) / TBM, tiles_n = (g.N + TBN - 1) / TBN;
  const dim3 grid(tiles_m * tiles_n, g.splits), block(NW * 64);
  const size_t shmem = (size_t)NST * (TBM + TBN) * TBK * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<STAGING, NST, TBN, MINB, TBK, NW, TBM>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    attr_set = true;
  }
  if (g_mh_prof_on) mh_prof_pre(stream, TBM == 160 ? (TBN == 96 ? 5 : 4) : (TBM == 64 ? 6 : (TBN == 64 ? 3 : 1)), g.M, g.N, g.K, g.splits, g.flags);
  // tps / kt_per_split are in units of 64-deep K tiles at the call sites; rescale for 32-deep kernels
  hipLaunchKernelGGL((gemm_nt_kernel<STAGING, NST, TBN, MINB, TBK, NW, TBM>), grid, block, shmem, stream, (const bf16_t*)g.A,
                     (const bf16_t*)g.B, g.C, g.bias, g.residual, g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr, g.flags,
                     g.alpha, tiles_m, g.tps * (64 / TBK), g.split_stride, g.aux, g.ldaux);
  if (g_mh_prof_on) mh_prof_post(stream);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

int mh_launch_gemm_256(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                       const float* bias, const float* residual, int ldr, int flags, float alpha, int splits, int tps,
                       long split_stride, hipStream_t stream, void* aux = nullptr, int ldaux = 0);

static int dispatch(const GemmArgs& g, hipStream_t stream) {
  int variant = (g.flags >> MH_GEMM_VARIANT_SHIFT) & 15;
  if (g.flags & MH_GEMM_REGSTAGE) return launch_gemm<1, 2, 128, 2>(g, stream);
  if (variant == 0) variant = 1;   // measured best on every shape of the step (profiles/r01_gemm_variants.md)
  switch (variant) {
    case 1: return launch_gemm<0, 2, 128, 2>(g, stream);   // 64 KiB, 2 blocks/CU
    case 2: return launch_gemm<0, 4, 128, 1>(g, stream);   // 128 KiB, 1 block/CU, 3 tiles in flight
    case 3: return launch_gemm<0, 3, 64, 2>(g, stream);    // 72 KiB, 2 blocks/CU, 2 tiles in flight each
    case 4: return launch_gemm<0, 2, 64, 3>(g, stream);    // 48 KiB, 3 blocks/CU
    case 5: return launch_gemm<0, 3, 128, 1>(g, stream);   // 96 KiB, 1 block/CU
    case 6: return launch_gemm<0, 2, 128, 4, 32>(g, stream);   // BK=32: 32 KiB, 4 blocks/CU
    case 7: return launch_gemm<0, 3, 128, 3, 32>(g, stream);   // BK=32: 48 KiB, 3 blocks/CU, 2 tiles in flight
    case 8: return launch_gemm<0, 4, 128, 2, 32>(g, stream);   // BK=32: 64 KiB, 2 blocks/CU, 3 tiles in flight
    case 9: return launch_gemm<0, 2, 128, 2, 64, 8>(g, stream);   // 8 waves/block (wave tile 32x64), 2 blocks/CU
    case 10: return launch_gemm<0, 4, 128, 1, 64, 8>(g, stream);  // 8 waves/block, 4-deep ring, 1 block/CU
    case 11: return launch_gemm<0, 8, 64, 1, 64, 4, 64>(g, stream);     // 64x64 tile, 8-deep ring (128 KiB): 7 K tiles in flight, 1 block/CU
    case 13: return launch_gemm<0, 6, 64, 1>(g, stream);          // 128x64, 6-deep ring (144 KiB): 5 K tiles in flight, 1 block/CU
    case 14: return launch_gemm<0, 4, 128, 1, 64, 4, 160>(g, stream);   // 160x128 tile, 4-deep ring (144 KiB): the M <= 160 rows of the
    case 15: return launch_gemm<0, 4, 96, 1, 64, 4, 160>(g, stream);    // batch-1 step as ONE row tile, 160x96 (128 KiB) when N / 96 fills the chip
    case 12:                                                      // 256x256x32, 4-deep ring (gemm_256.hip), 1 block/CU
      return mh_launch_gemm_256(g.A, g.lda, g.B, g.ldb, g.C, g.ldc, g.M, g.N, g.K, g.bias, g.residual, g.ldr, g.flags,
                                g.alpha, g.splits, g.tps, g.split_stride, stream);
    default: return MH_ERR_ARG;
  }
}

// ---- split-K: partial fp32 slabs ws[splits][M][N], then a fixed-order reduction that applies the epilogue ----
// Used (a) for skinny wgrad outputs with a huge reduction and (b) automatically for shapes whose 128x128 tile
// count leaves the 256 CUs under-filled (N = 4096 dgrad / down-proj at M ~ 1.2k: 320 tiles; ViT N = 1408: 187 tiles):
// measured cold-weight gains +9..25 % including the reduce pass (profiles/r01_gemm_splitk.md).
template <int sbf>
__global__ void splitk_reduce_kernel(const void* __restrict__ ws, void* __restrict__ outv, const float* __restrict__ bias,
                                     const float* res, int M, int N, int ldc, int ldr, int splits, int flags,
                                     float alpha) {
  const long total4 = (long)M * N / 4;
  const long slab = (long)M * N;
  const bool out_f32 = flags & MH_GEMM_OUT_F32;
  const bool do_gelu = flags & MH_GEMM_GELU;
  for (long i4 = blockIdx.x * (long)blockDim.x + threadIdx.x; i4 < total4; i4 += (long)gridDim.x * blockDim.x) {
    const long i = i4 * 4;
    float4_t s = slab_load4(ws, i, sbf);
    // slabs in ascending order, eight loads in flight at a time: a wgrad with a long reduction has up to 196 slabs of a tiny output,
    // and one load latency per slab made that launch 100+ us (the batch-1 step's 7-11 slabs: 6 -> 4 us)
    int k = 1;
    if (splits > 4)
    for (; k + 8 <= splits; k += 8) {
      float4_t p[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] = slab_load4(ws, (long)(k + u) * slab + i, sbf);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s[0] += p[u][0]; s[1] += p[u][1]; s[2] += p[u][2]; s[3] += p[u][3]; }
    }
    for (; k < splits; ++k) {
      const float4_t p = slab_load4(ws, (long)k * slab + i, sbf);
      s[0] += p[0]; s[1] += p[1]; s[2] += p[2]; s[3] += p[3];
    }
    const long m = i / N;
    const int n = (int)(i - m * N);
    float v[4] = {s[0] * alpha, s[1] * alpha, s[2] * alpha, s[3] * alpha};
    if (bias) {
      const float4_t b4 = *reinterpret_cast<const float4_t*>(bias + n);
      v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3];
    }
    if (do_gelu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
    }
    if (res) {
      const float4_t r4 = *reinterpret_cast<const float4_t*>(res + m * ldr + n);
      v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3];
    }
    if (out_f32) {
      *reinterpret_cast<float4_t*>(reinterpret_cast<float*>(outv) + m * ldc + n) = (float4_t){v[0], v[1], v[2], v[3]};
    } else {
      uint2 pk;
      pk.x = pack_bf2(v[0], v[1]);
      pk.y = pack_bf2(v[2], v[3]);
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(outv) + m * ldc + n) = pk;
    }
  }
}

// split-K reduce fused with the RMSNorm that consumes the result (o_proj -> post-attention norm, down_proj -> next
// layer's input norm; modeling_llama.py:66-74, 281-293): one workgroup per row sums the slabs in the fixed order, adds the
// fp32 residual, writes the new residual stream AND its normalised bf16 copy.  Same expressions, same summation order
// and the same 256-thread block reduction as splitk_reduce_kernel followed by rmsnorm_fwd_kernel, so the pair of outputs
// is bit-identical to the two-launch form it replaces (one launch and one read of the fp32 stream less per use).
// NORM 0: RMSNorm (y = w * h * rsqrt(mean(h^2) + eps));  NORM 1: LayerNorm (y = (h - mean) * rsqrt(var + eps) * w + nb,
// eva_vit.py:175-179 / ImageBind transformer.py:160-163) -- each written exactly as norm.hip writes it.
// Round 6: templated on the float4 chunks a thread holds (N <= 4096: 4, else 8) and every load of the row requested up front.
// With the run-time chunk loop each chunk was a memory round trip of its own (the store of `hout`, which may alias `res`, kept
// the next chunk's loads behind it): four dependent trips per row.  Same expressions in the same order: same bits.
template <int NORM, int sbf, int NIT>
__global__ __launch_bounds__(256, NIT <= 4 ? 5 : 2) void splitk_reduce_norm_kernel(const void* __restrict__ ws, const float* __restrict__ bias,
                                                                 const float* res, float* hout, const float* __restrict__ w,
                                                                 const float* __restrict__ nb, bf16_t* __restrict__ y, int N,
                                                                 long ldr, long ldh, long ldy, long slab, int splits,
                                                                 float eps) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  float4_t hv[NIT];                                 // N <= 1024 * NIT
  float4_t rv[NIT];
  float acc1 = 0.f;                                 // sum of squares (RMS) or plain sum (LayerNorm)
  if (res) {
#pragma unroll
    for (int c = 0; c < NIT; ++c) {
      const int i = threadIdx.x * 4 + c * 1024;
      if (i < N) rv[c] = *reinterpret_cast<const float4_t*>(res + row * ldr + i);
    }
  }
#pragma unroll
  for (int c = 0; c < NIT; ++c) {
    const int i = threadIdx.x * 4 + c * 1024;
    if (i < N) {
      float4_t sacc = slab_load4(ws, row * N + i, sbf);
      int k = 1;
      if (splits > 4)                                 // (the batch-8 step's 3-slab launches keep the plain loop: measured faster)
      for (; k + 4 <= splits; k += 4) {               // ascending order, four loads in flight (the batch-1 step's 8-11 slabs)
        float4_t p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = slab_load4(ws, (long)(k + u) * slab + row * N + i, sbf);
#pragma unroll
        for (int u = 0; u < 4; ++u) { sacc[0] += p[u][0]; sacc[1] += p[u][1]; sacc[2] += p[u][2]; sacc[3] += p[u][3]; }
      }
      for (; k < splits; ++k) {
        const float4_t p = slab_load4(ws, (long)k * slab + row * N + i, sbf);
        sacc[0] += p[0]; sacc[1] += p[1]; sacc[2] += p[2]; sacc[3] += p[3];
      }
      hv[c] = sacc;
    }
  }
#pragma unroll
  for (int c = 0; c < NIT; ++c) {
    const int i = threadIdx.x * 4 + c * 1024;
    if (i < N) {
      float v[4] = {hv[c][0] * 1.0f, hv[c][1] * 1.0f, hv[c][2] * 1.0f, hv[c][3] * 1.0f};
      if (bias) {
        const float4_t b4 = *reinterpret_cast<const float4_t*>(bias + i);
        v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3];
      }
      if (res) { v[0] += rv[c][0]; v[1] += rv[c][1]; v[2] += rv[c][2]; v[3] += rv[c][3]; }
      hv[c] = (float4_t){v[0], v[1], v[2], v[3]};
      *reinterpret_cast<float4_t*>(hout + row * ldh + i) = hv[c];
      if (NORM == 0) acc1 += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
      else acc1 += v[0] + v[1] + v[2] + v[3];
    }
  }
  float mean = 0.f, r;
  if (NORM == 0) {
    r = rsqrtf(block_sum<4>(acc1, red) / N + eps);
  } else {
    mean = block_sum<4>(acc1, red) / N;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NIT; ++c) {
      const int i = threadIdx.x * 4 + c * 1024;
      if (i < N) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ss += (hv[c][e] - mean) * (hv[c][e] - mean);
      }
    }
    r = rsqrtf(block_sum<4>(ss, red) / N + eps);
  }
#pragma unroll
  for (int c = 0; c < NIT; ++c) {
    const int i = threadIdx.x * 4 + c * 1024;
    if (i < N) {
      const float4_t g = *reinterpret_cast<const float4_t*>(w + i);
      float o[4];
      if (NORM == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = g[e] * (hv[c][e] * r);
      } else {
        const float4_t bb = *reinterpret_cast<const float4_t*>(nb + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (hv[c][e] - mean) * r * g[e] + bb[e];
      }
      uint2 pk;
      pk.x = pack_bf2(o[0], o[1]);
      pk.y = pack_bf2(o[2], o[3]);
      *reinterpret_cast<uint2*>(y + row * ldy + i) = pk;
    }
  }
}

// Slabs of the 256x256 kernel and of the 160-row tiles are bf16 (MYRIAD_SLAB_BF16=0: fp32): these are the K <= 22016 forward / dgrad products of the
// LLaMA and ViT Linears, whose results are rounded to bf16 (or added to the fp32 residual stream) anyway; the partial sums cost
// 2 x 4 B per output element per split in fp32 -- 9 GB per step at batch 8.  The 128x128 kernel's slabs (weight gradients: long
// cancelling reductions over tokens) stay fp32.  *slab_bf16 tells the caller which it got.

static int run_splitk(const GemmArgs& g0, int splits, float* ws, hipStream_t stream, bool reduce = true, int* slab_bf16 = nullptr) {
  const int g_slab_bf16 = mh_opt(MH_OPT_SLAB_BF16);
  const int nt = g0.K / 64;
  if (splits > nt) splits = nt;
  const int tps = (nt + splits - 1) / splits;
  splits = (nt + tps - 1) / tps;
  const int var0 = (g0.flags >> MH_GEMM_VARIANT_SHIFT) & 15;
  const bool big = var0 == 12;
  const bool row_tile = var0 == 14 || var0 == 15;       // the 160-row tiles (batch-1 step): the same forward / dgrad products
  const int sbf = ((big || row_tile) && g_slab_bf16 && (g0.N % 8) == 0) ? 1 : 0;
  if (slab_bf16) *slab_bf16 = sbf;
  GemmArgs g = g0;
  g.C = (void*)ws; g.ldc = g0.N; g.bias = nullptr; g.residual = nullptr; g.ldr = 0;
  g.flags = (sbf ? 0 : MH_GEMM_OUT_F32) | (g0.flags & MH_GEMM_PACKED_B); g.alpha = 1.0f; g.splits = splits; g.tps = tps; g.split_stride = (long)g0.M * g0.N;
  int rc;
  const int var = (g0.flags >> MH_GEMM_VARIANT_SHIFT) & 15;
  if (big)
    rc = mh_launch_gemm_256(g.A, g.lda, g.B, g.ldb, g.C, g.ldc, g.M, g.N, g.K, nullptr, nullptr, 0, g.flags, 1.0f,
                            g.splits, g.tps, g.split_stride, stream);
  else if (var == 14)
    rc = launch_gemm<0, 4, 128, 1, 64, 4, 160>(g, stream);
  else if (var == 15)
    rc = launch_gemm<0, 4, 96, 1, 64, 4, 160>(g, stream);
  else
    rc = launch_gemm<0, 2, 128, 2>(g, stream);
  if (rc) return rc;
  if (!reduce) return MH_OK;                        // the caller consumes the slabs itself
  long gsz = ((long)g0.M * g0.N / 4 + 255) / 256;
  if (gsz > 4096) gsz = 4096;
  if (sbf)
    hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3((int)gsz), dim3(256), 0, stream, (const void*)ws, g0.C, g0.bias, g0.residual,
                       g0.M, g0.N, g0.ldc, g0.ldr, splits, g0.flags, g0.alpha);
  else
    hipLaunchKernelGGL(splitk_reduce_kernel<0>, dim3((int)gsz), dim3(256), 0, stream, (const void*)ws, g0.C, g0.bias, g0.residual,
                       g0.M, g0.N, g0.ldc, g0.ldr, splits, g0.flags, g0.alpha);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// caller-provided scratch for the automatic split-K path (no hidden allocation).  It lives in a scratch record: the
// process-default one (mh_set_workspace / mh_set_stream_workspace) or the record of the mh_ctx the caller made current
// (ctx.hip: mh_ctx_make_current) -- SURVEY 8(b): library state behind an opaque context.
MhScratch g_default_scratch = {nullptr, 0, {nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
MhScratch* g_scratch = &g_default_scratch;
#define g_ws (g_scratch->ws)
#define g_ws_bytes (g_scratch->bytes)
#define g_ws_alt (g_scratch->alt)
#define g_ws_alt_stream (g_scratch->alt_stream)
extern "C" int mh_set_workspace(void* ptr, long bytes) {
  g_ws = (float*)ptr;
  g_ws_bytes = ptr ? (size_t)bytes : 0;
  if (!ptr)
    for (int i = 0; i < MH_MAX_ALT_WS; ++i) { g_ws_alt[i] = nullptr; g_ws_alt_stream[i] = nullptr; }
  return MH_OK;
}
// Further scratches of the same size for launches on other streams (a frozen forward, a leaf backward running beside the
// main stream): the split-K slabs of concurrent GEMMs must not share memory.  Up to 4 streams; ptr = NULL unregisters.
extern "C" int mh_set_stream_workspace(hipStream_t stream, void* ptr, long bytes) {
  if (!stream || (ptr && (!g_ws || (size_t)bytes < g_ws_bytes))) return MH_ERR_ARG;
  int slot = -1;
  for (int i = 0; i < MH_MAX_ALT_WS; ++i)
    if (g_ws_alt_stream[i] == stream) slot = i;
  if (slot < 0)
    for (int i = MH_MAX_ALT_WS - 1; i >= 0; --i)
      if (!g_ws_alt_stream[i]) slot = i;
  if (slot < 0) return ptr ? MH_ERR_UNSUPPORTED : MH_OK;
  g_ws_alt[slot] = (float*)ptr;
  g_ws_alt_stream[slot] = ptr ? stream : nullptr;
  return MH_OK;
}
static inline float* ws_for(hipStream_t stream) {
  for (int i = 0; i < MH_MAX_ALT_WS; ++i)
    if (g_ws_alt_stream[i] == stream && g_ws_alt[i]) return g_ws_alt[i];
  return g_ws;
}

int mh_launch_gemv(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                   const float* bias, const float* residual, int ldr, int out_f32, float alpha, hipStream_t stream);

static int auto_splits(int M, int N, int K) {
  const long tiles = (long)((M + BM - 1) / BM) * ((N + 127) / 128);
  const int kt = K / 64;
  if (N % 4) return 1;
  if (M <= 512) {
    // skinny M (batch-1 step: 148 LLaMA rows, 257 ViT rows): the launch streams the weight matrix once and the only
    // question is whether enough workgroups are streaming -- aim at ~512 of them, at least 256 of K each.  Measured
    // (tools/gemm_skinny_sweep.py, cold weights): 148x4096x22016 288 -> 62 us at s = 8, 148x4096x4096 60 -> 22 us,
    // 257x1408x6144 86 -> 23 us; the reduce pass is small because M is.
    int s = (int)(512 / (tiles > 0 ? tiles : 1));
    s = s < 1 ? 1 : (s > 16 ? 16 : s);
    while (s > 1 && kt / s < 4) --s;
    return s;
  }
  if (tiles >= 256 && tiles < 400 && kt >= 128) return 3;
  if (tiles >= 128 && tiles < 256 && kt >= 64) return 2;
  if (tiles < 64 && kt >= 48) return 4;
  return 1;
}

// 8-wave kernels run one workgroup per CU: pick the split count that best fills whole rounds of 256 workgroups
static int g_force_big_splits = 0;
#ifdef MH_DEBUG_HOOKS
extern "C" void mhdbg_set_big_splits(int s) { g_force_big_splits = s; }   // sweep tools (libmyriad_hip_dbg.so only)
#endif

static int big_tile_splits(int M, int N, int K, int tile_n) {
  if (g_force_big_splits > 0) return g_force_big_splits;
  const long tiles = (long)((M + 255) / 256) * ((N + tile_n - 1) / tile_n);
  const int kt = K / 64;
  int best = 1;
  double best_eff = 0.0;
  for (int s = 1; s <= 5; ++s) {
    if (s > 1 && kt / s < 12) break;
    const long wg = tiles * s, rounds = (wg + 255) / 256;
    // the reduce pass re-reads s slabs: its cost relative to the GEMM grows as s / K.  Round 6 (tools/gemm_m648_splits.py, the
    // MiniGPT-4 arch's M = 648): a five-way split of the 48-tile N = 4096 launches (240 workgroups) beats four (192) by 5-8 %,
    // and a split that only trades a half-empty round for slabs + a reduce (144 / 129 / 258 tiles: 74.7 / 74.5 / 141.6 us unsplit
    // against 77.7 / 76.0 / 145.7 three-way) is not worth it -- unsplit also keeps the fused SwiGLU / direct bf16 epilogues.
    // The batch-8 Myriad shapes (80 / 85 / 240 / 215 / 430 tiles) and the ViT's keep their plans.
    const double eff = (double)wg / (double)(rounds * 256) / (s > 1 ? 1.0 + 900.0 * s / K : 1.0);
    if (eff > best_eff + 1e-9) { best_eff = eff; best = s; }
  }
  return best;
}

// plan kernel id -> forced-variant number of dispatch()
static inline int plan_variant(int kernel) {
  return kernel == 2 ? 12 : (kernel == 3 ? 3 : (kernel == 4 ? 14 : (kernel == 5 ? 15 : (kernel == 6 ? 11 : 0))));
}

// The automatic policy (flags carry no variant): which kernel runs and with how many K splits.
//   kernel 0: gemv.hip weight streaming (M <= 16: decode)
//   kernel 2: gemm_256.hip 256x256 tile, when it can put >= 128 workgroups on the chip with >= 1024 of K each
//             (+20..26 % over the 128x128 kernel on the step's LLaMA / ViT shapes, cold weights,
//             profiles/r01_gemm_256.md)
//   kernel 1: the 128x128 kernel for everything smaller -- it co-schedules two workgroups per CU and so hides its
//             own prologue / store tail, which the one-workgroup-per-CU 256x256 kernel cannot
//   kernel 6: 64x64 tiles, 8-deep ring, for one-round grids with short K (see gemm_plan)
//   kernel 3: the same kernel with a 128x64 tile and a 3-deep ring when even the 128x128 grid leaves most CUs without a
//             workgroup (Q-Former / VE-net shapes: 648x768 is 36 tiles).  A lone workgroup streams its operands at one
//             CU's L2 rate (~0.5 us per 64-deep step), so twice the workgroups is twice the CUs pulling: 14 -> 10 us at
//             K = 768, 27 -> 19 us at K = 2304 (tools/gemm_small_sweep.py); same k order per accumulator, same bits
static int g_force_kernel = -1, g_force_splits = 0;
static int g_dbg_flags = 0;
#ifdef MH_DEBUG_HOOKS
extern "C" void mhdbg_or_gemm_flags(int f) { g_dbg_flags = f; }   // timing experiments: internal flag bits OR-ed into every auto-planned launch
extern "C" void mhdbg_set_force_plan(int kernel, int splits) { g_force_kernel = kernel; g_force_splits = splits; }   // sweep tools (libmyriad_hip_dbg.so only)
#endif

// One row tile of 160 for 128 < M <= 160 (the batch-1 step's 148 LLaMA rows), two for M <= 320 (its 257 ViT rows): the
// weight matrix is streamed ONCE (the 128-row tile reads it twice and multiplies 108 padding rows, the 256-row tile
// multiplies 108), 160 x 96 or 160 x 128 x 64 tiles with a 4-deep ring (3 K tiles in flight per CU), K split so that the
// launch is one round of <= 256 workgroups.  Picks the column width / split count that fills the most CUs.
static bool skinny_plan(int M, int N, int K, bool can_split, int* kernel, int* splits) {
  const int g_skinny = mh_opt(MH_OPT_GEMM_SKINNY);
  // ... for weight matrices worth streaming (>= 4 M elements): the Q-Former / VE-net shapes stay on the 128x64 tiles
  if (!g_skinny || M <= 128 || M > 320 || N < 512 || K < 512 || (long)N * K < (4L << 20)) return false;
  const int tm = (M + 159) / 160;
  int best_k = 0, best_s = 1;
  double best_fill = 0.0;
  for (int pass = 0; pass < 2; ++pass) {
    const int tbn = pass ? 128 : 96;
    const int tiles = tm * ((N + tbn - 1) / tbn);
    int s = 256 / tiles;
    if (!can_split || s < 1) s = 1;
    while (s > 1 && K / s < 512) --s;
    if (s > 16) s = 16;
    const long wg = (long)tiles * s, rounds = (wg + 255) / 256;
    const double fill = (double)wg / (double)(rounds * 256) * ((double)N / (double)(((N + tbn - 1) / tbn) * tbn));
    if (fill >= best_fill - 1e-9) { best_fill = fill; best_k = pass ? 4 : 5; best_s = s; }
  }
  *kernel = best_k;
  *splits = best_s;
  return true;
}

static void gemm_plan(int M, int N, int K, int flags, int* kernel, int* splits) {
  *kernel = 1;
  *splits = 1;
  if (g_force_kernel >= 0) { *kernel = g_force_kernel; *splits = g_force_splits > 0 ? g_force_splits : 1; return; }
  if (M <= 16 && !(flags & (MH_GEMM_REGSTAGE | MH_GEMM_GELU))) { *kernel = 0; return; }
  if (flags & MH_GEMM_REGSTAGE) return;
  const bool can_split = g_ws && (N % 4) == 0;
  {
    int k2, s2;
    // the scratch only has to hold the slabs of a launch that actually splits K (an unsplit 160-row launch was
    // refused when the registered workspace was small)
    if (skinny_plan(M, N, K, can_split, &k2, &s2) && (s2 == 1 || (size_t)s2 * M * N * sizeof(float) <= g_ws_bytes)) {
      *kernel = k2;
      *splits = s2;
      return;
    }
  }
  if (M > 128) {
    const long tiles = (long)((M + 255) / 256) * ((N + 255) / 256);
    int s = can_split ? big_tile_splits(M, N, K, 256) : 1;
    if ((size_t)s * M * N * sizeof(float) > g_ws_bytes) s = 1;
    if (tiles * s >= 128 && K / s >= 768) { *kernel = 2; *splits = s; return; }
  }
  if (can_split) {
    const int s = auto_splits(M, N, K);
    if (s > 1 && (size_t)s * M * N * sizeof(float) <= g_ws_bytes) *splits = s;
  }
  // kernel 6: one round of 64x64 tiles with an 8-deep ring (7 k-tiles in flight, 128 KiB of LDS, one workgroup per CU) when
  // that round fits the chip and K is short: a K = 768 product has all 12 of its k-tiles requested before the first MFMA, so
  // the launch is one memory latency instead of four (round 3, tools/gemm_small_sweep.py, back-to-back launches: 648x768x768
  // 9.9 -> 8.2 us, 81x768x768 split 3 + reduce 10.3 -> 8.1, 648x768x2304 19.2 -> 14.8, 81x3072x768 11.0 -> 8.4,
  // 256x1408x640 11.4 -> 8.0); never K-split, so no scratch and no reduce launch behind it
  {
    const long t64 = (long)((M + 63) / 64) * ((N + 63) / 64);
    if (M > 16 && t64 <= 256 && K <= 3072 && K >= 256) { *kernel = 6; *splits = 1; return; }
  }
  const long t128 = (long)((M + BM - 1) / BM) * ((N + 127) / 128);
  // < 256: the 128x128 grid would not give every CU a workgroup (round 3, tools/gemm_small_sweep.py: 2056x1408x1408, 187
  // tiles, 22.0 -> 17.9 us; 648x4096x768 15.1 -> 12.7 us; the bound was 160 before)
  // unsplit only: a K-split launch of this policy branch runs on the 128x128 instance (run_splitk), and the plan says so
  // (mh_gemm_plan and the profiler reported kernel 3 for a launch that ran kernel 1)
  if (t128 * *splits < 256 && *splits == 1) *kernel = 3;
}

extern "C" int mh_gemm_plan(int M, int N, int K, int flags, int* kernel, int* splits) {
  if (!kernel || !splits || M <= 0 || N <= 0 || K <= 0 || ((flags >> MH_GEMM_VARIANT_SHIFT) & 15)) return MH_ERR_ARG;
  gemm_plan(M, N, K, flags, kernel, splits);
  return MH_OK;
}

extern "C" int mh_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                               const float* bias, const float* residual, int ldr, int flags, float alpha,
                               hipStream_t stream) {
  if (M <= 0 || N <= 0) return MH_OK;
  if (K <= 0 || (K % 64) != 0 || (lda % 8) != 0 || (ldb % 8) != 0) return MH_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15)) return MH_ERR_ARG;
  if ((ldc % 4) != 0 || (residual && (ldr % 4) != 0)) return MH_ERR_ARG;
  GemmArgs g = {A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, flags, alpha, 1, K / 64, 0L};
  const int variant = (flags >> MH_GEMM_VARIANT_SHIFT) & 15;
  if (variant == 0) {
    int kernel, splits;
    gemm_plan(M, N, K, flags, &kernel, &splits);
    if (kernel == 0)
      return mh_launch_gemv(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, (flags & MH_GEMM_OUT_F32) ? 1 : 0,
                            alpha, stream);
    g.flags |= (plan_variant(kernel) << MH_GEMM_VARIANT_SHIFT) | g_dbg_flags;
    // the split-K reduce reads `residual` and writes C element-wise, so C may alias residual (in-place accumulate)
    return splits > 1 ? run_splitk(g, splits, ws_for(stream), stream) : dispatch(g, stream);
  }
  // forced variants (A/B tools): the 8-wave kernels still pick their own split count
  if (variant == 12 && g_ws && (N % 4) == 0) {
    const int best = big_tile_splits(M, N, K, 256);
    if (best > 1 && (size_t)best * M * N * sizeof(float) <= g_ws_bytes) return run_splitk(g, best, ws_for(stream), stream);
  }
  return dispatch(g, stream);
}

extern "C" int mh_rmsnorm_fwd(const float* x, const float* w, void* y_bf16, long ldy, int M, int D, float eps, hipStream_t stream);
extern "C" int mh_layernorm_fwd(const float* x, const float* w, const float* b, void* y_bf16, float* y_f32, int M, int D,
                                float eps, hipStream_t stream);

// H[M,N] = A.B^T (+bias) + residual (fp32), Y = norm(H) (bf16): a Linear, its residual add and the norm that reads the sum
static int gemm_residual_norm(int norm, const void* A, int lda, const void* B, int ldb, float* H, int ldh, const float* bias,
                              const float* residual, int ldr, const float* norm_w, const float* norm_b, float eps, void* Y,
                              long ldy, int M, int N, int K, hipStream_t stream) {
  if (M <= 0 || N <= 0) return MH_OK;
  if (!norm_w || !H || !Y || (N % 4) != 0 || (ldy % 4) != 0 || (norm == 1 && (!norm_b || ldy != N))) return MH_ERR_ARG;
  int kernel = 1, splits = 1;
  if (K > 0) gemm_plan(M, N, K, MH_GEMM_OUT_F32, &kernel, &splits);
  if (splits > 1 && kernel != 0 && N <= 8192 && (ldh % 4) == 0 && (!residual || (ldr % 4) == 0) && (K % 64) == 0 &&
      (lda % 8) == 0 && (ldb % 8) == 0 && !(((uintptr_t)A | (uintptr_t)B | (uintptr_t)H) & 15)) {
    GemmArgs g = {A, lda, B, ldb, (void*)H, ldh, M, N, K, nullptr, residual, ldr, MH_GEMM_OUT_F32, 1.0f, 1, K / 64, 0L};
    g.flags |= plan_variant(kernel) << MH_GEMM_VARIANT_SHIFT;
    const int nt = K / 64;                          // the split count run_splitk will settle on
    int sp = splits > nt ? nt : splits;
    const int tps = (nt + sp - 1) / sp;
    sp = (nt + tps - 1) / tps;
    float* wsp = ws_for(stream);
    int sbf = 0;
    int rc = run_splitk(g, splits, wsp, stream, /*reduce=*/false, &sbf);
    if (rc) return rc;
#define RN_LAUNCH1(NORM_, SBF_, NIT_)                                                                                       \
  hipLaunchKernelGGL((splitk_reduce_norm_kernel<NORM_, SBF_, NIT_>), dim3(M), dim3(256), 0, stream, (const void*)wsp, bias, residual, H, \
                     norm_w, norm_b, (bf16_t*)Y, N, (long)ldr, (long)ldh, ldy, (long)M * N, sp, eps)
#define RN_LAUNCH(NORM_, SBF_)                                                                                              \
  do { if (N <= 4096) RN_LAUNCH1(NORM_, SBF_, 4); else RN_LAUNCH1(NORM_, SBF_, 8); } while (0)
    if (norm == 0) { if (sbf) RN_LAUNCH(0, 1); else RN_LAUNCH(0, 0); }
    else { if (sbf) RN_LAUNCH(1, 1); else RN_LAUNCH(1, 0); }
#undef RN_LAUNCH
#undef RN_LAUNCH1
    MH_CHECK_LAUNCH();
    return MH_OK;
  }
  int rc = mh_gemm_bf16_nt(A, lda, B, ldb, H, ldh, M, N, K, bias, residual, ldr, MH_GEMM_OUT_F32, 1.0f, stream);
  if (rc) return rc;
  if (ldh != N) return MH_ERR_ARG;                  // the norm kernels read a dense [M, N] stream
  if (norm == 0) return mh_rmsnorm_fwd(H, norm_w, Y, ldy, M, N, eps, stream);
  return mh_layernorm_fwd(H, norm_w, norm_b, Y, nullptr, M, N, eps, stream);
}

extern "C" int mh_gemm_residual_rmsnorm(const void* A, int lda, const void* B, int ldb, float* H, int ldh,
                                        const float* residual, int ldr, const float* norm_w, float eps, void* Y, long ldy,
                                        int M, int N, int K, hipStream_t stream) {
  return gemm_residual_norm(0, A, lda, B, ldb, H, ldh, nullptr, residual, ldr, norm_w, nullptr, eps, Y, ldy, M, N, K, stream);
}

extern "C" int mh_gemm_residual_layernorm(const void* A, int lda, const void* B, int ldb, float* H, int ldh, const float* bias,
                                          const float* residual, int ldr, const float* norm_w, const float* norm_b, float eps,
                                          void* Y, int M, int N, int K, hipStream_t stream) {
  return gemm_residual_norm(1, A, lda, B, ldb, H, ldh, bias, residual, ldr, norm_w, norm_b, eps, Y, N, M, N, K, stream);
}

int mh_launch_rmsnorm_bwd(const void* dy, int slab_bf16, int nslab, long slab, long ldy, const float* x, const float* w,
                          const float* dres, float* dx, void* dx_bf16, int M, int D, float eps, hipStream_t stream);

// dY[M,N] = A.B^T (a dgrad GEMM), then the RMSNorm backward that consumes it (modeling_llama.py:66-74 under autograd):
// dx = rmsnorm_bwd(dY, x, w) + dres.  When the policy splits K the norm kernel sums the partial slabs itself; otherwise dY
// goes through dy_buf [M, N] f32.  Bit-identical to mh_gemm_bf16_nt followed by mh_rmsnorm_bwd.
extern "C" int mh_gemm_rmsnorm_bwd(const void* A, int lda, const void* B, int ldb, float* dy_buf, const float* x, const float* w,
                                   const float* dres, float* dx, void* dx_bf16, int M, int N, int K, float eps,
                                   hipStream_t stream) {
  if (M <= 0 || N <= 0) return MH_OK;
  if (!dy_buf || !x || !w || (N % 4) != 0 || N > 8192) return MH_ERR_ARG;
  int kernel = 1, splits = 1;
  if (K > 0) gemm_plan(M, N, K, MH_GEMM_OUT_F32, &kernel, &splits);
  if (splits > 1 && kernel != 0 && (K % 64) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 &&
      !(((uintptr_t)A | (uintptr_t)B) & 15)) {
    GemmArgs g = {A, lda, B, ldb, (void*)dy_buf, N, M, N, K, nullptr, nullptr, 0, MH_GEMM_OUT_F32, 1.0f, 1, K / 64, 0L};
    g.flags |= plan_variant(kernel) << MH_GEMM_VARIANT_SHIFT;
    const int nt = K / 64;                          // the split count run_splitk will settle on
    int sp = splits > nt ? nt : splits;
    const int tps = (nt + sp - 1) / sp;
    sp = (nt + tps - 1) / tps;
    float* wsp = ws_for(stream);
    int sbf = 0;
    const int rc = run_splitk(g, splits, wsp, stream, /*reduce=*/false, &sbf);
    if (rc) return rc;
    return mh_launch_rmsnorm_bwd(wsp, sbf, sp, (long)M * N, N, x, w, dres, dx, dx_bf16, M, N, eps, stream);
  }
  const int rc = mh_gemm_bf16_nt(A, lda, B, ldb, dy_buf, N, M, N, K, nullptr, nullptr, 0, MH_GEMM_OUT_F32, 1.0f, stream);
  if (rc) return rc;
  return mh_launch_rmsnorm_bwd(dy_buf, 0, 1, 0, N, x, w, dres, dx, dx_bf16, M, N, eps, stream);
}

int mh_launch_lora_dx(const void* dx_ext, int slab_bf16, long ld, int nslab, long slab, const float* A, float* out, float* border_out,
                      int M, int D, int R2_, float s, float p, unsigned long long seed, hipStream_t stream);

// The qkv dgrad with the LoRA border, [M, D + 64] = dqkv . [W_qkv | B_ext] (myriad_amd/lora.py), followed by the LoRA dx
// correction that reads it (lora.hip).  When the policy splits K the correction kernel sums the partial slabs itself (same
// order as splitk_reduce_kernel, so the same bits as mh_gemm_bf16_nt + mh_lora_dx) and writes the summed border [M, 64] for the
// weight-gradient kernel (border_out [M, 64] f32); otherwise (mh_gemm_plan(M, D + 64, K) reports one split) the product goes
// through dx_ext_buf [M, D + 64] f32, which then also holds the border, and border_out is not written.  dxn [M, D] f32.
extern "C" int mh_gemm_lora_dx(const void* A, int lda, const void* Bw, int ldb, float* dx_ext_buf, const float* loraA, float* dxn,
                               float* border_out, int M, int D, int K, int R2, float s, float p, unsigned long long seed,
                               hipStream_t stream) {
  if (M <= 0) return MH_OK;
  const int N = D + 64;
  if (!loraA || !dxn || D <= 0 || (D % 4) != 0) return MH_ERR_ARG;
  int kernel = 1, splits = 1;
  if (K > 0) gemm_plan(M, N, K, MH_GEMM_OUT_F32, &kernel, &splits);
  if (splits > 1 && kernel != 0 && border_out && (K % 64) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 &&
      !(((uintptr_t)A | (uintptr_t)Bw) & 15)) {
    GemmArgs g = {A, lda, Bw, ldb, (void*)dx_ext_buf, N, M, N, K, nullptr, nullptr, 0, MH_GEMM_OUT_F32, 1.0f, 1, K / 64, 0L};
    g.flags |= plan_variant(kernel) << MH_GEMM_VARIANT_SHIFT;
    const int nt = K / 64;                          // the split count run_splitk will settle on
    int sp = splits > nt ? nt : splits;
    const int tps = (nt + sp - 1) / sp;
    sp = (nt + tps - 1) / tps;
    float* wsp = ws_for(stream);
    int sbf = 0;
    const int rc = run_splitk(g, splits, wsp, stream, /*reduce=*/false, &sbf);
    if (rc) return rc;
    return mh_launch_lora_dx(wsp, sbf, N, sp, (long)M * N, loraA, dxn, border_out, M, D, R2, s, p, seed, stream);
  }
  if (!dx_ext_buf) return MH_ERR_ARG;
  const int rc = mh_gemm_bf16_nt(A, lda, Bw, ldb, dx_ext_buf, N, M, N, K, nullptr, nullptr, 0, MH_GEMM_OUT_F32, 1.0f, stream);
  if (rc) return rc;
  return mh_launch_lora_dx(dx_ext_buf, 0, N, 1, 0, loraA, dxn, nullptr, M, D, R2, s, p, seed, stream);
}

int mh_launch_lora_dx_rmsnorm_bwd(const void* dx_ext, int slab_bf16, long ld, int nslab, long slab, const float* A, const float* x,
                                  const float* w, const float* dres, float* dx, void* dx_bf16, float* border_out, int M, int D,
                                  int R2_, float s, float p, unsigned long long seed, float eps, hipStream_t stream);

// mh_gemm_lora_dx followed by the backward of the RMSNorm whose output the qkv projection read (the layer's input norm):
// dx = d rmsnorm(x; w)(dxn) + dres.  With D <= 4096 and r = 8 the LoRA correction and the norm backward are ONE kernel that sums
// the dgrad's split-K slabs itself (lora.hip: lora_dx_rmsnorm_bwd_kernel); otherwise the two kernels run back to back through
// dxn_buf [M, D] f32.  Same bits either way.  dx_ext_buf / border_out as in mh_gemm_lora_dx.
extern "C" int mh_gemm_lora_rmsnorm_bwd(const void* A, int lda, const void* Bw, int ldb, float* dx_ext_buf, const float* loraA,
                                        float* dxn_buf, float* border_out, const float* x, const float* w, const float* dres,
                                        float* dx, void* dx_bf16, int M, int D, int K, int R2, float s, float p,
                                        unsigned long long seed, float eps, hipStream_t stream) {
  if (M <= 0) return MH_OK;
  const int N = D + 64;
  if (!loraA || !x || !w || D <= 0 || (D % 4) != 0) return MH_ERR_ARG;
  int kernel = 1, splits = 1;
  if (K > 0) gemm_plan(M, N, K, MH_GEMM_OUT_F32, &kernel, &splits);
  const void* prod = dx_ext_buf;
  int sbf = 0, sp = 1;
  long slab = 0;
  float* bout = nullptr;
  if (splits > 1 && kernel != 0 && border_out && (K % 64) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 &&
      !(((uintptr_t)A | (uintptr_t)Bw) & 15)) {
    GemmArgs g = {A, lda, Bw, ldb, (void*)dx_ext_buf, N, M, N, K, nullptr, nullptr, 0, MH_GEMM_OUT_F32, 1.0f, 1, K / 64, 0L};
    g.flags |= plan_variant(kernel) << MH_GEMM_VARIANT_SHIFT;
    const int nt = K / 64;                          // the split count run_splitk will settle on
    sp = splits > nt ? nt : splits;
    const int tps = (nt + sp - 1) / sp;
    sp = (nt + tps - 1) / tps;
    float* wsp = ws_for(stream);
    const int rc = run_splitk(g, splits, wsp, stream, /*reduce=*/false, &sbf);
    if (rc) return rc;
    prod = wsp; slab = (long)M * N; bout = border_out;
  } else {
    if (!dx_ext_buf) return MH_ERR_ARG;
    const int rc = mh_gemm_bf16_nt(A, lda, Bw, ldb, dx_ext_buf, N, M, N, K, nullptr, nullptr, 0, MH_GEMM_OUT_F32, 1.0f, stream);
    if (rc) return rc;
  }
  if (mh_opt(MH_OPT_LORA_NORM_FUSED)) {
    const int rc = mh_launch_lora_dx_rmsnorm_bwd(prod, sbf, N, sp, slab, loraA, x, w, dres, dx, dx_bf16, bout, M, D, R2, s, p, seed,
                                                 eps, stream);
    if (rc != MH_ERR_UNSUPPORTED) return rc;
  }
  if (!dxn_buf) return MH_ERR_ARG;
  const int rc = mh_launch_lora_dx(prod, sbf, N, sp, slab, loraA, dxn_buf, bout, M, D, R2, s, p, seed, stream);
  if (rc) return rc;
  return mh_launch_rmsnorm_bwd(dxn_buf, 0, 1, 0, D, x, w, dres, dx, dx_bf16, M, D, eps, stream);
}

int mh_launch_attn_rope_bwd(const void* qkv, int ld, const void* o, int ldo, const void* dout, int dout_is_bf16, int nslab,
                            long slab, int ldd, const float* lse, void* dqkv, const int* pos, const float* cos_tab,
                            const float* sin_tab, const int* kv_len, int B, int H, int S, int D, float scale,
                            hipStream_t stream);

// dO[M = B*S, N = H*D] = A.Bw^T (the o_proj dgrad, modeling_llama.py:222-224 under autograd), then the fused rotary
// attention backward that consumes it (attn_seq.hip).  When the policy splits K the attention kernel sums the partial
// slabs itself while it loads its dO rows (the same fixed order and single rounding as splitk_reduce_kernel, so the same
// bits as mh_gemm_bf16_nt + mh_attn_rope_bwd); otherwise dO goes through do_buf [M, N] bf16.
extern "C" int mh_gemm_attn_rope_bwd(const void* A, int lda, const void* Bw, int ldb, void* do_buf, int K, const void* qkv,
                                     int ld, const void* o, int ldo, const float* lse, void* dqkv, const int* pos,
                                     const float* cos_tab, const float* sin_tab, const int* kv_len, int B, int H, int S,
                                     int D, float scale, hipStream_t stream) {
  const int M = B * S, N = H * D;
  if (M <= 0 || N <= 0) return MH_OK;
  if (!do_buf) return MH_ERR_ARG;
  int kernel = 1, splits = 1;
  if (K > 0) gemm_plan(M, N, K, MH_GEMM_OUT_F32, &kernel, &splits);
  if (splits > 1 && kernel != 0 && (K % 64) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 && (N % 8) == 0 &&
      !(((uintptr_t)A | (uintptr_t)Bw) & 15)) {
    GemmArgs g = {A, lda, Bw, ldb, do_buf, N, M, N, K, nullptr, nullptr, 0, MH_GEMM_OUT_F32, 1.0f, 1, K / 64, 0L};
    g.flags |= plan_variant(kernel) << MH_GEMM_VARIANT_SHIFT;
    const int nt = K / 64;                          // the split count run_splitk will settle on
    int sp = splits > nt ? nt : splits;
    const int tps = (nt + sp - 1) / sp;
    sp = (nt + tps - 1) / tps;
    float* wsp = ws_for(stream);
    int sbf = 0;
    const int rc = run_splitk(g, splits, wsp, stream, /*reduce=*/false, &sbf);
    if (rc) return rc;
    return mh_launch_attn_rope_bwd(qkv, ld, o, ldo, wsp, sbf, sp, (long)M * N, N, lse, dqkv, pos, cos_tab, sin_tab, kv_len, B, H, S,
                                   D, scale, stream);
  }
  const int rc = mh_gemm_bf16_nt(A, lda, Bw, ldb, do_buf, N, M, N, K, nullptr, nullptr, 0, 0, 1.0f, stream);
  if (rc) return rc;
  return mh_launch_attn_rope_bwd(qkv, ld, o, ldo, do_buf, 1, 1, 0, N, lse, dqkv, pos, cos_tab, sin_tab, kv_len, B, H, S, D, scale,
                                 stream);
}

// ---- SiLU-gated MLP (modeling_llama.py:139-140) fused into the two GEMMs around it -------------------------------------------
// The gate|up weight is held with its rows interleaved in blocks of 128 (llama.py builds it so): gu[M, 2I] =
// [g 0..127 | u 0..127 | g 128..255 | ...].  When the policy runs the unsplit 256-column kernel the elementwise part rides
// its epilogue (gemm_256.hip, MH_GEMM_SWIGLU_*); otherwise the GEMM and the block-layout silu kernel run back to back --
// the same bits either way (g, u, dact are rounded to bf16 before the elementwise math in both forms).
#define MH_GEMM_SWIGLU_FWD 16
#define MH_GEMM_SWIGLU_BWD 32
extern "C" int mh_silu_mul_fwd_blk(const void* gu, void* h, int M, int I, int blk, hipStream_t stream);
extern "C" int mh_silu_mul_bwd_blk(const void* dh, const void* gu, void* dgu, int M, int I, int blk, hipStream_t stream);
int mh_launch_silu_mul_bwd_slabs(const void* ws, int slab_bf16, int nslab, long slab, const void* gu, void* dgu, int M, int I, int blk,
                                 hipStream_t stream);


static bool swiglu_fusable(int M, int N, int K, int lda, int ldb, const void* A, const void* B) {
  int kernel = 1, splits = 1;
  gemm_plan(M, N, K, 0, &kernel, &splits);
  // Measured, batch-8 step, same box A/B: round 2 fused 52.21 ms vs separate launches 51.98; round 4 (eight-wave loop) 44.8 vs 44.8 --
  // the read-out of a one-workgroup-per-CU GEMM was instruction-bound (integer bf16 rounding), so the elementwise work added
  // to it cost what the saved launch and round trip of dact / gu gave back.  With the hardware rounding the fused forms win:
  // 42.20 / 42.20 ms against 42.34 / 42.41 (late round 4).  On by default; MYRIAD_SWIGLU_FUSED=0 runs the separate launches.
  return mh_opt(MH_OPT_SWIGLU_FUSED) && kernel == 2 && splits == 1 && (N % 128) == 0 && (K % 64) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 &&
         !(((uintptr_t)A | (uintptr_t)B) & 15);
}

// gu[M, 2I] = X[M, K] . Wgu[2I, K]^T (bf16, saved for the backward) and act[M, I] = silu(g) * u
int mh_launch_silu_mul_fwd_blk_ld(const void* gu, long ldg, void* h, long ldh, int M, int I, int blk, hipStream_t stream);

extern "C" int mh_gemm_swiglu_fwd(const void* X, int ldx, const void* Wgu, int ldw, void* gu, int ldgu, void* act, int ldact,
                                  int M, int I, int K, hipStream_t stream) {
  if (M <= 0 || I <= 0) return MH_OK;
  if (!gu || !act || (I % 128) || (ldgu % 8) || (ldact % 8) || ldgu < 2 * I || ldact < I) return MH_ERR_ARG;
  if (swiglu_fusable(M, 2 * I, K, ldx, ldw, X, Wgu)) {
    // Round 6: a tile count a few tiles past a whole number of rounds (the MiniGPT-4 arch: 648 rows x 22016 columns = 258 tiles
    // on 256 CUs, a second round for two tiles: 141 us where one round is ~75) -- the fused launch takes the column blocks that
    // fill whole rounds, the one or two left-over 256-column blocks (whole gate | up pairs of the block-128 layout) go through
    // the ordinary planned product on their slice (a K-split launch of small tiles) and the strided gate kernel.  Those columns
    // then carry split-K rounding (fp32 partial sums added before the bf16 rounding) instead of one unsplit accumulation.
    const int tm = (M + 255) / 256, tn = (2 * I) / 256;
    const long tiles = (long)tm * tn;
    const int over = (int)(tiles % 256);
    const int lc = (over + tm - 1) / tm;                      // column tiles to peel off
    if (tiles > 256 && over > 0 && lc <= 2 && (2 * I) % 256 == 0 && lc < tn && g_ws) {
      const int N1 = (tn - lc) * 256, NL = lc * 256;
      int rc = mh_launch_gemm_256(X, ldx, Wgu, ldw, gu, ldgu, M, N1, K, nullptr, nullptr, 0, MH_GEMM_SWIGLU_FWD, 1.0f, 1, K / 64,
                                  0L, stream, act, ldact);
      if (rc) return rc;
      bf16_t* gul = reinterpret_cast<bf16_t*>(gu) + N1;
      rc = mh_gemm_bf16_nt(X, ldx, reinterpret_cast<const bf16_t*>(Wgu) + (long)N1 * ldw, ldw, gul, ldgu, M, NL, K, nullptr, nullptr,
                           0, 0, 1.0f, stream);
      if (rc) return rc;
      return mh_launch_silu_mul_fwd_blk_ld(gul, ldgu, reinterpret_cast<bf16_t*>(act) + N1 / 2, ldact, M, NL / 2, 128, stream);
    }
    return mh_launch_gemm_256(X, ldx, Wgu, ldw, gu, ldgu, M, 2 * I, K, nullptr, nullptr, 0, MH_GEMM_SWIGLU_FWD, 1.0f, 1, K / 64,
                              0L, stream, act, ldact);
  }
  if (ldgu != 2 * I || ldact != I) return MH_ERR_ARG;         // the elementwise kernels take dense rows
  const int rc = mh_gemm_bf16_nt(X, ldx, Wgu, ldw, gu, ldgu, M, 2 * I, K, nullptr, nullptr, 0, 0, 1.0f, stream);
  if (rc) return rc;
  return mh_silu_mul_fwd_blk(gu, act, M, I, 128, stream);
}

// dgu[M, 2I] = silu_mul_bwd(dact, gu) with dact[M, I] = dH[M, K] . WdT[I, K]^T (the down projection's dgrad); dact_buf [M, I]
// bf16 is scratch for the unfused case
extern "C" int mh_gemm_swiglu_bwd(const void* dH, int lddh, const void* WdT, int ldw, const void* gu, int ldgu, void* dgu,
                                  int lddgu, void* dact_buf, int M, int I, int K, hipStream_t stream) {
  if (M <= 0 || I <= 0) return MH_OK;
  if (!gu || !dgu || (I % 128) || (ldgu % 8) || (lddgu % 8) || ldgu < 2 * I || lddgu < 2 * I) return MH_ERR_ARG;
  if (swiglu_fusable(M, I, K, lddh, ldw, dH, WdT))
    return mh_launch_gemm_256(dH, lddh, WdT, ldw, dgu, lddgu, M, I, K, nullptr, nullptr, 0, MH_GEMM_SWIGLU_BWD, 1.0f, 1, K / 64,
                              0L, stream, const_cast<void*>(gu), ldgu);
  if (!dact_buf || ldgu != 2 * I || lddgu != 2 * I) return MH_ERR_ARG;
  {
    // K-split product (the batch-1 step's 160-row tiles): the gate backward sums the slabs itself -- no reduce launch, same bits
    int kernel = 1, splits = 1;
    gemm_plan(M, I, K, 0, &kernel, &splits);
    if (splits > 1 && kernel != 0 && (I % 8) == 0 && (K % 64) == 0 && (lddh % 8) == 0 && (ldw % 8) == 0 &&
        !(((uintptr_t)dH | (uintptr_t)WdT) & 15)) {
      GemmArgs g = {dH, lddh, WdT, ldw, dact_buf, I, M, I, K, nullptr, nullptr, 0, 0, 1.0f, 1, K / 64, 0L};
      g.flags |= plan_variant(kernel) << MH_GEMM_VARIANT_SHIFT;
      const int nt = K / 64;
      int sp = splits > nt ? nt : splits;
      const int tps = (nt + sp - 1) / sp;
      sp = (nt + tps - 1) / tps;
      float* wsp = ws_for(stream);
      int sbf = 0;
      const int rc = run_splitk(g, splits, wsp, stream, /*reduce=*/false, &sbf);
      if (rc) return rc;
      return mh_launch_silu_mul_bwd_slabs(wsp, sbf, sp, (long)M * I, gu, dgu, M, I, 128, stream);
    }
  }
  const int rc = mh_gemm_bf16_nt(dH, lddh, WdT, ldw, dact_buf, I, M, I, K, nullptr, nullptr, 0, 0, 1.0f, stream);
  if (rc) return rc;
  return mh_silu_mul_bwd_blk(dact_buf, gu, dgu, M, I, 128, stream);
}

// ---- erf-GELU MLP (Qformer.py:481-484 intermediate_query / output_query, eva_vit.py:54-61) fused into the two GEMMs around it ----
// Forward: pre[M, N] = X . W^T + bias (bf16, saved for the backward) and act = gelu(pre); backward: dpre = dact * gelu'(pre) with
// dact = dY . WT^T rounded to bf16.  When the plan runs one of the unsplit LDS-DMA tile kernels (plan kernels 1 / 3 / 6: the
// Q-Former's 648-row products) the elementwise half rides that launch's epilogue; otherwise GEMM and elementwise kernel run back
// to back -- the same bits either way (the epilogue rounds to bf16 where the separate launches do).  One launch less per
// direction per BertLayer on a chain of ~400 dependent 5-15-us launches.
extern "C" int mh_gelu_fwd(const void* x, void* y, long n, hipStream_t stream);
extern "C" int mh_gelu_bwd(const void* dy, const void* x, void* dx, long n, hipStream_t stream);

static bool gelu_fusable(int M, int N, int K, int* variant) {
  int kernel = 1, splits = 1;
  gemm_plan(M, N, K, 0, &kernel, &splits);
  *variant = plan_variant(kernel);
  return mh_opt(MH_OPT_GELU_FUSED) && splits == 1 && (kernel == 1 || kernel == 3 || kernel == 6);
}

extern "C" int mh_gemm_gelu_fwd(const void* X, int ldx, const void* W, int ldw, const float* bias, void* pre, int ldpre, void* act,
                                int ldact, int M, int N, int K, hipStream_t stream) {
  if (M <= 0 || N <= 0) return MH_OK;
  if (!pre || !act || K <= 0 || (K % 64) || (ldx % 8) || (ldw % 8) || (ldpre % 4) || (ldact % 4) || ldpre < N || ldact < N) return MH_ERR_ARG;
  if (((uintptr_t)X | (uintptr_t)W | (uintptr_t)pre | (uintptr_t)act) & 15) return MH_ERR_ARG;
  int variant = 0;
  if (gelu_fusable(M, N, K, &variant)) {
    GemmArgs g = {X, ldx, W, ldw, act, ldact, M, N, K, bias, nullptr, 0, MH_GEMM_GELU_PRE | (variant << MH_GEMM_VARIANT_SHIFT), 1.0f,
                  1, K / 64, 0L, pre, ldpre};
    return dispatch(g, stream);
  }
  if (ldpre != N || ldact != N || (((long)M * N) % 8)) return MH_ERR_ARG;        // the elementwise kernels take dense rows
  const int rc = mh_gemm_bf16_nt(X, ldx, W, ldw, pre, ldpre, M, N, K, bias, nullptr, 0, 0, 1.0f, stream);
  if (rc) return rc;
  return mh_gelu_fwd(pre, act, (long)M * N, stream);
}

// dact_buf [M, N] bf16 is scratch for the unfused case (may be NULL when the caller knows the plan fuses: then MH_ERR_ARG if not)
extern "C" int mh_gemm_gelu_bwd(const void* dY, int lddy, const void* WT, int ldw, const void* pre, int ldpre, void* dpre, int lddpre,
                                void* dact_buf, int M, int N, int K, hipStream_t stream) {
  if (M <= 0 || N <= 0) return MH_OK;
  if (!pre || !dpre || K <= 0 || (K % 64) || (lddy % 8) || (ldw % 8) || (ldpre % 4) || (lddpre % 4) || ldpre < N || lddpre < N) return MH_ERR_ARG;
  if (((uintptr_t)dY | (uintptr_t)WT | (uintptr_t)pre | (uintptr_t)dpre) & 15) return MH_ERR_ARG;
  int variant = 0;
  if (gelu_fusable(M, N, K, &variant)) {
    GemmArgs g = {dY, lddy, WT, ldw, dpre, lddpre, M, N, K, nullptr, nullptr, 0, MH_GEMM_GELU_BWD | (variant << MH_GEMM_VARIANT_SHIFT),
                  1.0f, 1, K / 64, 0L, const_cast<void*>(pre), ldpre};
    return dispatch(g, stream);
  }
  if (!dact_buf || ldpre != N || lddpre != N || (((long)M * N) % 8)) return MH_ERR_ARG;
  const int rc = mh_gemm_bf16_nt(dY, lddy, WT, ldw, dact_buf, N, M, N, K, nullptr, nullptr, 0, 0, 1.0f, stream);
  if (rc) return rc;
  return mh_gelu_bwd(dact_buf, pre, dpre, (long)M * N, stream);
}

// ---- explicit split-K entry (wgrad of the conv stem: M,N small, K huge); caller passes the scratch ----
extern "C" long mh_gemm_splitk_ws_floats(int M, int N, int splits) { return (long)M * N * splits; }

extern "C" int mh_gemm_bf16_nt_splitk(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N,
                                      int K, int splits, float* ws, hipStream_t stream) {
  if (M <= 0 || N <= 0) return MH_OK;
  if (K <= 0 || (K % 64) != 0 || (lda % 8) != 0 || (ldb % 8) != 0 || (N % 4) != 0 || (ldc % 4) != 0 || splits < 1)
    return MH_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)ws & 15) || ((uintptr_t)C & 15)) return MH_ERR_ARG;
  GemmArgs g = {A, lda, B, ldb, (void*)C, ldc, M, N, K, nullptr, nullptr, 0, MH_GEMM_OUT_F32, 1.0f, 1, K / 64, 0L};
  return run_splitk(g, splits, ws, stream);
}

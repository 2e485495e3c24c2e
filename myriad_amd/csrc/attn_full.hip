// Whole-sequence attention forward for the frozen encoders (no mask, no rotary, no bias): EVA-ViT-g's 257-token self-attention
// (eva_vit.py:118-148: q, k, v of one Linear, head dim 88, scale q.k^T, softmax, .v) and the Q-Former's self / cross attention
// (Qformer.py:169-275 with head dim 64) when their keys fit a CU's LDS.
//
// Why (round 5): the 64x64-tile kernel of attention.hip runs the ViT's attention as 5 query tiles x 128 (image, head) workgroups,
// each staging K and V^T tile by tile through registers with transposed 2-byte LDS stores and two barriers per key tile:
// 28.5 us per ViT block at 3 % matrix utilisation, 1.1 ms of the look-ahead ViT per step.  Here -- the structure of attn_seq.hip
// without its rotary and causal parts -- a workgroup stages the (image, head)'s K and V ONCE as key-major LDS images, every wave
// owns whole 16-query fragments, reads the V^T operand through the gfx950 transpose read (ds_read_b64_tr_b16) and keeps the
// softmax per lane (online over 32-key chunks).  The queries of one (image, head) are split over gridDim.y workgroups (each
// stages all keys) so that the launch fills the chip: ViT 8 images x 16 heads x 2 = 256 workgroups.
//
// MFMA mapping as attn_seq.hip: S^T = K.Q^T (keys in the accumulator rows, one query per lane column), O^T = V^T.P^T.
// Head dims that are not a multiple of 32 (ViT: 88) are zero-padded to DP = 96 in the LDS images and in the Q fragments.
#include "common.h"

#define AF_NW 8              // waves per workgroup
typedef __attribute__((address_space(3))) short4_t af_lds_s4;

struct AttnFullParams {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o; float* lse;
  long q_bs, k_bs, v_bs, o_bs;     // batch strides (elements)
  int ldq, ldk, ldv, ldo;          // token strides (elements)
  int B, H, Sq, Sk, D;
  float scale;
  int frag_per_wg;                 // query fragments (16 rows) per workgroup of one (batch, head)
};

template <int DP>
__device__ __forceinline__ short8_t af_frag_rm(const bf16_t* img, int j, int kk, int lr, int lg) {
  return *reinterpret_cast<const short8_t*>(img + (16 * j + lr) * (DP + 16) + kk * 32 + lg * 8);
}
// A operand of V^T.P^T out of the key-major V image (see attn_seq.hip as_frag_tr)
template <int DP>
__device__ __forceinline__ short8_t af_frag_tr(const bf16_t* img, int jd, int c, int lr, int lg) {
  const bf16_t* p = img + (32 * c + 4 * lg + (lr >> 2)) * (DP + 16) + 16 * jd + 4 * (lr & 3);
  const short4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((af_lds_s4*)p);
  const short4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((af_lds_s4*)(p + 16 * (DP + 16)));
  return (short8_t){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

#define AF_NEG_INF (-__builtin_inff())

template <int DP, int MAXROWS>
__global__ __launch_bounds__(AF_NW * 64) void attn_full_fwd_kernel(AttnFullParams p) {
  constexpr int RS = DP + 16, NKK = DP / 32, NJD = DP / 16, CPR = DP / 8;      // row stride, k-steps, output blocks, 16-B chunks per row
  constexpr int ITEMS = (MAXROWS * CPR + AF_NW * 64 - 1) / (AF_NW * 64);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Vs = Ks + MAXROWS * RS;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int Sk = p.Sk, Sq = p.Sq, D = p.D;
  const int rows = ((Sk + 31) >> 5) << 5;                                     // keys are walked in chunks of 32
  const bf16_t* qb = p.q + (long)b * p.q_bs + h * D;
  const bf16_t* kb = p.k + (long)b * p.k_bs + h * D;
  const bf16_t* vb = p.v + (long)b * p.v_bs + h * D;

  // every global load of the staging and of this wave's query rows is issued before the first LDS write
  short8_t kr[ITEMS], vr[ITEMS];
#pragma unroll
  for (int u = 0; u < ITEMS; ++u) {
    const int it = threadIdx.x + u * AF_NW * 64, row = it / CPR, c = (it - row * CPR) * 8;
    kr[u] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
    vr[u] = kr[u];
    if (row < Sk && c < D) {
      kr[u] = *reinterpret_cast<const short8_t*>(kb + (long)row * p.ldk + c);
      vr[u] = *reinterpret_cast<const short8_t*>(vb + (long)row * p.ldv + c);
    }
  }
  const int f_lo = blockIdx.y * p.frag_per_wg;
  const int nfq = (Sq + 15) >> 4;
  short8_t qa[NKK], qn[NKK];
  int fown[2];
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    int f = f_lo + wave + which * AF_NW;
    if (wave + which * AF_NW >= p.frag_per_wg || f >= nfq) f = -1;
    fown[which] = f;
    const int row = f >= 0 ? 16 * f + lr : Sq;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      short8_t t = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
      const int c = kk * 32 + lg * 8;
      if (row < Sq && c < D) t = *reinterpret_cast<const short8_t*>(qb + (long)row * p.ldq + c);
      if (which) qn[kk] = t; else qa[kk] = t;
    }
  }
#pragma unroll
  for (int u = 0; u < ITEMS; ++u) {
    const int it = threadIdx.x + u * AF_NW * 64, row = it / CPR, c = (it - row * CPR) * 8;
    if (row < rows) {
      *reinterpret_cast<short8_t*>(Ks + row * RS + c) = kr[u];
      *reinterpret_cast<short8_t*>(Vs + row * RS + c) = vr[u];
    }
  }
  __syncthreads();

  const int nchunks = rows >> 5;
#pragma unroll 1
  for (int which = 0; which < 2; ++which) {
    const int f = fown[which];
    if (f >= 0) {
      const int qi = 16 * f + lr;
      float4_t acc[NJD];
#pragma unroll
      for (int jd = 0; jd < NJD; ++jd) acc[jd] = (float4_t){0.f, 0.f, 0.f, 0.f};
      float mrun = AF_NEG_INF, lsum = 0.f;
#pragma unroll 1
      for (int c = 0; c < nchunks; ++c) {
        float4_t s[2];
        float tmax = AF_NEG_INF;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int j = 2 * c + u;
          s[u] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < NKK; ++kk)
            s[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af_frag_rm<DP>(Ks, j, kk, lr, lg), qa[kk], s[u], 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 16 * j + 4 * lg + r;
            s[u][r] = key < Sk ? s[u][r] * p.scale : AF_NEG_INF;
            tmax = fmaxf(tmax, s[u][r]);
          }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(mrun, tmax);
        const float alpha = (m_new == AF_NEG_INF) ? 1.f : __expf(mrun - m_new);
        float psum = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = (m_new == AF_NEG_INF) ? 0.f : __expf(s[u][r] - m_new);
            s[u][r] = e;
            psum += e;
          }
        lsum = lsum * alpha + psum;
        mrun = m_new;
        const short8_t pb = {(short)f2bf(s[0][0]), (short)f2bf(s[0][1]), (short)f2bf(s[0][2]), (short)f2bf(s[0][3]),
                             (short)f2bf(s[1][0]), (short)f2bf(s[1][1]), (short)f2bf(s[1][2]), (short)f2bf(s[1][3])};
#pragma unroll
        for (int jd = 0; jd < NJD; ++jd) {
          acc[jd] *= alpha;
          acc[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af_frag_tr<DP>(Vs, jd, c, lr, lg), pb, acc[jd], 0, 0, 0);
        }
      }
      lsum += __shfl_xor(lsum, 16, 64);
      lsum += __shfl_xor(lsum, 32, 64);
      if (qi < Sq) {
        const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
        bf16_t* ob = p.o + (long)b * p.o_bs + (long)qi * p.ldo + h * D;
#pragma unroll
        for (int jd = 0; jd < NJD; ++jd) {
          const int d = jd * 16 + lg * 4;
          if (d < D) {                                            // D % 8 == 0: a 4-element group is inside or outside the head
            uint2 pk;
            pk.x = pack_bf2(acc[jd][0] * inv, acc[jd][1] * inv);
            pk.y = pack_bf2(acc[jd][2] * inv, acc[jd][3] * inv);
            *reinterpret_cast<uint2*>(ob + d) = pk;
          }
        }
        if (p.lse && lg == 0) p.lse[((long)b * p.H + h) * Sq + qi] = (lsum > 0.f) ? mrun + __logf(lsum) : AF_NEG_INF;
      }
    }
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) qa[kk] = qn[kk];
  }
}

// MH_ERR_UNSUPPORTED: the caller (mh_attn_fwd) runs the tiled kernel
template <int DP, int MAXROWS>
static int af_launch(const AttnFullParams& p0, hipStream_t stream) {
  AttnFullParams p = p0;
  const int nfq = (p.Sq + 15) >> 4;
  // query fragments per workgroup: as few as give every CU a workgroup (<= 256 in all), at most 16 (two per wave)
  int per = nfq;
  while (per > 1 && (long)p.B * p.H * ((nfq + (per - 1) - 1) / (per - 1)) <= 256) --per;
  if (per > 16) per = 16;
  p.frag_per_wg = per;
  const int gy = (nfq + per - 1) / per;
  const size_t lds = (size_t)2 * MAXROWS * (DP + 16) * 2;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_full_fwd_kernel<DP, MAXROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((attn_full_fwd_kernel<DP, MAXROWS>), dim3(p.B * p.H, gy), dim3(AF_NW * 64), lds, stream, p);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

int mh_launch_attn_full_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Sq, int Sk, int D,
                            long q_bs, int ldq, long k_bs, int ldk, long v_bs, int ldv, long o_bs, int ldo, float scale,
                            hipStream_t stream) {
  if (D % 8 || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4 || Sk > 288 || Sq <= 0 || Sk <= 0) return MH_ERR_UNSUPPORTED;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15 || ((uintptr_t)o & 7)) return MH_ERR_UNSUPPORTED;
  if ((D * 2) % 16 || (q_bs % 8) || (k_bs % 8) || (v_bs % 8) || (o_bs % 4)) return MH_ERR_UNSUPPORTED;   // 16-byte row chunks of every head; O is stored as 8-byte words
  AttnFullParams p = {(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse, q_bs, k_bs, v_bs, o_bs,
                      ldq, ldk, ldv, ldo, B, H, Sq, Sk, D, scale, 0};
  if (D > 64 && D <= 96) return af_launch<96, 288>(p, stream);
  if (D > 32 && D <= 64) return af_launch<64, 288>(p, stream);
  return MH_ERR_UNSUPPORTED;
}

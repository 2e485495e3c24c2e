// K3/K4/K5: fused multi-head attention, forward + backward, MFMA-tiled with LDS-staged K/V (fwd, dQ) or
// Q/dO (dK/dV).  One kernel family covers
//   * EVA-ViT self-attention (head_dim 88 padded to 96 in LDS, optional additive rel-pos "window" bias,
//     reference eva_vit.py:118-148),
//   * Q-Former self- and cross-attention (head_dim 64, Sq != Sk, reference Qformer.py:169-275),
//   * LLaMA causal attention with right-padding mask and the KV-cache decode case (head_dim 128,
//     reference modeling_llama.py:168-231; masked scores behave as finfo.min there == exp()->0 here).
//
// Layout: Q/K/V/O/dO are token-major [B, S, ld] bf16 with head h at columns [h*D, (h+1)*D) -- exactly what
// the QKV projection GEMM writes and the output projection GEMM reads (no head split/merge kernels).
//
// gfx950 mapping (wave64, v_mfma_f32_16x16x32_bf16, 4 waves / workgroup, 64x64 tiles):
//   fwd/dQ compute S^T = K.Q^T so that each lane owns ONE query column (l&15) and 16 keys: the softmax row
//   statistics are per-lane scalars (+2 cross-lane xor-shuffles), and the probabilities sit in registers
//   already in MFMA B-operand layout for O^T = V^T.P^T / dQ^T = K^T.dS^T -- P never touches LDS.
//   dK/dV computes S = Q.K^T (lane owns one key column) for the same reason.  The only transposed LDS images
//   are V^T / K^T / Q^T / dO^T, built while staging.  Softmax statistics and accumulators are fp32.
#include "common.h"

#define TQ 64
#define TK 64

struct AttnParams {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  bf16_t* o;          // fwd out / bwd: unused
  const bf16_t* dout; // bwd
  bf16_t* dq;
  bf16_t* dk;
  bf16_t* dv;
  float* lse;          // [B,H,Sq]
  float* delta;        // [B,H,Sq] (bwd): written by the dQ kernel, read by the dK|dV kernel
  const float* bias;   // optional additive [H,Sq,Sk] fp32
  const int* kv_len;   // optional [B] valid key count (right padding)
  int B, H, Sq, Sk, D;
  long q_bs, k_bs, v_bs, o_bs;  // batch strides (elements)
  int ldq, ldk, ldv, ldo;       // token strides (elements)
  long dq_bs, dk_bs, dv_bs, do_bs;
  int lddq, lddk, lddv, lddo;
  float scale;
  int causal;
  int q_off;  // key j visible to query i iff j <= i + q_off (causal)
  // decode token with the rotary + KV append fused in (mh_attn_decode_rope): q / k / v of the new token are columns
  // [0, W), [W, 2W), [2W, 3W) of qkv rows (W = H*D); the cache row is [k | v]; all NULL / 0 otherwise
  bf16_t* qkv;
  long ld_qkv;
  const float* cs;
  const float* sn;
  const int* pos;       // [B] rotary position of the new token
  const int* pos_dev;   // [1] cache row the new token is appended at
};

template <int DP>
struct Lds {
  static constexpr int ROW = DP + 8;   // row-major tile row stride (elements)
  static constexpr int TROW = TK + 8;  // transposed tile row stride (elements)
  static constexpr int RM_BYTES = 64 * ROW * 2;
  static constexpr int TR_BYTES = DP * TROW * 2;
};

// stage a [64 x D] tile (rows r0.., head column offset already applied to base) row-major into LDS, zero padded
template <int DP>
__device__ __forceinline__ void stage_rowmajor(bf16_t* lds, const bf16_t* base, long ld, int r0, int nrows, int D) {
  constexpr int CH = DP / 8;
  for (int idx = threadIdx.x; idx < 64 * CH; idx += 256) {
    const int row = idx / CH, ch = idx - row * CH;
    short8_t v = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
    if (r0 + row < nrows && ch * 8 < D) v = *reinterpret_cast<const short8_t*>(base + (long)(r0 + row) * ld + ch * 8);
    *reinterpret_cast<short8_t*>(lds + row * Lds<DP>::ROW + ch * 8) = v;
  }
}
// stage the same tile transposed: lds[d][row]
template <int DP>
__device__ __forceinline__ void stage_transposed(bf16_t* lds, const bf16_t* base, long ld, int r0, int nrows, int D) {
  constexpr int CH = DP / 8;
  for (int idx = threadIdx.x; idx < 64 * CH; idx += 256) {
    const int row = idx & 63, ch = idx >> 6;
    short8_t v = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
    if (r0 + row < nrows && ch * 8 < D) v = *reinterpret_cast<const short8_t*>(base + (long)(r0 + row) * ld + ch * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) lds[(ch * 8 + e) * Lds<DP>::TROW + row] = (bf16_t)v[e];
  }
}

// one global read of the tile, both LDS images (row-major for the S = X.Y^T operand, transposed for the X^T.dS one);
// lanes run along rows so the 2-byte transposed stores are conflict-free, and the 16-byte row-major stores of 8
// consecutive rows land on 8 different bank quads (row stride DP+8 elements)
template <int DP>
__device__ __forceinline__ void stage_both(bf16_t* lds_rm, bf16_t* lds_tr, const bf16_t* base, long ld, int r0, int nrows,
                                           int D) {
  constexpr int CH = DP / 8;
  for (int idx = threadIdx.x; idx < 64 * CH; idx += 256) {
    const int row = idx & 63, ch = idx >> 6;
    short8_t v = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
    if (r0 + row < nrows && ch * 8 < D) v = *reinterpret_cast<const short8_t*>(base + (long)(r0 + row) * ld + ch * 8);
    *reinterpret_cast<short8_t*>(lds_rm + row * Lds<DP>::ROW + ch * 8) = v;
#pragma unroll
    for (int e = 0; e < 8; ++e) lds_tr[(ch * 8 + e) * Lds<DP>::TROW + row] = (bf16_t)v[e];
  }
}

// A-operand fragment from a row-major tile: row = 16*j + (lane&15), k-chunk (kk*4 + lane>>4)
template <int DP>
__device__ __forceinline__ short8_t frag_rm(const bf16_t* lds, int j, int kk, int lr, int lg) {
  return *reinterpret_cast<const short8_t*>(lds + (16 * j + lr) * Lds<DP>::ROW + kk * 32 + lg * 8);
}
// A-operand fragment from a transposed tile: row d = 16*jd + (lane&15); reduction elements
// {32c+4g+r} U {32c+16+4g+r}, r=0..3 -- matches the register order of a packed S^T / S accumulator pair.
template <int DP>
__device__ __forceinline__ short8_t frag_tr(const bf16_t* lds, int jd, int c, int lr, int lg) {
  const bf16_t* p = lds + (16 * jd + lr) * Lds<DP>::TROW + 32 * c + 4 * lg;
  const short4_t a = *reinterpret_cast<const short4_t*>(p);
  const short4_t b = *reinterpret_cast<const short4_t*>(p + 16);
  return (short8_t){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ short8_t pack8(const float4_t& a, const float4_t& b) {
  return (short8_t){(short)f2bf(a[0]), (short)f2bf(a[1]), (short)f2bf(a[2]), (short)f2bf(a[3]),
                    (short)f2bf(b[0]), (short)f2bf(b[1]), (short)f2bf(b[2]), (short)f2bf(b[3])};
}
// B-operand fragments straight from global: token row `row` (or zeros), chunk kk*32 + g*8
template <int DP>
__device__ __forceinline__ void load_row_frags(short8_t (&f)[DP / 32], const bf16_t* base, long ld, int row, int nrows,
                                               int D, int lg) {
#pragma unroll
  for (int kk = 0; kk < DP / 32; ++kk) {
    const int col = kk * 32 + lg * 8;
    f[kk] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
    if (row < nrows && col < D) f[kk] = *reinterpret_cast<const short8_t*>(base + (long)row * ld + col);
  }
}

#define NEG_INF (-__builtin_inff())

// ------------------------------------------------------------------------------------------- forward
template <int DP>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Vt = reinterpret_cast<bf16_t*>(smem + Lds<DP>::RM_BYTES);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
  const int q0 = blockIdx.x * TQ;
  const int qi = q0 + wave * 16 + lr;  // this lane's query row
  const bf16_t* qb = p.q + b * p.q_bs + h * p.D;
  const bf16_t* kb = p.k + b * p.k_bs + h * p.D;
  const bf16_t* vb = p.v + b * p.v_bs + h * p.D;
  int kv_valid = p.kv_len ? p.kv_len[b] : p.Sk;
  kv_valid = kv_valid < p.Sk ? kv_valid : p.Sk;
  int kv_end = kv_valid;
  if (p.causal) {
    const int lim = q0 + TQ - 1 + p.q_off + 1;
    kv_end = kv_end < lim ? kv_end : lim;
  }
  short8_t qf[DP / 32];
  load_row_frags<DP>(qf, qb, p.ldq, qi, p.Sq, p.D, lg);

  float4_t acc[DP / 16];
#pragma unroll
  for (int jd = 0; jd < DP / 16; ++jd) acc[jd] = (float4_t){0.f, 0.f, 0.f, 0.f};
  float m = NEG_INF, lsum = 0.f;

  for (int k0 = 0; k0 < kv_end; k0 += TK) {
    __syncthreads();
    // rows >= kv_valid are zero-filled: an unwritten KV-cache row may hold NaN/Inf and 0 * NaN = NaN in the MFMA
    stage_rowmajor<DP>(Ks, kb, p.ldk, k0, kv_valid, p.D);
    stage_transposed<DP>(Vt, vb, p.ldv, k0, kv_valid, p.D);
    __syncthreads();
    float4_t s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < DP / 32; ++kk)
        s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm<DP>(Ks, j, kk, lr, lg), qf[kk], s[j], 0, 0, 0);
    }
    float tmax = NEG_INF;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * j + 4 * lg + r;
        const bool ok = key < kv_valid && (!p.causal || key <= qi + p.q_off) && qi < p.Sq;
        float val = s[j][r] * p.scale;
        if (p.bias && ok) val += p.bias[((long)h * p.Sq + qi) * p.Sk + key];
        s[j][r] = ok ? val : NEG_INF;
        tmax = fmaxf(tmax, s[j][r]);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m, tmax);
    const float alpha = (m_new == NEG_INF) ? 1.f : __expf(m - m_new);
    float psum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = (m_new == NEG_INF) ? 0.f : __expf(s[j][r] - m_new);
        s[j][r] = e;
        psum += e;
      }
    lsum = lsum * alpha + psum;
    m = m_new;
    const short8_t pb0 = pack8(s[0], s[1]), pb1 = pack8(s[2], s[3]);
#pragma unroll
    for (int jd = 0; jd < DP / 16; ++jd) {
      acc[jd] *= alpha;
      acc[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DP>(Vt, jd, 0, lr, lg), pb0, acc[jd], 0, 0, 0);
      acc[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DP>(Vt, jd, 1, lr, lg), pb1, acc[jd], 0, 0, 0);
    }
  }
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  if (qi < p.Sq) {
    const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
    bf16_t* ob = p.o + b * p.o_bs + (long)qi * p.ldo + h * p.D;
#pragma unroll
    for (int jd = 0; jd < DP / 16; ++jd) {
      const int d = jd * 16 + lg * 4;
      if (d < p.D) {
        uint2 pk;
        pk.x = pack_bf2(acc[jd][0] * inv, acc[jd][1] * inv);
        pk.y = pack_bf2(acc[jd][2] * inv, acc[jd][3] * inv);
        *reinterpret_cast<uint2*>(ob + d) = pk;
      }
    }
    if (p.lse && lg == 0) p.lse[((long)b * p.H + h) * p.Sq + qi] = (lsum > 0.f) ? m + __logf(lsum) : NEG_INF;
  }
}

// ------------------------------------------------------------------------------------------- dQ
template <int DP>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Vs = reinterpret_cast<bf16_t*>(smem + Lds<DP>::RM_BYTES);
  bf16_t* Kt = reinterpret_cast<bf16_t*>(smem + 2 * Lds<DP>::RM_BYTES);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
  const int q0 = blockIdx.x * TQ;
  const int qi = q0 + wave * 16 + lr;
  const bf16_t* qb = p.q + b * p.q_bs + h * p.D;
  const bf16_t* kb = p.k + b * p.k_bs + h * p.D;
  const bf16_t* vb = p.v + b * p.v_bs + h * p.D;
  const bf16_t* gb = p.dout + b * p.do_bs + h * p.D;
  int kv_valid = p.kv_len ? p.kv_len[b] : p.Sk;
  kv_valid = kv_valid < p.Sk ? kv_valid : p.Sk;
  int kv_end = kv_valid;
  if (p.causal) {
    const int lim = q0 + TQ - 1 + p.q_off + 1;
    kv_end = kv_end < lim ? kv_end : lim;
  }
  short8_t qf[DP / 32], gf[DP / 32];
  load_row_frags<DP>(qf, qb, p.ldq, qi, p.Sq, p.D, lg);
  load_row_frags<DP>(gf, gb, p.lddo, qi, p.Sq, p.D, lg);
  const long sidx = ((long)b * p.H + h) * p.Sq + qi;
  const float lse = qi < p.Sq ? p.lse[sidx] : 0.f;
  // delta[b,h,q] = sum_d dO * O for this lane's row: the four lanes of a row (lg = 0..3) hold disjoint column groups of dO
  // already; the same groups of O are read here and the partial sums combined across the lanes.  Written out for the
  // dK|dV kernel, which runs next on the stream (this replaces a separate delta launch per attention backward).
  float dlt = 0.f;
  {
    short8_t of[DP / 32];
    load_row_frags<DP>(of, p.o + b * p.o_bs + h * p.D, p.ldo, qi, p.Sq, p.D, lg);
#pragma unroll
    for (int kk = 0; kk < DP / 32; ++kk)
#pragma unroll
      for (int e = 0; e < 8; ++e) dlt += bf2f((bf16_t)of[kk][e]) * bf2f((bf16_t)gf[kk][e]);
    dlt += __shfl_xor(dlt, 16);
    dlt += __shfl_xor(dlt, 32);
    if (lg == 0 && qi < p.Sq) p.delta[sidx] = dlt;
  }

  float4_t acc[DP / 16];
#pragma unroll
  for (int jd = 0; jd < DP / 16; ++jd) acc[jd] = (float4_t){0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < kv_end; k0 += TK) {
    __syncthreads();
    stage_both<DP>(Ks, Kt, kb, p.ldk, k0, kv_valid, p.D);
    stage_rowmajor<DP>(Vs, vb, p.ldv, k0, kv_valid, p.D);
    __syncthreads();
    float4_t s[4], dp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = (float4_t){0.f, 0.f, 0.f, 0.f};
      dp[j] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < DP / 32; ++kk) {
        s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm<DP>(Ks, j, kk, lr, lg), qf[kk], s[j], 0, 0, 0);
        dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm<DP>(Vs, j, kk, lr, lg), gf[kk], dp[j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * j + 4 * lg + r;
        const bool ok = key < kv_valid && (!p.causal || key <= qi + p.q_off) && qi < p.Sq;
        float val = s[j][r] * p.scale;
        if (p.bias && ok) val += p.bias[((long)h * p.Sq + qi) * p.Sk + key];
        const float pr = ok ? __expf(val - lse) : 0.f;
        s[j][r] = ok ? pr * (dp[j][r] - dlt) * p.scale : 0.f;  // dS (select, not multiply: masked dp may be non-finite)
      }
    const short8_t d0 = pack8(s[0], s[1]), d1 = pack8(s[2], s[3]);
#pragma unroll
    for (int jd = 0; jd < DP / 16; ++jd) {
      acc[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DP>(Kt, jd, 0, lr, lg), d0, acc[jd], 0, 0, 0);
      acc[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DP>(Kt, jd, 1, lr, lg), d1, acc[jd], 0, 0, 0);
    }
  }
  if (qi < p.Sq) {
    bf16_t* ob = p.dq + b * p.dq_bs + (long)qi * p.lddq + h * p.D;
#pragma unroll
    for (int jd = 0; jd < DP / 16; ++jd) {
      const int d = jd * 16 + lg * 4;
      if (d < p.D) {
        uint2 pk;
        pk.x = pack_bf2(acc[jd][0], acc[jd][1]);
        pk.y = pack_bf2(acc[jd][2], acc[jd][3]);
        *reinterpret_cast<uint2*>(ob + d) = pk;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------- dK, dV
template <int DP>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Gs = reinterpret_cast<bf16_t*>(smem + Lds<DP>::RM_BYTES);
  bf16_t* Qt = reinterpret_cast<bf16_t*>(smem + 2 * Lds<DP>::RM_BYTES);
  bf16_t* Gt = reinterpret_cast<bf16_t*>(smem + 2 * Lds<DP>::RM_BYTES + Lds<DP>::TR_BYTES);
  float* lse_s = reinterpret_cast<float*>(smem + 2 * Lds<DP>::RM_BYTES + 2 * Lds<DP>::TR_BYTES);
  float* dlt_s = lse_s + 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
  const int k0 = blockIdx.x * TK;
  const int ki = k0 + wave * 16 + lr;  // this lane's key column
  const bf16_t* qb = p.q + b * p.q_bs + h * p.D;
  const bf16_t* kb = p.k + b * p.k_bs + h * p.D;
  const bf16_t* vb = p.v + b * p.v_bs + h * p.D;
  const bf16_t* gb = p.dout + b * p.do_bs + h * p.D;
  int kv_valid = p.kv_len ? p.kv_len[b] : p.Sk;
  kv_valid = kv_valid < p.Sk ? kv_valid : p.Sk;
  short8_t kf[DP / 32], vf[DP / 32];
  load_row_frags<DP>(kf, kb, p.ldk, ki, p.Sk, p.D, lg);
  load_row_frags<DP>(vf, vb, p.ldv, ki, p.Sk, p.D, lg);

  float4_t adk[DP / 16], adv[DP / 16];
#pragma unroll
  for (int jd = 0; jd < DP / 16; ++jd) {
    adk[jd] = (float4_t){0.f, 0.f, 0.f, 0.f};
    adv[jd] = (float4_t){0.f, 0.f, 0.f, 0.f};
  }
  int q_start = 0;
  if (p.causal) {
    const int first = k0 - p.q_off;  // smallest query that can see key k0
    q_start = first > 0 ? (first / TQ) * TQ : 0;
  }
  const bool key_ok = ki < kv_valid;
  const long sbase = ((long)b * p.H + h) * p.Sq;
  for (int q0 = q_start; q0 < p.Sq; q0 += TQ) {
    __syncthreads();
    stage_both<DP>(Qs, Qt, qb, p.ldq, q0, p.Sq, p.D);
    stage_both<DP>(Gs, Gt, gb, p.lddo, q0, p.Sq, p.D);
    if (threadIdx.x < 64) {
      const int qq = q0 + threadIdx.x;
      lse_s[threadIdx.x] = qq < p.Sq ? p.lse[sbase + qq] : 1e30f;
      dlt_s[threadIdx.x] = qq < p.Sq ? p.delta[sbase + qq] : 0.f;
    }
    __syncthreads();
    float4_t s[4], dp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = (float4_t){0.f, 0.f, 0.f, 0.f};
      dp[j] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < DP / 32; ++kk) {
        s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm<DP>(Qs, j, kk, lr, lg), kf[kk], s[j], 0, 0, 0);
        dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm<DP>(Gs, j, kk, lr, lg), vf[kk], dp[j], 0, 0, 0);
      }
    }
    float4_t pr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4_t l4 = *reinterpret_cast<const float4_t*>(lse_s + 16 * j + 4 * lg);
      const float4_t d4 = *reinterpret_cast<const float4_t*>(dlt_s + 16 * j + 4 * lg);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = q0 + 16 * j + 4 * lg + r;
        const bool ok = key_ok && qq < p.Sq && (!p.causal || ki <= qq + p.q_off);
        float val = s[j][r] * p.scale;
        if (p.bias && ok) val += p.bias[((long)h * p.Sq + qq) * p.Sk + ki];
        const float e = ok ? __expf(val - l4[r]) : 0.f;
        pr[j][r] = e;
        s[j][r] = ok ? e * (dp[j][r] - d4[r]) * p.scale : 0.f;
      }
    }
    const short8_t p0 = pack8(pr[0], pr[1]), p1 = pack8(pr[2], pr[3]);
    const short8_t d0 = pack8(s[0], s[1]), d1 = pack8(s[2], s[3]);
#pragma unroll
    for (int jd = 0; jd < DP / 16; ++jd) {
      adv[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DP>(Gt, jd, 0, lr, lg), p0, adv[jd], 0, 0, 0);
      adv[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DP>(Gt, jd, 1, lr, lg), p1, adv[jd], 0, 0, 0);
      adk[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DP>(Qt, jd, 0, lr, lg), d0, adk[jd], 0, 0, 0);
      adk[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DP>(Qt, jd, 1, lr, lg), d1, adk[jd], 0, 0, 0);
    }
  }
  if (ki < p.Sk) {
    bf16_t* okp = p.dk + b * p.dk_bs + (long)ki * p.lddk + h * p.D;
    bf16_t* ovp = p.dv + b * p.dv_bs + (long)ki * p.lddv + h * p.D;
#pragma unroll
    for (int jd = 0; jd < DP / 16; ++jd) {
      const int d = jd * 16 + lg * 4;
      if (d < p.D) {
        uint2 pk;
        pk.x = pack_bf2(adk[jd][0], adk[jd][1]);
        pk.y = pack_bf2(adk[jd][2], adk[jd][3]);
        *reinterpret_cast<uint2*>(okp + d) = pk;
        pk.x = pack_bf2(adv[jd][0], adv[jd][1]);
        pk.y = pack_bf2(adv[jd][2], adv[jd][3]);
        *reinterpret_cast<uint2*>(ovp + d) = pk;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------- C ABI
// ---- Sq == 1 (KV-cache decode, modeling_llama.py:197-222 with past_key_value): one 16-wave workgroup per (b, h).
// The MFMA tile kernel above spends ~19 us on this shape (one 64-row query tile with a single live row, K/V staged
// through LDS tile by tile).  There are only B*H workgroups, so the kernel is a chain of load latencies: the work is
// spread over 1024 threads to keep that chain short (a 4-wave version with one load per pass measured 16.5 us):
//   scores: 16 lanes share a key (16 B of the row each), 64 keys per workgroup pass, two passes in flight -> LDS
//   softmax: every wave reduces the LDS scores itself (len <= a few hundred)
//   PV: lane owns two head dims, waves take keys round-robin, up to 8 independent 256-B row loads in flight per wave
// fp32 throughout (the tile kernel rounds P to bf16 for its MFMA), output rounded to bf16 once.
#define DNW 16
__global__ __launch_bounds__(DNW * 64) void attn_decode_kernel(AttnParams p) {
  extern __shared__ float dsm[];                 // [Sk] scores, then [DNW][128] partial outputs
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int len = p.kv_len ? p.kv_len[b] : p.Sk;
  len = len < p.Sk ? len : p.Sk;
  const int D = p.D;
  if (p.qkv) {
    // rotary on this head's q (in place) and k, k | v into the cache row -- what rope_kv_append_kernel does for the whole
    // token (modeling_llama.py:186-195), same arithmetic (fp32 rotate-half, one rounding to bf16), per (b, h) here
    const int half = D >> 1, items = half >> 2, W = p.H * D;
    bf16_t* src = p.qkv + (size_t)b * p.ld_qkv;
    bf16_t* crow = const_cast<bf16_t*>(p.k) + (size_t)b * p.k_bs + (size_t)p.pos_dev[0] * p.ldk;
    if (tid < 2 * items) {
      const int which = tid / items, i = (tid % items) * 4;
      bf16_t* e = src + which * W + h * D + i;
      const int ps = p.pos[b];
      const float4_t c4 = *reinterpret_cast<const float4_t*>(p.cs + (size_t)ps * half + i);
      const float4_t s4 = *reinterpret_cast<const float4_t*>(p.sn + (size_t)ps * half + i);
      const short4_t a = *reinterpret_cast<const short4_t*>(e);
      const short4_t bb = *reinterpret_cast<const short4_t*>(e + half);
      short4_t oa, ob;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float x1 = bf2f((bf16_t)a[t]), x2 = bf2f((bf16_t)bb[t]);
        oa[t] = (short)f2bf(x1 * c4[t] - x2 * s4[t]);
        ob[t] = (short)f2bf(x2 * c4[t] + x1 * s4[t]);
      }
      bf16_t* dst = which == 0 ? e : crow + h * D + i;
      *reinterpret_cast<short4_t*>(dst) = oa;
      *reinterpret_cast<short4_t*>(dst + half) = ob;
    } else if (tid < 2 * items + (D >> 3)) {
      const int c = (tid - 2 * items) * 8;
      *reinterpret_cast<short8_t*>(crow + W + h * D + c) = *reinterpret_cast<const short8_t*>(src + 2 * W + h * D + c);
    }
    __syncthreads();                               // the row is read back below by other waves of this workgroup
  }
  const bf16_t* qp = p.q + (size_t)b * p.q_bs + h * D;
  const bf16_t* kp = p.k + (size_t)b * p.k_bs + h * D;
  const bf16_t* vp = p.v + (size_t)b * p.v_bs + h * D;
  float* sc = dsm;
  float* part = dsm + ((p.Sk + 63) & ~63);
  // phase 1: scores
  const int sub = lane & 15, kq = lane >> 4;     // 16 lanes per key, dims sub*8 .. +7
  float qf[8];
  const bool dim_ok = sub * 8 < D;
  {
    short8_t qv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (dim_ok) qv = *reinterpret_cast<const short8_t*>(qp + sub * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[e] = bf2f((bf16_t)qv[e]) * p.scale;
  }
  for (int j0 = 0; j0 < len; j0 += 2 * DNW * 4) {
    short8_t kv[2];
    int jj[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      jj[u] = j0 + u * DNW * 4 + wave * 4 + kq;
      kv[u] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
      if (jj[u] < len && dim_ok) kv[u] = *reinterpret_cast<const short8_t*>(kp + (size_t)jj[u] * p.ldk + sub * 8);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += qf[e] * bf2f((bf16_t)kv[u][e]);
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      s += __shfl_xor(s, 8, 64);
      if (sub == 0 && jj[u] < len) sc[jj[u]] = s;
    }
  }
  __syncthreads();
  // phase 2: softmax statistics (each wave for itself)
  float mx = -INFINITY;
  for (int j = lane; j < len; j += 64) mx = fmaxf(mx, sc[j]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < len; j += 64) sum += __expf(sc[j] - mx);
  sum = wave_sum(sum);
  // phase 3: o = sum_j p_j V[j]; lane owns dims 2*lane, 2*lane+1; wave w takes keys w, w+DNW, ...
  float o0 = 0.f, o1 = 0.f;
  const bool own = 2 * lane < D;
  int j = wave;
  for (; j + 7 * DNW < len; j += 8 * DNW) {
    unsigned vv[8];
    float pj[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      vv[u] = own ? *reinterpret_cast<const unsigned*>(vp + (size_t)(j + DNW * u) * p.ldv + 2 * lane) : 0u;
      pj[u] = __expf(sc[j + DNW * u] - mx);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      o0 += pj[u] * bf2f((bf16_t)(vv[u] & 0xffffu));
      o1 += pj[u] * bf2f((bf16_t)(vv[u] >> 16));
    }
  }
  {                                              // remainder: up to 7 keys, loads issued together
    unsigned vv[7];
    float pj[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int jr = j + DNW * u;
      const bool ok = jr < len;
      vv[u] = (own && ok) ? *reinterpret_cast<const unsigned*>(vp + (size_t)jr * p.ldv + 2 * lane) : 0u;
      pj[u] = ok ? __expf(sc[jr] - mx) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      o0 += pj[u] * bf2f((bf16_t)(vv[u] & 0xffffu));
      o1 += pj[u] * bf2f((bf16_t)(vv[u] >> 16));
    }
  }
  part[wave * 128 + 2 * lane] = o0;
  part[wave * 128 + 2 * lane + 1] = o1;
  __syncthreads();
  if (wave == 0 && own) {
    const float inv = len > 0 ? 1.f / sum : 0.f;
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int w = 0; w < DNW; ++w) {
      r0 += part[w * 128 + 2 * lane];
      r1 += part[w * 128 + 2 * lane + 1];
    }
    *reinterpret_cast<unsigned*>(p.o + (size_t)b * p.o_bs + h * D + 2 * lane) = pack_bf2(r0 * inv, r1 * inv);
  }
  if (p.lse && tid == 0) p.lse[(size_t)b * p.H + h] = len > 0 ? mx + __logf(sum) : -INFINITY;
}

static int check_common(const AttnParams& p) {
  if (p.D % 8 || p.D > 128 || p.D <= 0) return MH_ERR_UNSUPPORTED;
  if (p.ldq % 8 || p.ldk % 8 || p.ldv % 8) return MH_ERR_ARG;
  if (p.B <= 0 || p.H <= 0 || p.Sq <= 0 || p.Sk <= 0) return MH_ERR_ARG;
  return MH_OK;
}

template <void (*KERN)(AttnParams)>
static void allow_lds(size_t bytes) {
  // > 64 KiB of dynamic LDS needs an explicit opt-in (gfx950 has 160 KiB per CU).  Once per kernel instantiation
  // (the static lives in this template instance), so nothing but the launch happens during hipGraph capture.
  static bool done = false;
  if (!done && bytes > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  done = true;
}

template <int DP>
static int launch_fwd(const AttnParams& p, hipStream_t s) {
  const size_t sh = Lds<DP>::RM_BYTES + Lds<DP>::TR_BYTES;
  allow_lds<attn_fwd_kernel<DP>>(sh);
  hipLaunchKernelGGL(attn_fwd_kernel<DP>, dim3((p.Sq + TQ - 1) / TQ, p.B * p.H), dim3(256), sh, s, p);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
template <int DP>
static int launch_bwd(const AttnParams& p, hipStream_t s) {
  const size_t sh_dq = 2 * Lds<DP>::RM_BYTES + Lds<DP>::TR_BYTES;
  allow_lds<attn_bwd_dq_kernel<DP>>(sh_dq);
  hipLaunchKernelGGL(attn_bwd_dq_kernel<DP>, dim3((p.Sq + TQ - 1) / TQ, p.B * p.H), dim3(256), sh_dq, s, p);
  MH_CHECK_LAUNCH();
  const size_t sh_kv = 2 * Lds<DP>::RM_BYTES + 2 * Lds<DP>::TR_BYTES + 512;
  allow_lds<attn_bwd_dkv_kernel<DP>>(sh_kv);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel<DP>, dim3((p.Sk + TK - 1) / TK, p.B * p.H), dim3(256), sh_kv, s, p);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

int mh_launch_attn_full_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Sq, int Sk, int D,
                            long q_bs, int ldq, long k_bs, int ldk, long v_bs, int ldv, long o_bs, int ldo, float scale,
                            hipStream_t stream);

extern "C" int mh_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* bias,
                           const int* kv_len, int B, int H, int Sq, int Sk, int D, long q_bs, int ldq, long k_bs,
                           int ldk, long v_bs, int ldv, long o_bs, int ldo, float scale, int causal,
                           hipStream_t stream) {
  AttnParams p = {};
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o;
  p.lse = lse; p.bias = bias; p.kv_len = kv_len;
  p.B = B; p.H = H; p.Sq = Sq; p.Sk = Sk; p.D = D;
  p.q_bs = q_bs; p.k_bs = k_bs; p.v_bs = v_bs; p.o_bs = o_bs;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.scale = scale; p.causal = causal; p.q_off = Sk - Sq;
  int rc = check_common(p);
  if (rc) return rc;
  if (ldo % 4) return MH_ERR_ARG;
  if (Sq == 1 && !bias && Sk <= 8192 && (ldv % 2) == 0) {   // KV-cache decode (causal or not: the one query sees every valid key)
    const size_t sh = (((size_t)Sk + 63) & ~(size_t)63) * 4 + DNW * 128 * 4;
    hipLaunchKernelGGL(attn_decode_kernel, dim3(B * H), dim3(DNW * 64), sh, stream, p);
    MH_CHECK_LAUNCH();
    return MH_OK;
  }
  // the frozen encoders' unmasked attention with all keys of a (batch, head) in one CU's LDS: attn_full.hip
  if (!bias && !kv_len && !causal && mh_opt(MH_OPT_ATTN_FULL)) {
    rc = mh_launch_attn_full_fwd(q, k, v, o, lse, B, H, Sq, Sk, D, q_bs, ldq, k_bs, ldk, v_bs, ldv, o_bs, ldo, scale, stream);
    if (rc != MH_ERR_UNSUPPORTED) return rc;
  }
  if (D <= 64) return launch_fwd<64>(p, stream);
  if (D <= 96) return launch_fwd<96>(p, stream);
  return launch_fwd<128>(p, stream);
}

// One decode token: rotary on q / k, k | v appended to the cache at row pos_dev[0], attention of the one query over
// kv_len[b] keys -- mh_rope_kv_append + mh_attn_fwd(Sq = 1) in one launch, bit-identical to the pair.
// qkv [B, ld_qkv] bf16 = [q | k | v] (q is rotated in place); cache [B][T_cap][2W] bf16 rows [k | v]; out [B, W] bf16.
extern "C" int mh_attn_decode_rope(void* qkv, long ld_qkv, void* cache, long cache_bstride, long ld_cache, const int* pos,
                                   const int* pos_dev, const int* kv_len, const float* cos_tab, const float* sin_tab, void* out,
                                   long ldo, int B, int H, int D, int T_cap, float scale, hipStream_t stream) {
  if (B <= 0) return MH_OK;
  if (!qkv || !cache || !pos || !pos_dev || !kv_len || !cos_tab || !sin_tab || !out) return MH_ERR_ARG;
  if (D % 8 || D > 128 || (D >> 1) % 4 || ld_qkv % 8 || ld_cache % 8 || cache_bstride % 8 || ldo % 4 || T_cap <= 0 || T_cap > 8192)
    return MH_ERR_ARG;
  const int W = H * D;
  AttnParams p = {};
  p.q = (const bf16_t*)qkv; p.k = (const bf16_t*)cache; p.v = (const bf16_t*)cache + W; p.o = (bf16_t*)out;
  p.kv_len = kv_len; p.B = B; p.H = H; p.Sq = 1; p.Sk = T_cap; p.D = D;
  p.q_bs = ld_qkv; p.k_bs = cache_bstride; p.v_bs = cache_bstride; p.o_bs = ldo;
  p.ldq = (int)ld_qkv; p.ldk = (int)ld_cache; p.ldv = (int)ld_cache; p.ldo = (int)ldo;
  p.scale = scale; p.causal = 0; p.q_off = T_cap - 1;
  p.qkv = (bf16_t*)qkv; p.ld_qkv = ld_qkv; p.cs = cos_tab; p.sn = sin_tab; p.pos = pos; p.pos_dev = pos_dev;
  const size_t sh = (((size_t)T_cap + 63) & ~(size_t)63) * 4 + DNW * 128 * 4;
  hipLaunchKernelGGL(attn_decode_kernel, dim3(B * H), dim3(DNW * 64), sh, stream, p);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                           const float* lse, float* delta_ws, void* dq, void* dk, void* dv, const float* bias,
                           const int* kv_len, int B, int H, int Sq, int Sk, int D, long q_bs, int ldq, long k_bs,
                           int ldk, long v_bs, int ldv, long o_bs, int ldo, long do_bs, int lddo, long dq_bs,
                           int lddq, long dk_bs, int lddk, long dv_bs, int lddv, float scale, int causal,
                           hipStream_t stream) {
  AttnParams p = {};
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.dout = (const bf16_t*)dout;
  p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv;
  p.lse = const_cast<float*>(lse); p.delta = delta_ws; p.bias = bias; p.kv_len = kv_len;
  p.B = B; p.H = H; p.Sq = Sq; p.Sk = Sk; p.D = D;
  p.q_bs = q_bs; p.k_bs = k_bs; p.v_bs = v_bs; p.o_bs = o_bs;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.do_bs = do_bs; p.lddo = lddo; p.dq_bs = dq_bs; p.lddq = lddq; p.dk_bs = dk_bs; p.lddk = lddk;
  p.dv_bs = dv_bs; p.lddv = lddv;
  p.scale = scale; p.causal = causal; p.q_off = Sk - Sq;
  int rc = check_common(p);
  if (rc) return rc;
  if (lddo % 8 || ldo % 8 || lddq % 4 || lddk % 4 || lddv % 4) return MH_ERR_ARG;
  p.o = (bf16_t*)const_cast<void*>(o);                // read-only here: the dQ kernel derives delta from it
  if (D <= 64) return launch_bwd<64>(p, stream);
  if (D <= 96) return launch_bwd<96>(p, stream);
  return launch_bwd<128>(p, stream);
}

// K5 + K8 fused (training path): LLaMA causal self-attention with the rotary embedding applied on the way in, one
// workgroup per (batch, head) holding the WHOLE sequence's K and V in LDS.  Reference modeling_llama.py:109-123
// (apply_rotary_pos_emb), :168-231 (LlamaAttention.forward).
//
// Why a second attention family (round-2 profile): at the step's S = 148 the 64x64-tile kernels of attention.hip are
// latency chains -- 3 query tiles x 256 (b, h) workgroups, each staging K / V^T tile by tile through registers with
// 2-byte transposed LDS stores and two barriers per tile: 24.7 us forward, 81.5 us backward per layer at 4-5 % matrix
// utilisation, plus two rope launches (10 us each) around them.  Here the sequence fits the CU: 148 x 128 bf16 = 37 KB
// per operand, so K and V are staged ONCE (rotary applied to K in the staging registers), every wave owns whole query
// fragments and reads the key-major V through the gfx950 transpose read (ds_read_b64_tr_b16) instead of building a
// V^T image, and the softmax is single-pass (all keys are present: no running max, no rescale).
//
// Layout: qkv is the projection GEMM's token-major output [B, S, ld] bf16 = [q | k | v], head h at columns h*128 ..;
// NOT rotated (the saved tensor stays pre-rotary; the backward kernel rotates again on load and un-rotates dq / dk).
// LDS rows are 144 elements (288 B): a 16-lane ds_read_b128 group and a 32-lane tr_b16 group both land on 64
// distinct banks (row r starts at bank 8r mod 64).
//
// MFMA mapping (v_mfma_f32_16x16x32_bf16, as attention.hip): S^T = K.Q^T puts one query per lane column (lane & 15) and
// keys {16j + 4*(lane>>4) + r} in the accumulator registers, so softmax statistics are per-lane scalars (+2 xor
// shuffles) and P is already the B operand of O^T = V^T.P^T.  The V^T A-operand (row d = lane & 15, reduction elements
// = keys {32c + 4g + r} U {32c + 16 + 4g + r}) is two transpose reads of [4 keys][16 d] blocks of the key-major image.
#include "common.h"
#include <cstdlib>

#define AS_D 128
#define AS_RS 144            // LDS row stride in elements
#define AS_MAXF 10           // query / key fragments of 16 (S <= 160)
#define AS_NW 8              // waves per workgroup

typedef __attribute__((address_space(3))) short4_t as_lds_s4;

struct AttnSeqParams {
  const bf16_t* qkv;     // [B, S, ld]  q | k | v, pre-rotary
  bf16_t* o;             // fwd out [B, S, ldo]
  float* lse;            // [B, H, S]
  const int* pos;        // [B*S] position ids
  const float* cos_tab;  // [max_pos, 64]
  const float* sin_tab;
  const int* kv_len;     // optional [B]
  int B, H, S, ld, ldo;
  float scale;
  // backward
  const bf16_t* o_in;    // [B, S, ldo]
  const float* dout;     // fp32 slabs [nslab][B*S, ldd] (a split-K dgrad's partial sums) or one fp32 matrix
  const bf16_t* dout_bf; // or bf16 [B*S, ldd]
  int nslab; long slab; int ldd;
  bf16_t* dqkv;          // [B, S, ld]  dq | dk | dv (dq, dk un-rotated)
  long long* trace;      // debug: workgroup 0's waves 0 and 7 stamp the 100 MHz counter at the phase boundaries (NULL: off)
};
#define AS_STAMP(i) if (p.trace && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x == 0 || threadIdx.x == 448)) p.trace[(threadIdx.x ? 16 : 0) + (i)] = (long long)__builtin_amdgcn_s_memrealtime();

// rotate-half rotary on one 8-element chunk pair (columns c .. c+7 and c+64 .. c+71), fp32, one rounding
__device__ __forceinline__ void as_rope_pair(short8_t& a, short8_t& b, const float* cs, const float* sn, float sign) {
  const float4_t c0 = *reinterpret_cast<const float4_t*>(cs), c1 = *reinterpret_cast<const float4_t*>(cs + 4);
  const float4_t s0 = *reinterpret_cast<const float4_t*>(sn), s1 = *reinterpret_cast<const float4_t*>(sn + 4);
  short8_t oa, ob;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float c = e < 4 ? c0[e & 3] : c1[e & 3], s = e < 4 ? s0[e & 3] : s1[e & 3];
    const float x1 = bf2f((bf16_t)a[e]), x2 = bf2f((bf16_t)b[e]);
    oa[e] = (short)f2bf(x1 * c - sign * x2 * s);
    ob[e] = (short)f2bf(x2 * c + sign * x1 * s);
  }
  a = oa;
  b = ob;
}

// A operand of the X^T.Y products: X is a key-major (row-major) LDS image, the operand row is column d = 16*jd + lr
// of X and its 8 reduction elements are rows {32c + 4g + r} U {32c + 16 + 4g + r}.  ds_read_b64_tr_b16: lane i of a
// 16-lane group addresses row (i >> 2), columns 4*(i & 3).. of a [4][16] block and receives column i of that block
// (tools/micro/tr_probe.hip).
__device__ __forceinline__ short8_t as_frag_tr(const bf16_t* img, int jd, int c, int lr, int lg) {
  const bf16_t* p = img + (32 * c + 4 * lg + (lr >> 2)) * AS_RS + 16 * jd + 4 * (lr & 3);
  const short4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((as_lds_s4*)p);
  const short4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((as_lds_s4*)(p + 16 * AS_RS));
  return (short8_t){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ short8_t as_frag_rm(const bf16_t* img, int j, int kk, int lr, int lg) {
  return *reinterpret_cast<const short8_t*>(img + (16 * j + lr) * AS_RS + kk * 32 + lg * 8);
}
__device__ __forceinline__ short8_t as_pack8(const float4_t& a, const float4_t& b) {
  return (short8_t){(short)f2bf(a[0]), (short)f2bf(a[1]), (short)f2bf(a[2]), (short)f2bf(a[3]),
                    (short)f2bf(b[0]), (short)f2bf(b[1]), (short)f2bf(b[2]), (short)f2bf(b[3])};
}

// Staging of one head's K (rotated) and V into the LDS images, rows [0, 16 * nfe) with nfe = nf rounded up to even (the
// X^T operands are read in 32-row chunks), rows >= S zero.  All global loads of a thread are issued before the first
// LDS write: a workgroup is alone on its CU, so a load -> write -> load loop would expose one memory latency per
// iteration (8 of them) instead of one or two per kernel.
#define AS_KIT 3             // ceil(160 * 8 / 512) rotary pair items per thread
#define AS_VIT 5             // ceil(160 * 16 / 512) 16-byte items per thread
struct AsStage {
  short8_t ka[AS_KIT], kb[AS_KIT], v[AS_VIT];
  int kps[AS_KIT];
};
__device__ __forceinline__ void as_stage_issue(AsStage& st, const bf16_t* kbase, const bf16_t* vbase, long ld, int S, int rows,
                                               const int* pos) {
#pragma unroll
  for (int u = 0; u < AS_KIT; ++u) {
    const int it = threadIdx.x + u * AS_NW * 64, row = it >> 3, c = (it & 7) * 8;
    st.ka[u] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
    st.kb[u] = st.ka[u];
    st.kps[u] = 0;
    if (row < S) {
      st.kps[u] = pos[row];
      st.ka[u] = *reinterpret_cast<const short8_t*>(kbase + (long)row * ld + c);
      st.kb[u] = *reinterpret_cast<const short8_t*>(kbase + (long)row * ld + c + 64);
    }
  }
#pragma unroll
  for (int u = 0; u < AS_VIT; ++u) {
    const int it = threadIdx.x + u * AS_NW * 64, row = it >> 4, c = (it & 15) * 8;
    st.v[u] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
    if (row < S) st.v[u] = *reinterpret_cast<const short8_t*>(vbase + (long)row * ld + c);
  }
}
__device__ __forceinline__ void as_stage_commit(AsStage& st, bf16_t* kimg, bf16_t* vimg, int S, int rows, const float* cs,
                                                const float* sn) {
#pragma unroll
  for (int u = 0; u < AS_KIT; ++u) {
    const int it = threadIdx.x + u * AS_NW * 64, row = it >> 3, c = (it & 7) * 8;
    if (row >= rows) continue;
    if (row < S) as_rope_pair(st.ka[u], st.kb[u], cs + (size_t)st.kps[u] * 64 + c, sn + (size_t)st.kps[u] * 64 + c, 1.f);
    *reinterpret_cast<short8_t*>(kimg + row * AS_RS + c) = st.ka[u];
    *reinterpret_cast<short8_t*>(kimg + row * AS_RS + c + 64) = st.kb[u];
  }
#pragma unroll
  for (int u = 0; u < AS_VIT; ++u) {
    const int it = threadIdx.x + u * AS_NW * 64, row = it >> 4, c = (it & 15) * 8;
    if (row >= rows) continue;
    *reinterpret_cast<short8_t*>(vimg + row * AS_RS + c) = st.v[u];
  }
}

// this lane's row of a token-major operand as B-operand fragments (columns kk*32 + lg*8 ..), optionally rotated
template <bool ROPE>
__device__ __forceinline__ void as_row_frags(short8_t (&f)[4], const bf16_t* base, long ld, int row, int S, int lg,
                                             const int* pos, const float* cs, const float* sn) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    f[kk] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
    if (row < S) f[kk] = *reinterpret_cast<const short8_t*>(base + (long)row * ld + kk * 32 + lg * 8);
  }
  if (ROPE && row < S) {
    const int ps = pos[row];
    as_rope_pair(f[0], f[2], cs + (size_t)ps * 64 + lg * 8, sn + (size_t)ps * 64 + lg * 8, 1.f);
    as_rope_pair(f[1], f[3], cs + (size_t)ps * 64 + 32 + lg * 8, sn + (size_t)ps * 64 + 32 + lg * 8, 1.f);
  }
}

// Fragment ownership: wave w owns fragments {nf-1-w, nf-16+w} (those that exist): under the causal mask fragment f
// costs f+1 units, waves w and w+4 share a SIMD, and this pairing keeps the four SIMDs within ~15 % of each other.
__device__ __forceinline__ int as_own(int nf, int wave, int which) {
  const int f = which == 0 ? nf - 1 - wave : nf - 16 + wave;
  return (f >= 0 && f < nf) ? f : -1;
}

// key fragments, mirrored: key fragment f costs nf - f units (the queries at or after it): wave w owns {w, 15 - w}
__device__ __forceinline__ int as_own_key(int nf, int wave, int which) {
  const int f = which == 0 ? wave : 15 - wave;
  return f < nf ? f : -1;
}

#define AS_NEG_INF (-__builtin_inff())

// ------------------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(AS_NW * 64) void attn_seq_fwd_kernel(AttnSeqParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Vs = Ks + AS_MAXF * 16 * AS_RS;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int S = p.S, nf = (S + 15) >> 4, W = p.H * AS_D;
  const bf16_t* qb = p.qkv + (long)b * S * p.ld + h * AS_D;
  const bf16_t* kb = qb + W;
  const bf16_t* vb = qb + 2 * W;
  const int* pos = p.pos + (long)b * S;
  int kv_valid = p.kv_len ? p.kv_len[b] : S;
  kv_valid = kv_valid < S ? kv_valid : S;

  const int rows = ((nf + 1) & ~1) * 16;
  // this wave's query rows: their positions and raw values are requested right behind the staging's K / V rows, before the
  // staging is committed (its rotary tables are a second, dependent round of loads); rotated afterwards -- as_row_frags<true>'s
  // expression.  The loops below are ROLLED on purpose: the first version unrolled fragments x key chunks into 37 KB (forward) /
  // 90 KB (backward) of straight-line code that every workgroup executes exactly once -- instruction-fetch bound (55-60 us per
  // workgroup even on an idle chip).
  short8_t qa[4], qn[4];
  {
    AsStage st;
    as_stage_issue(st, kb, vb, p.ld, S, rows, pos);
    const int f0 = as_own(nf, wave, 0), f1 = as_own(nf, wave, 1);
    const int r0 = f0 >= 0 ? 16 * f0 + lr : S, r1 = f1 >= 0 ? 16 * f1 + lr : S;
    int ps0 = 0, ps1 = 0;
    if (r0 < S) ps0 = pos[r0];
    if (r1 < S) ps1 = pos[r1];
    as_row_frags<false>(qa, qb, p.ld, r0, S, lg, pos, nullptr, nullptr);
    as_row_frags<false>(qn, qb, p.ld, r1, S, lg, pos, nullptr, nullptr);
    as_stage_commit(st, Ks, Vs, S, rows, p.cos_tab, p.sin_tab);
    if (r0 < S) {
      as_rope_pair(qa[0], qa[2], p.cos_tab + (size_t)ps0 * 64 + lg * 8, p.sin_tab + (size_t)ps0 * 64 + lg * 8, 1.f);
      as_rope_pair(qa[1], qa[3], p.cos_tab + (size_t)ps0 * 64 + 32 + lg * 8, p.sin_tab + (size_t)ps0 * 64 + 32 + lg * 8, 1.f);
    }
    if (r1 < S) {
      as_rope_pair(qn[0], qn[2], p.cos_tab + (size_t)ps1 * 64 + lg * 8, p.sin_tab + (size_t)ps1 * 64 + lg * 8, 1.f);
      as_rope_pair(qn[1], qn[3], p.cos_tab + (size_t)ps1 * 64 + 32 + lg * 8, p.sin_tab + (size_t)ps1 * 64 + 32 + lg * 8, 1.f);
    }
  }
  __syncthreads();

#pragma unroll 1
  for (int which = 0; which < 2; ++which) {
    const int f = as_own(nf, wave, which);
    if (f >= 0) {
      const int qi = 16 * f + lr;
      float4_t acc[8];
#pragma unroll
      for (int jd = 0; jd < 8; ++jd) acc[jd] = (float4_t){0.f, 0.f, 0.f, 0.f};
      float mrun = AS_NEG_INF, lsum = 0.f;          // per-lane partial row sum (its 8 keys per chunk); combined at the end
#pragma unroll 1
      for (int c = 0; 2 * c <= f; ++c) {            // 32 keys per step; causal: key fragments 0 .. f
        float4_t s[2];
        float tmax = AS_NEG_INF;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int j = 2 * c + u;
          s[u] = (float4_t){0.f, 0.f, 0.f, 0.f};
          if (j <= f) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
              s[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag_rm(Ks, j, kk, lr, lg), qa[kk], s[u], 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 16 * j + 4 * lg + r;
            const bool ok = j <= f && key < kv_valid && key <= qi && qi < S;
            s[u][r] = ok ? s[u][r] * p.scale : AS_NEG_INF;
            tmax = fmaxf(tmax, s[u][r]);
          }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(mrun, tmax);
        const float alpha = (m_new == AS_NEG_INF) ? 1.f : __expf(mrun - m_new);
        float psum = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = (m_new == AS_NEG_INF) ? 0.f : __expf(s[u][r] - m_new);
            s[u][r] = e;
            psum += e;
          }
        lsum = lsum * alpha + psum;
        mrun = m_new;
        const short8_t pb = as_pack8(s[0], s[1]);
#pragma unroll
        for (int jd = 0; jd < 8; ++jd) {
          acc[jd] *= alpha;
          acc[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag_tr(Vs, jd, c, lr, lg), pb, acc[jd], 0, 0, 0);
        }
      }
      lsum += __shfl_xor(lsum, 16, 64);
      lsum += __shfl_xor(lsum, 32, 64);
      if (qi < S) {
        const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
        bf16_t* ob = p.o + ((long)b * S + qi) * p.ldo + h * AS_D;
#pragma unroll
        for (int jd = 0; jd < 8; ++jd) {
          uint2 pk;
          pk.x = pack_bf2(acc[jd][0] * inv, acc[jd][1] * inv);
          pk.y = pack_bf2(acc[jd][2] * inv, acc[jd][3] * inv);
          *reinterpret_cast<uint2*>(ob + jd * 16 + lg * 4) = pk;
        }
        if (p.lse && lg == 0) p.lse[((long)b * p.H + h) * S + qi] = (lsum > 0.f) ? mrun + __logf(lsum) : AS_NEG_INF;
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qa[kk] = qn[kk];        // second fragment's rows move into the working set
  }
}

// ------------------------------------------------------------------------------------------- backward
// Phase A (lane owns a query): S^T = K.Q^T, dP^T = V.dO^T, dS^T = P o (dP^T - delta) * scale, dQ^T = K^T.dS^T
// (K^T by transpose reads of the key-major K image), dq un-rotated and stored.  Each wave keeps its query rows' Q
// (rotated) and dO fragments.
// Swap: every wave reads ITS key fragments' K / V rows out of the images into registers, barrier, every wave writes
// its Q / dO rows into the same LDS (now query-major images), barrier.
// Phase B (lane owns a key): S = Q.K^T, dP = dO.V^T, dV^T += dO^T.P, dK^T += Q^T.dS (transpose reads of the Q / dO
// images), dk un-rotated, dk / dv stored.
// dO arrives as the fp32 partial slabs of the split-K o_proj dgrad (summed here in slab order and rounded once:
// the same bits as the separate reduce launch this replaces) or as a plain bf16 matrix.
__device__ __forceinline__ void as_dout_frags(short8_t (&f)[4], const AttnSeqParams& p, long tok, int col0, bool live, int lg) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) f[kk] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
  if (!live) return;
  const int col = col0 + lg * 8;
  // Every load of a round is issued before the first add: a loop over the slabs with a load and an add per turn is one memory
  // latency per slab and per 32-column group (12 rounds for the step's 3 slabs, 32 for the batch-1 step's 8 -- it was a third
  // to a half of this kernel, tools/attn_seq_phases.py).  The sums keep the slab order: v = slab 0, v += slab 1, ...
  if (p.dout_bf && p.nslab <= 1) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f[kk] = *reinterpret_cast<const short8_t*>(p.dout_bf + tok * p.ldd + col + kk * 32);
  } else if (p.dout_bf) {                            // bf16 partial slabs: summed in fp32 in slab order, rounded once
    const bf16_t* src = p.dout_bf + tok * p.ldd + col;
    float v[4][8];
    for (int k0 = 0; k0 < p.nslab; k0 += 4) {        // 4 slabs x 4 column groups = 16 loads of 16 bytes in flight
      short8_t h[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          if (k0 + u < p.nslab) h[u][kk] = *reinterpret_cast<const short8_t*>(src + (long)(k0 + u) * p.slab + kk * 32);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (k0 + u >= p.nslab) continue;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = bf2f((bf16_t)h[u][kk][e]);
            if (k0 + u == 0) v[kk][e] = x; else v[kk][e] += x;
          }
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      f[kk] = as_pack8((float4_t){v[kk][0], v[kk][1], v[kk][2], v[kk][3]}, (float4_t){v[kk][4], v[kk][5], v[kk][6], v[kk][7]});
  } else {
    const float* src = p.dout + tok * p.ldd + col;
    float4_t a[4], c[4];
    for (int k0 = 0; k0 < p.nslab; k0 += 2) {        // 2 slabs x 4 column groups x 32 bytes in flight
      float4_t a2[2][4], c2[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          if (k0 + u < p.nslab) {
            a2[u][kk] = *reinterpret_cast<const float4_t*>(src + (long)(k0 + u) * p.slab + kk * 32);
            c2[u][kk] = *reinterpret_cast<const float4_t*>(src + (long)(k0 + u) * p.slab + kk * 32 + 4);
          }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (k0 + u >= p.nslab) continue;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          if (k0 + u == 0) { a[kk] = a2[u][kk]; c[kk] = c2[u][kk]; }
          else {
            a[kk][0] += a2[u][kk][0]; a[kk][1] += a2[u][kk][1]; a[kk][2] += a2[u][kk][2]; a[kk][3] += a2[u][kk][3];
            c[kk][0] += c2[u][kk][0]; c[kk][1] += c2[u][kk][1]; c[kk][2] += c2[u][kk][2]; c[kk][3] += c2[u][kk][3];
          }
        }
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f[kk] = as_pack8(a[kk], c[kk]);
  }
}

// un-rotate an fp32 gradient held as acc[jd][r] = g[d = 16*jd + 4*lg + r] (partner d + 64 is acc[jd + 4]) and pack to bf16
__device__ __forceinline__ void as_unrope_store(float4_t (&acc)[8], bf16_t* dst, bool rope, const float* cs, const float* sn,
                                                int lg) {
#pragma unroll
  for (int jd = 0; jd < 4; ++jd) {
    float4_t lo = acc[jd], hi = acc[jd + 4];
    if (rope) {
      const float4_t c = *reinterpret_cast<const float4_t*>(cs + jd * 16 + lg * 4);
      const float4_t s = *reinterpret_cast<const float4_t*>(sn + jd * 16 + lg * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {          // inverse rotation = rotation by -theta (sign -1 in as_rope_pair's formula)
        const float x1 = acc[jd][r], x2 = acc[jd + 4][r];
        lo[r] = x1 * c[r] + x2 * s[r];
        hi[r] = x2 * c[r] - x1 * s[r];
      }
    }
    uint2 pk;
    pk.x = pack_bf2(lo[0], lo[1]);
    pk.y = pack_bf2(lo[2], lo[3]);
    *reinterpret_cast<uint2*>(dst + jd * 16 + lg * 4) = pk;
    pk.x = pack_bf2(hi[0], hi[1]);
    pk.y = pack_bf2(hi[2], hi[3]);
    *reinterpret_cast<uint2*>(dst + 64 + jd * 16 + lg * 4) = pk;
  }
}

__global__ __launch_bounds__(AS_NW * 64) void attn_seq_bwd_kernel(AttnSeqParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* I0 = reinterpret_cast<bf16_t*>(smem);                 // K, then Q
  bf16_t* I1 = I0 + AS_MAXF * 16 * AS_RS;                       // V, then dO
  float* lse_s = reinterpret_cast<float*>(I1 + AS_MAXF * 16 * AS_RS);
  float* dlt_s = lse_s + AS_MAXF * 16;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int S = p.S, nf = (S + 15) >> 4, W = p.H * AS_D;
  const bf16_t* qb = p.qkv + (long)b * S * p.ld + h * AS_D;
  const bf16_t* kb = qb + W;
  const bf16_t* vb = qb + 2 * W;
  bf16_t* dqb = p.dqkv + (long)b * S * p.ld + h * AS_D;
  const int* pos = p.pos + (long)b * S;
  int kv_valid = p.kv_len ? p.kv_len[b] : S;
  kv_valid = kv_valid < S ? kv_valid : S;

  const int rows = ((nf + 1) & ~1) * 16;
  AS_STAMP(0)
  // Load order: everything this wave needs from global memory is requested before the first use -- the K / V rows of the
  // staging, then this wave's own query rows (positions, Q, O), and only then the staging is committed (its rotary tables are a
  // second, dependent round of loads) and the dO slabs are summed.  As separate load -> use -> load steps the prologue was 6-8
  // memory latencies per owned fragment, two fragments back to back on the waves that own two (tools/attn_seq_phases.py).
  // Two register sets (first / second owned fragment); the loops over fragments and key chunks below are rolled (see the forward
  // kernel) and work on set 0, the sets are exchanged at the end of each fragment iteration.
  short8_t qf0[4], gf0[4], qf1[4], gf1[4], of0[4], of1[4];
  int ps0 = 0, ps1 = 0;
  float dlt0 = 0.f, dlt1 = 0.f;
  AsStage st;
  as_stage_issue(st, kb, vb, p.ld, S, rows, pos);
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    short8_t (&qf)[4] = which ? qf1 : qf0;
    short8_t (&of)[4] = which ? of1 : of0;
    const int f = as_own(nf, wave, which);
    const int qi = 16 * (f < 0 ? 0 : f) + lr;
    const int row = f < 0 ? S : qi;                    // row >= S: as_row_frags returns zeros and loads nothing
    if (row < S) { if (which) ps1 = pos[row]; else ps0 = pos[row]; }
    as_row_frags<false>(qf, qb, p.ld, row, S, lg, pos, nullptr, nullptr);
    as_row_frags<false>(of, p.o_in + (long)b * S * p.ldo + h * AS_D, p.ldo, row, S, lg, pos, nullptr, nullptr);
  }
  as_stage_commit(st, I0, I1, S, rows, p.cos_tab, p.sin_tab);
  AS_STAMP(1)
  for (int i = threadIdx.x; i < AS_MAXF * 16; i += AS_NW * 64) lse_s[i] = i < S ? p.lse[((long)b * p.H + h) * S + i] : 1e30f;

#pragma unroll
  for (int which = 0; which < 2; ++which) {
    short8_t (&qf)[4] = which ? qf1 : qf0;
    short8_t (&gf)[4] = which ? gf1 : gf0;
    short8_t (&of)[4] = which ? of1 : of0;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) gf[kk] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
    const int f = as_own(nf, wave, which);
    if (f < 0) continue;
    const int qi = 16 * f + lr;
    as_dout_frags(gf, p, (long)b * S + qi, h * AS_D, qi < S, lg);
    if (qi < S) {                                      // rotary on the Q rows: as_row_frags<true>'s expression
      const int ps = which ? ps1 : ps0;
      as_rope_pair(qf[0], qf[2], p.cos_tab + (size_t)ps * 64 + lg * 8, p.sin_tab + (size_t)ps * 64 + lg * 8, 1.f);
      as_rope_pair(qf[1], qf[3], p.cos_tab + (size_t)ps * 64 + 32 + lg * 8, p.sin_tab + (size_t)ps * 64 + 32 + lg * 8, 1.f);
    }
    // delta = sum_d dO * O over this lane's columns, combined across the row's four lanes
    float dlt = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int e = 0; e < 8; ++e) dlt += bf2f((bf16_t)of[kk][e]) * bf2f((bf16_t)gf[kk][e]);
    dlt += __shfl_xor(dlt, 16, 64);
    dlt += __shfl_xor(dlt, 32, 64);
    if (lg == 0) dlt_s[qi] = dlt;
    if (which) dlt1 = dlt; else dlt0 = dlt;
  }
  AS_STAMP(2)
  __syncthreads();
  AS_STAMP(3)

  // Small batches (B * H <= 128 workgroups would leave half the chip idle): the launch is two workgroups per (batch, head),
  // blockIdx.y = 0 computes dq (phase A), 1 computes dk | dv (phase B); both stage K / V and build delta for every row.
  const bool run_a = gridDim.y == 1 || blockIdx.y == 0, run_b = gridDim.y == 1 || blockIdx.y == 1;
  // ---------------- phase A
#pragma unroll 1
  for (int which = 0; which < 2; ++which) {
    const int f = as_own(nf, wave, which);
    if (f >= 0 && run_a) {
      const int qi = 16 * f + lr;
      const float dlt = dlt0, lse = lse_s[qi];
      float4_t acc[8];
#pragma unroll
      for (int jd = 0; jd < 8; ++jd) acc[jd] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int c = 0; 2 * c <= f; ++c) {
        float4_t ds[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int j = 2 * c + u;
          ds[u] = (float4_t){0.f, 0.f, 0.f, 0.f};
          if (j > f) continue;
          float4_t sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag_rm(I0, j, kk, lr, lg), qf0[kk], sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag_rm(I1, j, kk, lr, lg), gf0[kk], dp, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 16 * j + 4 * lg + r;
            const bool ok = key < kv_valid && key <= qi && qi < S;
            const float pr = ok ? __expf(sc[r] * p.scale - lse) : 0.f;
            ds[u][r] = ok ? pr * (dp[r] - dlt) * p.scale : 0.f;
          }
        }
        const short8_t db = as_pack8(ds[0], ds[1]);
#pragma unroll
        for (int jd = 0; jd < 8; ++jd)
          acc[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag_tr(I0, jd, c, lr, lg), db, acc[jd], 0, 0, 0);
      }
      if (qi < S) {
        const int ps = pos[qi];
        as_unrope_store(acc, dqb + (long)qi * p.ld, true, p.cos_tab + (size_t)ps * 64, p.sin_tab + (size_t)ps * 64, lg);
      }
    }
    // exchange the register sets: the next iteration (and the image swap below) address them by position
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const short8_t tq = qf0[kk], tg = gf0[kk];
      qf0[kk] = qf1[kk]; gf0[kk] = gf1[kk];
      qf1[kk] = tq; gf1[kk] = tg;
    }
    const float td = dlt0; dlt0 = dlt1; dlt1 = td;
  }
  AS_STAMP(4)
  if (!run_b) return;                                  // uniform over the workgroup: no barrier is skipped by part of it
  // ---------------- swap the images: keys' rows -> registers, queries' rows -> LDS
  short8_t kf0[4], vf0[4], kf1[4], vf1[4];
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    short8_t (&kf)[4] = which ? kf1 : kf0;
    short8_t (&vf)[4] = which ? vf1 : vf0;
    const int f = as_own_key(nf, wave, which);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      kf[kk] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
      vf[kk] = kf[kk];
      if (f >= 0) {
        kf[kk] = as_frag_rm(I0, f, kk, lr, lg);
        vf[kk] = as_frag_rm(I1, f, kk, lr, lg);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int which = 0; which < 2; ++which) {          // after two exchanges the sets are back in fragment order
    short8_t (&qf)[4] = which ? qf1 : qf0;
    short8_t (&gf)[4] = which ? gf1 : gf0;
    const int f = as_own(nf, wave, which);
    if (f < 0) continue;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      *reinterpret_cast<short8_t*>(I0 + (16 * f + lr) * AS_RS + kk * 32 + lg * 8) = qf[kk];
      *reinterpret_cast<short8_t*>(I1 + (16 * f + lr) * AS_RS + kk * 32 + lg * 8) = gf[kk];
    }
  }
  __syncthreads();
  AS_STAMP(5)
  // ---------------- phase B
#pragma unroll 1
  for (int which = 0; which < 2; ++which) {
    const int f = as_own_key(nf, wave, which);
    if (f >= 0) {
      const int ki = 16 * f + lr;
      const bool key_ok = ki < kv_valid;
      float4_t adk[8], adv[8];
#pragma unroll
      for (int jd = 0; jd < 8; ++jd) {
        adk[jd] = (float4_t){0.f, 0.f, 0.f, 0.f};
        adv[jd] = (float4_t){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll 1
      for (int c = f >> 1; 2 * c < nf; ++c) {          // query fragments 2c, 2c+1; causal: queries at or after the keys
        float4_t pr[2], ds[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int j = 2 * c + u;
          pr[u] = (float4_t){0.f, 0.f, 0.f, 0.f};
          ds[u] = (float4_t){0.f, 0.f, 0.f, 0.f};
          if (j < f || j >= nf) continue;
          float4_t sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag_rm(I0, j, kk, lr, lg), kf0[kk], sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag_rm(I1, j, kk, lr, lg), vf0[kk], dp, 0, 0, 0);
          }
          const float4_t l4 = *reinterpret_cast<const float4_t*>(lse_s + 16 * j + 4 * lg);
          const float4_t d4 = *reinterpret_cast<const float4_t*>(dlt_s + 16 * j + 4 * lg);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qq = 16 * j + 4 * lg + r;
            const bool ok = key_ok && qq < S && ki <= qq;
            const float e = ok ? __expf(sc[r] * p.scale - l4[r]) : 0.f;
            pr[u][r] = e;
            ds[u][r] = ok ? e * (dp[r] - d4[r]) * p.scale : 0.f;
          }
        }
        const short8_t pb = as_pack8(pr[0], pr[1]), db = as_pack8(ds[0], ds[1]);
#pragma unroll
        for (int jd = 0; jd < 8; ++jd) {
          adv[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag_tr(I1, jd, c, lr, lg), pb, adv[jd], 0, 0, 0);
          adk[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag_tr(I0, jd, c, lr, lg), db, adk[jd], 0, 0, 0);
        }
      }
      if (ki < S) {
        const int ps = pos[ki];
        as_unrope_store(adk, dqb + W + (long)ki * p.ld, true, p.cos_tab + (size_t)ps * 64, p.sin_tab + (size_t)ps * 64, lg);
        as_unrope_store(adv, dqb + 2 * W + (long)ki * p.ld, false, nullptr, nullptr, lg);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { kf0[kk] = kf1[kk]; vf0[kk] = vf1[kk]; }
  }
  AS_STAMP(6)
}

// ------------------------------------------------------------------------------------------- C ABI
static size_t as_lds_bytes(bool bwd) {
  return (size_t)2 * AS_MAXF * 16 * AS_RS * 2 + (bwd ? 2 * AS_MAXF * 16 * 4 : 0);
}

static int as_check(int B, int H, int S, int D, int ld, int ldo) {
  if (B <= 0 || H <= 0 || S <= 0) return MH_ERR_ARG;
  if (D != AS_D || S > AS_MAXF * 16) return MH_ERR_UNSUPPORTED;
  if (ld % 8 || ldo % 8 || ld < 3 * H * D || ldo < H * D) return MH_ERR_ARG;
  return MH_OK;
}

// o[b, s, h*128..] = softmax(causal(rope(q) rope(k)^T * scale)) v for qkv = [q | k | v] [B, S, ld] bf16 (pre-rotary)
extern "C" int mh_attn_rope_fwd(const void* qkv, int ld, void* o, int ldo, float* lse, const int* pos, const float* cos_tab,
                                const float* sin_tab, const int* kv_len, int B, int H, int S, int D, float scale,
                                hipStream_t stream) {
  int rc = as_check(B, H, S, D, ld, ldo);
  if (rc) return rc;
  if (!qkv || !o || !pos || !cos_tab || !sin_tab) return MH_ERR_ARG;
  AttnSeqParams p = {};
  p.qkv = (const bf16_t*)qkv; p.o = (bf16_t*)o; p.lse = lse; p.pos = pos; p.cos_tab = cos_tab; p.sin_tab = sin_tab;
  p.kv_len = kv_len; p.B = B; p.H = H; p.S = S; p.ld = ld; p.ldo = ldo; p.scale = scale;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_seq_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)as_lds_bytes(false));
    attr = true;
  }
  hipLaunchKernelGGL(attn_seq_fwd_kernel, dim3(B * H), dim3(AS_NW * 64), as_lds_bytes(false), stream, p);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

int mh_launch_attn_rope_bwd(const void* qkv, int ld, const void* o, int ldo, const void* dout, int dout_is_bf16, int nslab,
                            long slab, int ldd, const float* lse, void* dqkv, const int* pos, const float* cos_tab,
                            const float* sin_tab, const int* kv_len, int B, int H, int S, int D, float scale,
                            hipStream_t stream);

// dqkv = [dq | dk | dv] for the same layout, dout bf16 [B*S, ldd].  The internal launcher also takes dout as fp32
// [nslab][B*S, ldd] partial sums (slab = elements between slabs): mh_gemm_attn_rope_bwd in gemm.hip.
extern "C" int mh_attn_rope_bwd(const void* qkv, int ld, const void* o, int ldo, const void* dout, int ldd, const float* lse,
                                void* dqkv, const int* pos, const float* cos_tab, const float* sin_tab, const int* kv_len,
                                int B, int H, int S, int D, float scale, hipStream_t stream) {
  return mh_launch_attn_rope_bwd(qkv, ld, o, ldo, dout, 1, 1, 0, ldd, lse, dqkv, pos, cos_tab, sin_tab, kv_len, B, H, S, D, scale,
                                 stream);
}

extern "C" int mh_attn_rope_supported(int S, int D) { return (D == AS_D && S > 0 && S <= AS_MAXF * 16) ? 1 : 0; }

static long long* g_as_trace = nullptr;
#ifdef MH_DEBUG_HOOKS
extern "C" void mhdbg_set_attn_seq_trace(void* ptr) { g_as_trace = (long long*)ptr; }   // phase stamps (libmyriad_hip_dbg.so only)
#endif

int mh_launch_attn_rope_bwd(const void* qkv, int ld, const void* o, int ldo, const void* dout, int dout_is_bf16, int nslab,
                            long slab, int ldd, const float* lse, void* dqkv, const int* pos, const float* cos_tab,
                            const float* sin_tab, const int* kv_len, int B, int H, int S, int D, float scale,
                            hipStream_t stream) {
  int rc = as_check(B, H, S, D, ld, ldo);
  if (rc) return rc;
  if (!qkv || !o || !dout || !lse || !dqkv || !pos || !cos_tab || !sin_tab || nslab < 1 || (ldd % 8) || ldd < H * D)
    return MH_ERR_ARG;
  AttnSeqParams p = {};
  p.qkv = (const bf16_t*)qkv; p.o_in = (const bf16_t*)o; p.lse = const_cast<float*>(lse); p.pos = pos;
  p.cos_tab = cos_tab; p.sin_tab = sin_tab; p.kv_len = kv_len; p.B = B; p.H = H; p.S = S; p.ld = ld; p.ldo = ldo;
  p.scale = scale; p.dqkv = (bf16_t*)dqkv; p.nslab = nslab; p.slab = slab; p.ldd = ldd;
  if (dout_is_bf16) p.dout_bf = (const bf16_t*)dout;
  else p.dout = (const float*)dout;
  p.trace = g_as_trace;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_seq_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)as_lds_bytes(true));
    attr = true;
  }
  // option attn_bwd_split = 0 keeps one workgroup per (batch, head) (A/B runs)
  const int parts = (mh_opt(MH_OPT_ATTN_BWD_SPLIT) && B * H <= 128) ? 2 : 1;
  hipLaunchKernelGGL(attn_seq_bwd_kernel, dim3(B * H, parts), dim3(AS_NW * 64), as_lds_bytes(true), stream, p);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

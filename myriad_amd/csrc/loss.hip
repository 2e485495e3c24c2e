// K11: clamp-CE loss (reference modeling_llama.py:718-728): softmax -> clamp(p, 1e-7, 1-1e-7) -> log -> NLL,
// mean over labels != -100, with the analytically fused backward:
//     dL/dx_j = g * (p_j - [j==t])   if 1e-7 <= p_t <= 1-1e-7   else 0      (clamp kills the whole row's grad)
// fp32 logits in, bf16 d(logits) out (A operand of the lm_head dgrad GEMM).  One 256-thread workgroup per row;
// HBM-bound: 4*V bytes read twice + 2*V written per row.
// Also: deterministic fp32 sum (loss reduction) and row arg-max (greedy decode, K5 decode path).
#include "common.h"

#define LNT 256
#define LNW 4

__global__ __launch_bounds__(LNT) void clamp_ce_kernel(const float* __restrict__ logits, long ldl,
                                                       const long* __restrict__ labels, float* __restrict__ row_loss,
                                                       bf16_t* __restrict__ dlogits, long ldd, int V, float gscale) {
  __shared__ float red[LNW];
  const long row = blockIdx.x;
  const float* x = logits + row * ldl;
  const long t = labels[row];
  float mx = -__builtin_inff();
  for (int j = threadIdx.x; j < V; j += LNT) mx = fmaxf(mx, x[j]);
  mx = block_max<LNW>(mx, red);
  float se = 0.f;
  for (int j = threadIdx.x; j < V; j += LNT) se += expf(x[j] - mx);
  se = block_sum<LNW>(se, red);
  const float inv = 1.f / se;
  float gate = 0.f;
  if (t >= 0 && t < V) {
    const float pt = expf(x[t] - mx) * inv;
    const float pc = fminf(fmaxf(pt, 1e-7f), 1.f - 1e-7f);
    if (threadIdx.x == 0) row_loss[row] = -logf(pc);
    gate = (pt >= 1e-7f && pt <= 1.f - 1e-7f) ? gscale : 0.f;
  } else if (threadIdx.x == 0) {
    row_loss[row] = 0.f;
  }
  if (dlogits) {
    bf16_t* d = dlogits + row * ldd;
    for (int j = threadIdx.x; j < ldd; j += LNT) {
      float g = 0.f;
      if (j < V && gate != 0.f) g = gate * (expf(x[j] - mx) * inv - (j == t ? 1.f : 0.f));
      d[j] = f2bf(g);
    }
  }
}

// The same row held in registers by a 1024-thread workgroup (V <= 32768, 16-byte aligned rows): the logits are read ONCE with
// 16-byte loads instead of three times with 4-byte ones (the step's 128 label rows of V = 32000: 79 -> ~20 us).  Same
// expressions; the sum of exponentials is accumulated in another order than the 256-thread kernel's (both deterministic).
#define LNT_W 1024
__global__ __launch_bounds__(LNT_W) void clamp_ce_wide_kernel(const float* __restrict__ logits, long ldl, const long* __restrict__ labels,
                                                              float* __restrict__ row_loss, bf16_t* __restrict__ dlogits, long ldd,
                                                              int V, float gscale) {
  __shared__ float red[LNT_W / 64];
  const long row = blockIdx.x;
  const float* x = logits + row * ldl;
  const long t = labels[row];
  float4_t xv[8];
  float mx = -__builtin_inff();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = (i * LNT_W + threadIdx.x) * 4;
    xv[i] = (float4_t){-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    if (j + 3 < V) xv[i] = *reinterpret_cast<const float4_t*>(x + j);
    else
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (j + e < V) xv[i][e] = x[j + e];
    mx = fmaxf(fmaxf(mx, fmaxf(xv[i][0], xv[i][1])), fmaxf(xv[i][2], xv[i][3]));
  }
  mx = block_max<LNT_W / 64>(mx, red);
  float se = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xv[i][e] = expf(xv[i][e] - mx);                  // exp(-inf) = 0 for the columns past V
      se += xv[i][e];
    }
  se = block_sum<LNT_W / 64>(se, red);
  const float inv = 1.f / se;
  float gate = 0.f;
  if (t >= 0 && t < V) {
    const float pt = expf(x[t] - mx) * inv;
    const float pc = fminf(fmaxf(pt, 1e-7f), 1.f - 1e-7f);
    if (threadIdx.x == 0) row_loss[row] = -logf(pc);
    gate = (pt >= 1e-7f && pt <= 1.f - 1e-7f) ? gscale : 0.f;
  } else if (threadIdx.x == 0) {
    row_loss[row] = 0.f;
  }
  if (dlogits) {
    bf16_t* d = dlogits + row * ldd;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = (i * LNT_W + threadIdx.x) * 4;
      if (j >= ldd) continue;
      float g[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        g[e] = (j + e < V && gate != 0.f) ? gate * (xv[i][e] * inv - (j + e == t ? 1.f : 0.f)) : 0.f;
      if (j + 3 < ldd) {
        uint2 pk;
        pk.x = pack_bf2(g[0], g[1]);
        pk.y = pack_bf2(g[2], g[3]);
        *reinterpret_cast<uint2*>(d + j) = pk;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (j + e < ldd) d[j + e] = f2bf(g[e]);
      }
    }
  }
}

extern "C" int mh_clamp_ce(const float* logits, long ldl, const long* labels, float* row_loss, void* dlogits_bf16,
                           long ldd, int R, int V, float grad_scale, hipStream_t stream) {
  if (R <= 0) return MH_OK;
  if (dlogits_bf16 && ldd < V) return MH_ERR_ARG;
  if (V <= 8 * LNT_W * 4 && (!dlogits_bf16 || ldd <= 8 * LNT_W * 4) && (ldl % 4) == 0 && (ldd % 4) == 0 &&
      !((uintptr_t)logits & 15) && !((uintptr_t)dlogits_bf16 & 7)) {
    hipLaunchKernelGGL(clamp_ce_wide_kernel, dim3(R), dim3(LNT_W), 0, stream, logits, ldl, labels, row_loss, (bf16_t*)dlogits_bf16,
                       ldd, V, grad_scale);
    MH_CHECK_LAUNCH();
    return MH_OK;
  }
  hipLaunchKernelGGL(clamp_ce_kernel, dim3(R), dim3(LNT), 0, stream, logits, ldl, labels, row_loss,
                     (bf16_t*)dlogits_bf16, ldd, V, grad_scale);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// out[0] = scale * sum_i x[i]   (single workgroup, fixed summation order -> deterministic)
__global__ __launch_bounds__(LNT) void sum_kernel(const float* __restrict__ x, float* __restrict__ out, long n, float scale) {
  __shared__ float red[LNW];
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += LNT) s += x[i];
  s = block_sum<LNW>(s, red);
  if (threadIdx.x == 0) out[0] = s * scale;
}
extern "C" int mh_sum_f32(const float* x, float* out, long n, float scale, hipStream_t stream) {
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(LNT), 0, stream, x, out, n, scale);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// row arg-max (first index on ties, like torch.argmax) with optional banned id (min_length / EOS ban) and
// optional top1-top2 margin output.
__global__ __launch_bounds__(LNT) void argmax_kernel(const float* __restrict__ logits, long ldl, long* __restrict__ out,
                                                     float* __restrict__ margin, int V, int ban_id) {
  __shared__ float sv[LNT];
  __shared__ int si[LNT];
  __shared__ float s2[LNT];
  const long row = blockIdx.x;
  const float* x = logits + row * ldl;
  float best = -__builtin_inff(), second = -__builtin_inff();
  int bi = threadIdx.x < V ? threadIdx.x : 0;   // stays in range even if every logit is NaN
  for (int j = threadIdx.x; j < V; j += LNT) {
    const float v = (j == ban_id) ? -__builtin_inff() : x[j];
    if (v > best) { second = best; best = v; bi = j; }
    else if (v > second) second = v;
  }
  sv[threadIdx.x] = best; si[threadIdx.x] = bi; s2[threadIdx.x] = second;
  __syncthreads();
  for (int o = LNT / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const float a = sv[threadIdx.x], b = sv[threadIdx.x + o];
      const int ia = si[threadIdx.x], ib = si[threadIdx.x + o];
      const float a2 = s2[threadIdx.x], b2 = s2[threadIdx.x + o];
      const bool take_b = (b > a) || (b == a && ib < ia);
      sv[threadIdx.x] = take_b ? b : a;
      si[threadIdx.x] = take_b ? ib : ia;
      s2[threadIdx.x] = fmaxf(fmaxf(a2, b2), take_b ? a : b);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[row] = si[0];
    if (margin) margin[row] = sv[0] - s2[0];
  }
}
extern "C" int mh_argmax_rows(const float* logits, long ldl, long* out, float* margin, int R, int V, int ban_id,
                              hipStream_t stream) {
  if (R <= 0) return MH_OK;
  hipLaunchKernelGGL(argmax_kernel, dim3(R), dim3(LNT), 0, stream, logits, ldl, out, margin, V, ban_id);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
// p_max of each row after the arg-max is known: 1 / sum_j exp((x_j - best) * inv_temp), banned id excluded.  The decode loop
// uses it to decide whether the reference's `do_sample=True, top_p=p` call (evaluation_aqa_dataset.py:289-301) is the
// arg-max (HF's top-p warper keeps only the first token whenever p_max >= top_p) or a real draw.
__global__ __launch_bounds__(LNT) void pmax_kernel(const float* __restrict__ logits, long ldl, const long* __restrict__ best_id,
                                                   float* __restrict__ pmax, int V, int ban_id, float inv_temp) {
  __shared__ float red[LNW];
  const long row = blockIdx.x;
  const float* x = logits + row * ldl;
  const float best = x[best_id[row]];
  float s = 0.f;
  for (int j = threadIdx.x; j < V; j += LNT)
    if (j != ban_id) s += __expf((x[j] - best) * inv_temp);
  s = block_sum<LNW>(s, red);
  if (threadIdx.x == 0) pmax[row] = 1.f / s;
}
// End of a decode step on the device: the picked ids become the next step's input ids (the token step then needs nothing
// from the host and is a fixed hipGraph), the step's results are packed into ONE small record rec[3][R] f32 = (id, margin,
// p_max) -- ids < 2^24 are exact in fp32 -- so the host fetches a step with a single copy, and the step counter advances.
// (Writing the record straight into pinned host memory was measured: a kernel that stores to host memory ends with a
// system-scope release, +0.2 ms per token.)
__global__ void decode_record_kernel(const long* __restrict__ nxt, const float* __restrict__ margin, const float* __restrict__ pmax,
                                     float* __restrict__ rec, long* __restrict__ next_ids, int* __restrict__ step, int R) {
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    const long id = nxt[r];
    rec[r] = (float)id;
    rec[R + r] = margin[r];
    rec[2 * R + r] = pmax[r];
    next_ids[r] = id;
  }
  if (threadIdx.x == 0) *step += 1;
}
// The same plus the device-resident position / valid-length counters of the KV-cache decode (one int per row each) advanced by
// one: the three bookkeeping launches at the end of a token step as one.
__global__ void decode_advance_kernel(const long* __restrict__ nxt, const float* __restrict__ margin, const float* __restrict__ pmax,
                                      float* __restrict__ rec, long* __restrict__ next_ids, int* __restrict__ step,
                                      int* __restrict__ pos, int* __restrict__ kvlen, int R) {
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    const long id = nxt[r];
    rec[r] = (float)id;
    rec[R + r] = margin[r];
    rec[2 * R + r] = pmax[r];
    next_ids[r] = id;
    pos[r] += 1;
    kvlen[r] += 1;
  }
  if (threadIdx.x == 0) *step += 1;
}
extern "C" int mh_decode_advance(const long* nxt, const float* margin, const float* pmax, float* rec, long* next_ids, int* step,
                                 int* pos, int* kvlen, int R, hipStream_t stream) {
  if (R <= 0) return MH_OK;
  if (!nxt || !margin || !pmax || !rec || !next_ids || !step || !pos || !kvlen) return MH_ERR_ARG;
  hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(64), 0, stream, nxt, margin, pmax, rec, next_ids, step, pos, kvlen, R);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
extern "C" int mh_decode_record(const long* nxt, const float* margin, const float* pmax, float* rec, long* next_ids, int* step,
                                int R, hipStream_t stream) {
  if (R <= 0) return MH_OK;
  if (!nxt || !margin || !pmax || !rec || !next_ids || !step) return MH_ERR_ARG;
  hipLaunchKernelGGL(decode_record_kernel, dim3(1), dim3(64), 0, stream, nxt, margin, pmax, rec, next_ids, step, R);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// arg-max, top-1 / top-2 margin and p_max of a row in ONE launch with the row in registers (1024 threads, V <= 32768, 16-byte
// aligned rows): the decode step's lm-head tail was two single-workgroup scans of 128 KB with 4-byte loads (42 + 20 us per token).
// Same rules as argmax_kernel (first index on ties, the banned id counts as -inf, a duplicate of the maximum gives margin 0) and
// pmax_kernel (banned id excluded); the exponentials are summed in another order (both deterministic).
struct AmTop { float best; int idx; float second; };
__device__ __forceinline__ AmTop am_combine(const AmTop& a, const AmTop& b) {
  const bool take_b = (b.best > a.best) || (b.best == a.best && b.idx < a.idx);
  AmTop o;
  o.best = take_b ? b.best : a.best;
  o.idx = take_b ? b.idx : a.idx;
  o.second = fmaxf(fmaxf(a.second, b.second), take_b ? a.best : b.best);
  return o;
}
__global__ __launch_bounds__(LNT_W) void argmax_pmax_wide_kernel(const float* __restrict__ logits, long ldl, long* __restrict__ out,
                                                                 float* __restrict__ margin, float* __restrict__ pmax, int V, int ban_id,
                                                                 float inv_temp) {
  __shared__ float sb[LNT_W / 64], s2[LNT_W / 64], red[LNT_W / 64];
  __shared__ int si[LNT_W / 64];
  const long row = blockIdx.x;
  const float* x = logits + row * ldl;
  const float ninf = -__builtin_inff();
  float4_t xv[8];
  const int t4 = (int)threadIdx.x * 4;
  AmTop tp = {ninf, t4 < V ? t4 : 0, ninf};          // the index stays in range even if every logit is NaN
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = (i * LNT_W + threadIdx.x) * 4;
    xv[i] = (float4_t){ninf, ninf, ninf, ninf};
    if (j + 3 < V) xv[i] = *reinterpret_cast<const float4_t*>(x + j);
    else
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (j + e < V) xv[i][e] = x[j + e];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (j + e == ban_id) xv[i][e] = ninf;
      const float v = xv[i][e];
      if (v > tp.best) { tp.second = tp.best; tp.best = v; tp.idx = j + e; }
      else if (v > tp.second) tp.second = v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    AmTop q;
    q.best = __shfl_xor(tp.best, o, 64);
    q.idx = __shfl_xor(tp.idx, o, 64);
    q.second = __shfl_xor(tp.second, o, 64);
    tp = am_combine(tp, q);
  }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) { sb[w] = tp.best; si[w] = tp.idx; s2[w] = tp.second; }
  __syncthreads();
  AmTop all = {sb[0], si[0], s2[0]};
#pragma unroll
  for (int i = 1; i < LNT_W / 64; ++i) all = am_combine(all, (AmTop){sb[i], si[i], s2[i]});
  if (threadIdx.x == 0) {
    out[row] = all.idx;
    if (margin) margin[row] = all.best - all.second;
  }
  if (pmax) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += __expf((xv[i][e] - all.best) * inv_temp);      // -inf (past V, banned) adds exp(-inf) = 0
    s = block_sum<LNT_W / 64>(s, red);
    if (threadIdx.x == 0) pmax[row] = 1.f / s;
  }
}

extern "C" int mh_argmax_pmax_rows(const float* logits, long ldl, long* out, float* margin, float* pmax, int R, int V,
                                   int ban_id, float inv_temp, hipStream_t stream) {
  if (R <= 0) return MH_OK;
  if (!out || !pmax) return MH_ERR_ARG;
  if (V <= 8 * LNT_W * 4 && (ldl % 4) == 0 && !((uintptr_t)logits & 15) && inv_temp > 0.f) {
    hipLaunchKernelGGL(argmax_pmax_wide_kernel, dim3(R), dim3(LNT_W), 0, stream, logits, ldl, out, margin, pmax, V, ban_id, inv_temp);
    MH_CHECK_LAUNCH();
    return MH_OK;
  }
  hipLaunchKernelGGL(argmax_kernel, dim3(R), dim3(LNT), 0, stream, logits, ldl, out, margin, V, ban_id);
  hipLaunchKernelGGL(pmax_kernel, dim3(R), dim3(LNT), 0, stream, logits, ldl, out, pmax, V, ban_id, inv_temp);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

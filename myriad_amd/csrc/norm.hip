// K6/K7: fused LayerNorm / RMSNorm, fwd and dgrad-only bwd (all norm weights on the hot path are frozen).
// Residual streams are fp32 in HBM; the normalised output feeding an MFMA GEMM is written as bf16.
// One 256-thread workgroup per row, float4 loads (HBM-bound: algorithmic bytes = 4*D in + 2*D out per row).
//   RMSNorm  : reference modeling_llama.py:66-74   (fp32 variance, eps inside rsqrt)
//   LayerNorm: reference eva_vit.py:175-176 (eps 1e-6), blip2.py:119-125 (ln_vision fp32, eps 1e-5),
//              Qformer.py:106,288,374 (eps 1e-12)
#include "common.h"

#define NT 256
#define NW 4

// ---- RMSNorm ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void rmsnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         bf16_t* __restrict__ y, int D, long ldy, float eps) {
  __shared__ float red[NW];
  const size_t row = blockIdx.x;
  const float* xr = x + row * D;
  float ss = 0.f;
  for (int i = threadIdx.x * 4; i < D; i += NT * 4) {
    const float4_t v = *reinterpret_cast<const float4_t*>(xr + i);
    ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  ss = block_sum<NW>(ss, red);
  const float r = rsqrtf(ss / D + eps);
  for (int i = threadIdx.x * 4; i < D; i += NT * 4) {
    const float4_t v = *reinterpret_cast<const float4_t*>(xr + i);
    const float4_t g = *reinterpret_cast<const float4_t*>(w + i);
    uint2 pk;
    pk.x = pack_bf2(g[0] * (v[0] * r), g[1] * (v[1] * r));
    pk.y = pack_bf2(g[2] * (v[2] * r), g[3] * (v[3] * r));
    *reinterpret_cast<uint2*>(y + row * ldy + i) = pk;
  }
}

// dx = r*(w*dy) - x * r^3 * mean(x * w*dy)  (+ dres) ; optional bf16 copy for the next dgrad GEMM.
// The row (x and dy) stays in registers between the reduction and the output pass (D <= 8192: 2 x 8 float4 per thread),
// so each operand is read once.  dy may be given as `nslab` split-K partial slabs [nslab][M][ldy] of the dgrad GEMM that
// produced it (mh_gemm_rmsnorm_bwd): they are summed in slab order exactly as splitk_reduce_kernel sums them, which makes
// the fused form bit-identical to GEMM -> reduce -> rmsnorm_bwd while skipping one launch and a write + read of dy.
// Round 6: templated on the float4 chunks a thread holds (D <= 4096: 4, else 8).  With the run-time chunk count the kernel kept
// 2 x 8 float4 live (132 registers: 3 waves per SIMD = 768 resident rows, so the step's 1184-row launches ran as 1.5 rounds of a
// load -> reduce -> load(dres) -> store chain); with 4 chunks it holds x, the summed gradient AND the residual gradient (requested
// before the row sums, not after them) in <= 64 registers, every row is resident at once and a row is ONE memory round trip.
// Same expressions in the same order: bit-identical to the former kernel.
template <int NIT>
__global__ __launch_bounds__(NT, NIT <= 4 ? 5 : 2) void rmsnorm_bwd_kernel(const void* __restrict__ dy, int nslab, long slab, long ldy,
                                                         const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* dres, float* dx, bf16_t* dx_bf, int D, float eps, int sbf) {
  __shared__ float red[2 * NW];
  const size_t row = blockIdx.x;
  const float* xr = x + row * D;
  const long g0 = (long)row * ldy;                 // element offset of this row in a slab (fp32 or bf16: sbf)
  constexpr bool PRE = NIT <= 4;                   // the wide form (D > 4096) would need 256 registers with the prefetch
  float4_t xv[NIT], gv[NIT], dv[PRE ? NIT : 1];
  float ss = 0.f, dot = 0.f;
#pragma unroll
  for (int c = 0; c < NIT; ++c) {
    const int i = threadIdx.x * 4 + c * NT * 4;
    if (i < D) {
      xv[c] = *reinterpret_cast<const float4_t*>(xr + i);
      if (PRE && dres) dv[c] = *reinterpret_cast<const float4_t*>(dres + row * D + i);
    }
  }
#pragma unroll
  for (int c = 0; c < NIT; ++c) {
    const int i = threadIdx.x * 4 + c * NT * 4;
    if (i < D) {
      float4_t g = slab_load4(dy, g0 + i, sbf);
      int k = 1;
      if (nslab > 4)                                   // (the batch-8 step's 3-slab launches keep the plain loop: measured faster)
      for (; k + 4 <= nslab; k += 4) {                 // ascending order, four loads in flight (the batch-1 step's 8 slabs)
        float4_t p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = slab_load4(dy, g0 + (long)(k + u) * slab + i, sbf);
#pragma unroll
        for (int u = 0; u < 4; ++u) { g[0] += p[u][0]; g[1] += p[u][1]; g[2] += p[u][2]; g[3] += p[u][3]; }
      }
      for (; k < nslab; ++k) {
        const float4_t p = slab_load4(dy, g0 + (long)k * slab + i, sbf);
        g[0] += p[0]; g[1] += p[1]; g[2] += p[2]; g[3] += p[3];
      }
      if (nslab > 1) { g[0] *= 1.0f; g[1] *= 1.0f; g[2] *= 1.0f; g[3] *= 1.0f; }   // splitk_reduce_kernel's alpha
      gv[c] = g;
    }
  }
#pragma unroll
  for (int c = 0; c < NIT; ++c) {
    const int i = threadIdx.x * 4 + c * NT * 4;
    if (i < D) {
      const float4_t ww = *reinterpret_cast<const float4_t*>(w + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) rms_bwd_sums(xv[c][e], ww[e], gv[c][e], ss, dot);
    }
  }
  // both row sums through one pair of barriers (each is still wave_sum, then the four wave values added in wave order)
  ss = wave_sum(ss);
  dot = wave_sum(dot);
  {
    const int wv_ = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) { red[wv_] = ss; red[NW + wv_] = dot; }
    __syncthreads();
    float t = 0.f, u = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) { t += red[i]; u += red[NW + i]; }
    ss = t; dot = u;
  }
  const float r = rsqrtf(ss / D + eps);
  const float cc = r * r * r * dot / D;
#pragma unroll
  for (int c = 0; c < NIT; ++c) {
    const int i = threadIdx.x * 4 + c * NT * 4;
    if (i < D) {
      const float4_t ww = *reinterpret_cast<const float4_t*>(w + i);
      float4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rms_bwd_value(r, ww[e], gv[c][e], xv[c][e], cc);
      if (dres) {
        const float4_t d = PRE ? dv[c] : *reinterpret_cast<const float4_t*>(dres + row * D + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += d[e];
      }
      if (dx) *reinterpret_cast<float4_t*>(dx + row * D + i) = o;
      if (dx_bf) {
        uint2 pk;
        pk.x = pack_bf2(o[0], o[1]);
        pk.y = pack_bf2(o[2], o[3]);
        *reinterpret_cast<uint2*>(dx_bf + row * D + i) = pk;
      }
    }
  }
}

// ---- LayerNorm -------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, bf16_t* y_bf, float* y_f32,
                                                           int D, float eps) {
  __shared__ float red[NW];
  const size_t row = blockIdx.x;
  const float* xr = x + row * D;
  float s = 0.f;
  for (int i = threadIdx.x * 4; i < D; i += NT * 4) {
    const float4_t v = *reinterpret_cast<const float4_t*>(xr + i);
    s += v[0] + v[1] + v[2] + v[3];
  }
  const float mean = block_sum<NW>(s, red) / D;
  float ss = 0.f;
  for (int i = threadIdx.x * 4; i < D; i += NT * 4) {
    const float4_t v = *reinterpret_cast<const float4_t*>(xr + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) ss += (v[e] - mean) * (v[e] - mean);
  }
  const float r = rsqrtf(block_sum<NW>(ss, red) / D + eps);
  for (int i = threadIdx.x * 4; i < D; i += NT * 4) {
    const float4_t v = *reinterpret_cast<const float4_t*>(xr + i);
    const float4_t g = *reinterpret_cast<const float4_t*>(w + i);
    const float4_t bb = *reinterpret_cast<const float4_t*>(b + i);
    float4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * r * g[e] + bb[e];
    if (y_f32) *reinterpret_cast<float4_t*>(y_f32 + row * D + i) = o;
    if (y_bf) {
      uint2 pk;
      pk.x = pack_bf2(o[0], o[1]);
      pk.y = pack_bf2(o[2], o[3]);
      *reinterpret_cast<uint2*>(y_bf + row * D + i) = pk;
    }
  }
}

// dx = r * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*w   (+ dres)
__global__ __launch_bounds__(NT) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ w, const float* dres, float* dx,
                                                           bf16_t* dx_bf, int D, float eps) {
  __shared__ float red[NW];
  const size_t row = blockIdx.x;
  const float* xr = x + row * D;
  const float* gr = dy + row * D;
  float s = 0.f;
  for (int i = threadIdx.x * 4; i < D; i += NT * 4) {
    const float4_t v = *reinterpret_cast<const float4_t*>(xr + i);
    s += v[0] + v[1] + v[2] + v[3];
  }
  const float mean = block_sum<NW>(s, red) / D;
  float ss = 0.f;
  for (int i = threadIdx.x * 4; i < D; i += NT * 4) {
    const float4_t v = *reinterpret_cast<const float4_t*>(xr + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) ss += (v[e] - mean) * (v[e] - mean);
  }
  const float r = rsqrtf(block_sum<NW>(ss, red) / D + eps);
  float sg = 0.f, sgx = 0.f;
  for (int i = threadIdx.x * 4; i < D; i += NT * 4) {
    const float4_t v = *reinterpret_cast<const float4_t*>(xr + i);
    const float4_t g = *reinterpret_cast<const float4_t*>(gr + i);
    const float4_t ww = *reinterpret_cast<const float4_t*>(w + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = g[e] * ww[e];
      sg += gg;
      sgx += gg * (v[e] - mean) * r;
    }
  }
  sg = block_sum<NW>(sg, red) / D;
  sgx = block_sum<NW>(sgx, red) / D;
  for (int i = threadIdx.x * 4; i < D; i += NT * 4) {
    const float4_t v = *reinterpret_cast<const float4_t*>(xr + i);
    const float4_t g = *reinterpret_cast<const float4_t*>(gr + i);
    const float4_t ww = *reinterpret_cast<const float4_t*>(w + i);
    float4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = r * (g[e] * ww[e] - sg - (v[e] - mean) * r * sgx);
    if (dres) {
      const float4_t d = *reinterpret_cast<const float4_t*>(dres + row * D + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += d[e];
    }
    if (dx) *reinterpret_cast<float4_t*>(dx + row * D + i) = o;
    if (dx_bf) {
      uint2 pk;
      pk.x = pack_bf2(o[0], o[1]);
      pk.y = pack_bf2(o[2], o[3]);
      *reinterpret_cast<uint2*>(dx_bf + row * D + i) = pk;
    }
  }
}

extern "C" int mh_rmsnorm_fwd(const float* x, const float* w, void* y_bf16, long ldy, int M, int D, float eps,
                              hipStream_t stream) {
  if (M <= 0) return MH_OK;
  if (D % 4 || ldy % 4 || ldy < D) return MH_ERR_ARG;
  hipLaunchKernelGGL(rmsnorm_fwd_kernel, dim3(M), dim3(NT), 0, stream, x, w, (bf16_t*)y_bf16, D, ldy, eps);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// dy as nslab partial slabs [nslab][M][ldy] (nslab = 1: a plain [M, ldy] matrix); called from gemm.hip too
int mh_launch_rmsnorm_bwd(const void* dy, int slab_bf16, int nslab, long slab, long ldy, const float* x, const float* w,
                          const float* dres, float* dx, void* dx_bf16, int M, int D, float eps, hipStream_t stream) {
  if (M <= 0) return MH_OK;
  if ((D % 4) || D > 8192 || (ldy % 4) || nslab < 1) return MH_ERR_ARG;
  // (the slab type stays a run-time argument here: the templated form measured slower, 27.6 vs 20.1 us average in the step)
  if (D <= 4 * NT * 4)
    hipLaunchKernelGGL(rmsnorm_bwd_kernel<4>, dim3(M), dim3(NT), 0, stream, dy, nslab, slab, ldy, x, w, dres, dx, (bf16_t*)dx_bf16, D,
                       eps, slab_bf16);
  else
    hipLaunchKernelGGL(rmsnorm_bwd_kernel<8>, dim3(M), dim3(NT), 0, stream, dy, nslab, slab, ldy, x, w, dres, dx, (bf16_t*)dx_bf16, D,
                       eps, slab_bf16);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_rmsnorm_bwd(const float* dy, const float* x, const float* w, const float* dres, float* dx,
                              void* dx_bf16, int M, int D, float eps, hipStream_t stream) {
  if (M <= 0) return MH_OK;
  return mh_launch_rmsnorm_bwd(dy, 0, 1, 0, D, x, w, dres, dx, dx_bf16, M, D, eps, stream);
}

extern "C" int mh_layernorm_fwd(const float* x, const float* w, const float* b, void* y_bf16, float* y_f32, int M,
                                int D, float eps, hipStream_t stream) {
  if (M <= 0) return MH_OK;
  if (D % 4) return MH_ERR_ARG;
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(M), dim3(NT), 0, stream, x, w, b, (bf16_t*)y_bf16, y_f32, D, eps);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_layernorm_bwd(const float* dy, const float* x, const float* w, const float* dres, float* dx,
                                void* dx_bf16, int M, int D, float eps, hipStream_t stream) {
  if (M <= 0) return MH_OK;
  if (D % 4) return MH_ERR_ARG;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(M), dim3(NT), 0, stream, dy, x, w, dres, dx, (bf16_t*)dx_bf16, D,
                     eps);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

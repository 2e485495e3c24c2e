// HBM-bound elementwise / data-movement kernels (K8 rotary, K9 SiLU-gate, GELU, casts, transposes,
// embedding gather K13, strided copies).  All vectorised to 8-16 B per lane (wave64, coalesced).
#include "common.h"

#define EW_NT 256
static inline int ew_grid(long n_items) {
  long g = (n_items + EW_NT - 1) / EW_NT;
  if (g > 256L * 16) g = 256L * 16;  // grid-stride beyond 16 blocks/CU
  if (g < 1) g = 1;
  return (int)g;
}

// ---- K8 rotary (rotate-half form, gathered by position id) -----------------------------------
// reference modeling_llama.py:109-123.  x: [n_tok, ld] bf16, heads [nh] of width d starting at col0.
// cos/sin tables [max_pos, d/2] fp32.  sign=+1 forward, -1 backward (transpose of the rotation).
__global__ void rope_kernel(bf16_t* x, int ld, int col0, long n_tok, int nh, int d, const int* __restrict__ pos,
                            const float* __restrict__ cs, const float* __restrict__ sn, float sign) {
  const int half = d >> 1;
  const int per_tok = nh * (half >> 2);  // items of 4 pairs
  const long total = n_tok * per_tok;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long tok = it / per_tok;
    const int rem = (int)(it - tok * per_tok);
    const int h = rem / (half >> 2), i = (rem % (half >> 2)) * 4;
    bf16_t* p = x + tok * ld + col0 + h * d + i;
    const int ps = pos[tok];
    const float4_t c = *reinterpret_cast<const float4_t*>(cs + (size_t)ps * half + i);
    const float4_t s = *reinterpret_cast<const float4_t*>(sn + (size_t)ps * half + i);
    const short4_t a = *reinterpret_cast<const short4_t*>(p);
    const short4_t b = *reinterpret_cast<const short4_t*>(p + half);
    short4_t oa, ob;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x1 = bf2f((bf16_t)a[e]), x2 = bf2f((bf16_t)b[e]);
      oa[e] = (short)f2bf(x1 * c[e] - sign * x2 * s[e]);
      ob[e] = (short)f2bf(x2 * c[e] + sign * x1 * s[e]);
    }
    *reinterpret_cast<short4_t*>(p) = oa;
    *reinterpret_cast<short4_t*>(p + half) = ob;
  }
}

extern "C" int mh_rope_inplace(void* x, int ld, int col0, int n_tok, int n_heads, int head_dim, const int* pos,
                               const float* cos_tab, const float* sin_tab, float sign, hipStream_t stream) {
  if (n_tok <= 0) return MH_OK;
  if (head_dim % 8 || ld % 4 || col0 % 4) return MH_ERR_ARG;
  const long items = (long)n_tok * n_heads * (head_dim / 8);
  hipLaunchKernelGGL(rope_kernel, dim3(ew_grid(items)), dim3(EW_NT), 0, stream, (bf16_t*)x, ld, col0, (long)n_tok,
                     n_heads, head_dim, pos, cos_tab, sin_tab, sign);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ---- K9 SiLU-gated MLP: h = silu(g) * u, gu = [M, 2I] = [gate | up] ---------------------------
// reference modeling_llama.py:139-140
// blk = 0: gu row = [g 0..I-1 | u 0..I-1]; blk = 128: gate / up interleaved in blocks of blk columns (the layout the fused
// gate|up GEMM epilogue needs, gemm.hip mh_gemm_swiglu_*): g column c sits at (c / blk) * 2 blk + c % blk, u at + blk
__device__ __forceinline__ long silu_gcol(int c, int I, int blk, int* ustep) {
  if (blk == 0) { *ustep = I; return c; }
  *ustep = blk;
  return (long)(c / blk) * 2 * blk + (c % blk);
}

// ldg / ldh: row strides of gu / h in elements (0: dense, 2 I / I) -- a column slice of a wider pair of buffers (gemm.hip:
// mh_gemm_swiglu_fwd's left-over column blocks)
__global__ void silu_mul_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ h, long M, int I, int blk, long ldg = 0,
                                    long ldh = 0) {
  const int per_row = I >> 3;
  const long total = M * per_row;
  ldg = ldg ? ldg : 2L * I;
  ldh = ldh ? ldh : (long)I;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long m = it / per_row;
    const int c = (int)(it - m * per_row) * 8;
    int us;
    const long gc = silu_gcol(c, I, blk, &us);
    const short8_t g = *reinterpret_cast<const short8_t*>(gu + m * ldg + gc);
    const short8_t u = *reinterpret_cast<const short8_t*>(gu + m * ldg + gc + us);
    short8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gv = bf2f((bf16_t)g[e]), uv = bf2f((bf16_t)u[e]);
      o[e] = (short)f2bf(gv / (1.f + __expf(-gv)) * uv);
    }
    *reinterpret_cast<short8_t*>(h + m * ldh + c) = o;
  }
}

// nslab > 0: dh is not a bf16 matrix but the split-K partial slabs [nslab][M][I] (fp32, or bf16 when sbf) of the down projection's
// dgrad (mh_gemm_swiglu_bwd at the batch-1 step): summed here in slab order and rounded to bf16 as splitk_reduce_kernel does, so
// the same bits as GEMM -> reduce -> this kernel with one launch and one round trip of dact less.
__global__ void silu_mul_bwd_kernel(const bf16_t* __restrict__ dh, const bf16_t* __restrict__ gu,
                                    bf16_t* __restrict__ dgu, long M, int I, int blk, int nslab = 0, long slab = 0, int sbf = 0) {
  const int per_row = I >> 3;
  const long total = M * per_row;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long m = it / per_row;
    const int c = (int)(it - m * per_row) * 8;
    int us;
    const long gc = silu_gcol(c, I, blk, &us);
    const short8_t g = *reinterpret_cast<const short8_t*>(gu + m * 2 * I + gc);
    const short8_t u = *reinterpret_cast<const short8_t*>(gu + m * 2 * I + gc + us);
    short8_t d;
    if (nslab > 0) {
      const void* ws = dh;
      float4_t s0 = slab_load4(ws, m * I + c, sbf), s1 = slab_load4(ws, m * I + c + 4, sbf);
      for (int k = 1; k < nslab; ++k) {
        const float4_t p0 = slab_load4(ws, (long)k * slab + m * I + c, sbf), p1 = slab_load4(ws, (long)k * slab + m * I + c + 4, sbf);
        s0[0] += p0[0]; s0[1] += p0[1]; s0[2] += p0[2]; s0[3] += p0[3];
        s1[0] += p1[0]; s1[1] += p1[1]; s1[2] += p1[2]; s1[3] += p1[3];
      }
      // splitk_reduce_kernel: v = s * alpha (alpha = 1), then the packed bf16 conversion
      const unsigned q0 = pack_bf2(s0[0] * 1.0f, s0[1] * 1.0f), q1 = pack_bf2(s0[2] * 1.0f, s0[3] * 1.0f);
      const unsigned q2 = pack_bf2(s1[0] * 1.0f, s1[1] * 1.0f), q3 = pack_bf2(s1[2] * 1.0f, s1[3] * 1.0f);
      d = (short8_t){(short)(q0 & 0xffffu), (short)(q0 >> 16), (short)(q1 & 0xffffu), (short)(q1 >> 16),
                     (short)(q2 & 0xffffu), (short)(q2 >> 16), (short)(q3 & 0xffffu), (short)(q3 >> 16)};
    } else {
      d = *reinterpret_cast<const short8_t*>(dh + m * I + c);
    }
    short8_t og, ou;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gv = bf2f((bf16_t)g[e]), uv = bf2f((bf16_t)u[e]), dv = bf2f((bf16_t)d[e]);
      const float sg = 1.f / (1.f + __expf(-gv));
      const float silu = gv * sg;
      og[e] = (short)f2bf(dv * uv * (sg + silu * (1.f - sg)));
      ou[e] = (short)f2bf(dv * silu);
    }
    *reinterpret_cast<short8_t*>(dgu + m * 2 * I + gc) = og;
    *reinterpret_cast<short8_t*>(dgu + m * 2 * I + gc + us) = ou;
  }
}

extern "C" int mh_silu_mul_fwd_blk(const void* gu, void* h, int M, int I, int blk, hipStream_t stream) {
  if (M <= 0) return MH_OK;
  if (I % 8 || blk < 0 || (blk && ((blk % 8) || (I % blk)))) return MH_ERR_ARG;
  hipLaunchKernelGGL(silu_mul_fwd_kernel, dim3(ew_grid((long)M * (I / 8))), dim3(EW_NT), 0, stream,
                     (const bf16_t*)gu, (bf16_t*)h, (long)M, I, blk);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
// a column slice: gu [M, 2 I] at row stride ldg, h [M, I] at row stride ldh (gemm.hip)
int mh_launch_silu_mul_fwd_blk_ld(const void* gu, long ldg, void* h, long ldh, int M, int I, int blk, hipStream_t stream) {
  if (M <= 0) return MH_OK;
  if (I % 8 || blk <= 0 || (blk % 8) || (I % blk) || (ldg % 8) || (ldh % 8) || ldg < 2L * I || ldh < I) return MH_ERR_ARG;
  hipLaunchKernelGGL(silu_mul_fwd_kernel, dim3(ew_grid((long)M * (I / 8))), dim3(EW_NT), 0, stream,
                     (const bf16_t*)gu, (bf16_t*)h, (long)M, I, blk, ldg, ldh);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
extern "C" int mh_silu_mul_bwd_blk(const void* dh, const void* gu, void* dgu, int M, int I, int blk, hipStream_t stream) {
  if (M <= 0) return MH_OK;
  if (I % 8 || blk < 0 || (blk && ((blk % 8) || (I % blk)))) return MH_ERR_ARG;
  hipLaunchKernelGGL(silu_mul_bwd_kernel, dim3(ew_grid((long)M * (I / 8))), dim3(EW_NT), 0, stream,
                     (const bf16_t*)dh, (const bf16_t*)gu, (bf16_t*)dgu, (long)M, I, blk);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
// dh as split-K slabs (gemm.hip mh_gemm_swiglu_bwd)
int mh_launch_silu_mul_bwd_slabs(const void* ws, int slab_bf16, int nslab, long slab, const void* gu, void* dgu, int M, int I, int blk,
                                 hipStream_t stream) {
  if (M <= 0) return MH_OK;
  if (I % 8 || blk < 0 || (blk && ((blk % 8) || (I % blk))) || nslab < 1) return MH_ERR_ARG;
  hipLaunchKernelGGL(silu_mul_bwd_kernel, dim3(ew_grid((long)M * (I / 8))), dim3(EW_NT), 0, stream, (const bf16_t*)ws,
                     (const bf16_t*)gu, (bf16_t*)dgu, (long)M, I, blk, nslab, slab, slab_bf16);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
extern "C" int mh_silu_mul_fwd(const void* gu, void* h, int M, int I, hipStream_t stream) {
  return mh_silu_mul_fwd_blk(gu, h, M, I, 0, stream);
}
extern "C" int mh_silu_mul_bwd(const void* dh, const void* gu, void* dgu, int M, int I, hipStream_t stream) {
  return mh_silu_mul_bwd_blk(dh, gu, dgu, M, I, 0, stream);
}

// ---- GELU (erf form) on bf16, fwd and bwd ----------------------------------------------------
__global__ void gelu_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long n8) {
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < n8; it += (long)gridDim.x * blockDim.x) {
    const short8_t v = *reinterpret_cast<const short8_t*>(x + it * 8);
    short8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(gelu_erf(bf2f((bf16_t)v[e])));
    *reinterpret_cast<short8_t*>(y + it * 8) = o;
  }
}
__global__ void gelu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, bf16_t* __restrict__ dx,
                                long n8) {
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < n8; it += (long)gridDim.x * blockDim.x) {
    const short8_t v = *reinterpret_cast<const short8_t*>(x + it * 8);
    const short8_t g = *reinterpret_cast<const short8_t*>(dy + it * 8);
    short8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(bf2f((bf16_t)g[e]) * gelu_erf_grad(bf2f((bf16_t)v[e])));
    *reinterpret_cast<short8_t*>(dx + it * 8) = o;
  }
}
extern "C" int mh_gelu_fwd(const void* x, void* y, long n, hipStream_t stream) {
  if (n <= 0) return MH_OK;
  if (n % 8) return MH_ERR_ARG;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(ew_grid(n / 8)), dim3(EW_NT), 0, stream, (const bf16_t*)x, (bf16_t*)y,
                     n / 8);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
extern "C" int mh_gelu_bwd(const void* dy, const void* x, void* dx, long n, hipStream_t stream) {
  if (n <= 0) return MH_OK;
  if (n % 8) return MH_ERR_ARG;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_grid(n / 8)), dim3(EW_NT), 0, stream, (const bf16_t*)dy,
                     (const bf16_t*)x, (bf16_t*)dx, n / 8);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ---- casts -----------------------------------------------------------------------------------
__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n4) {
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < n4; it += (long)gridDim.x * blockDim.x) {
    const float4_t v = *reinterpret_cast<const float4_t*>(x + it * 4);
    uint2 pk;
    pk.x = pack_bf2(v[0], v[1]);
    pk.y = pack_bf2(v[2], v[3]);
    *reinterpret_cast<uint2*>(y + it * 4) = pk;
  }
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long n4) {
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < n4; it += (long)gridDim.x * blockDim.x) {
    const short4_t v = *reinterpret_cast<const short4_t*>(x + it * 4);
    *reinterpret_cast<float4_t*>(y + it * 4) =
        (float4_t){bf2f((bf16_t)v[0]), bf2f((bf16_t)v[1]), bf2f((bf16_t)v[2]), bf2f((bf16_t)v[3])};
  }
}
extern "C" int mh_cast_f32_to_bf16(const float* x, void* y, long n, hipStream_t stream) {
  if (n <= 0) return MH_OK;
  if (n % 4) return MH_ERR_ARG;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid(n / 4)), dim3(EW_NT), 0, stream, x, (bf16_t*)y, n / 4);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
extern "C" int mh_cast_bf16_to_f32(const void* x, float* y, long n, hipStream_t stream) {
  if (n <= 0) return MH_OK;
  if (n % 4) return MH_ERR_ARG;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(ew_grid(n / 4)), dim3(EW_NT), 0, stream, (const bf16_t*)x, y, n / 4);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ---- 2-D strided copies (concat / slice assembly, K13) ---------------------------------------
// dst[r*ldd + c] (=|+=) src[r*lds + c], fp32, cols % 4 == 0
__global__ void copy2d_f32_kernel(const float* __restrict__ src, long lds, float* dst, long ldd, long rows, int cols4,
                                  int accumulate) {
  const long total = rows * cols4;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long r = it / cols4;
    const int c = (int)(it - r * cols4) * 4;
    float4_t v = *reinterpret_cast<const float4_t*>(src + r * lds + c);
    if (accumulate) {
      const float4_t d = *reinterpret_cast<const float4_t*>(dst + r * ldd + c);
      v[0] += d[0]; v[1] += d[1]; v[2] += d[2]; v[3] += d[3];
    }
    *reinterpret_cast<float4_t*>(dst + r * ldd + c) = v;
  }
}
extern "C" int mh_copy2d_f32(const float* src, long lds, float* dst, long ldd, long rows, int cols, int accumulate,
                             hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return MH_OK;
  if (cols % 4 || lds % 4 || ldd % 4) return MH_ERR_ARG;
  hipLaunchKernelGGL(copy2d_f32_kernel, dim3(ew_grid(rows * (cols / 4))), dim3(EW_NT), 0, stream, src, lds, dst, ldd,
                     rows, cols / 4, accumulate);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// batched variant: dst[b][r][c] = src[b][r][c] with batch strides (elements)
__global__ void copy3d_f32_kernel(const float* __restrict__ src, long sb, long lds, float* dst, long db, long ldd,
                                  int nb, long rows, int cols4, int accumulate) {
  const long per_b = rows * cols4;
  const long total = per_b * nb;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long b = it / per_b;
    const long rem = it - b * per_b;
    const long r = rem / cols4;
    const int c = (int)(rem - r * cols4) * 4;
    float4_t v = *reinterpret_cast<const float4_t*>(src + b * sb + r * lds + c);
    float* d = dst + b * db + r * ldd + c;
    if (accumulate) {
      const float4_t o = *reinterpret_cast<const float4_t*>(d);
      v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
    }
    *reinterpret_cast<float4_t*>(d) = v;
  }
}
extern "C" int mh_copy3d_f32(const float* src, long src_bstride, long lds, float* dst, long dst_bstride, long ldd,
                             int nb, long rows, int cols, int accumulate, hipStream_t stream) {
  if (rows <= 0 || cols <= 0 || nb <= 0) return MH_OK;
  if (cols % 4 || lds % 4 || ldd % 4 || src_bstride % 4 || dst_bstride % 4) return MH_ERR_ARG;
  hipLaunchKernelGGL(copy3d_f32_kernel, dim3(ew_grid(nb * rows * (cols / 4))), dim3(EW_NT), 0, stream, src,
                     src_bstride, lds, dst, dst_bstride, ldd, nb, rows, cols / 4, accumulate);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ---- K13 embedding gather: out[dst_row[i]] = table[ids[i]] (bf16 table -> fp32 residual stream)
__global__ void embed_gather_kernel(const bf16_t* __restrict__ table, const long* __restrict__ ids,
                                    const int* __restrict__ dst_rows, float* out, long n, int D, long ldo) {
  const int per_row = D >> 3;
  const long total = n * per_row;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long i = it / per_row;
    const int c = (int)(it - i * per_row) * 8;
    const short8_t v = *reinterpret_cast<const short8_t*>(table + ids[i] * D + c);
    float* o = out + (long)(dst_rows ? dst_rows[i] : i) * ldo + c;
    *reinterpret_cast<float4_t*>(o) =
        (float4_t){bf2f((bf16_t)v[0]), bf2f((bf16_t)v[1]), bf2f((bf16_t)v[2]), bf2f((bf16_t)v[3])};
    *reinterpret_cast<float4_t*>(o + 4) =
        (float4_t){bf2f((bf16_t)v[4]), bf2f((bf16_t)v[5]), bf2f((bf16_t)v[6]), bf2f((bf16_t)v[7])};
  }
}
extern "C" int mh_embed_gather(const void* table_bf16, const long* ids, const int* dst_rows, float* out, long n, int D,
                               long ldo, hipStream_t stream) {
  if (n <= 0) return MH_OK;
  if (D % 8 || ldo % 4) return MH_ERR_ARG;
  hipLaunchKernelGGL(embed_gather_kernel, dim3(ew_grid(n * (D / 8))), dim3(EW_NT), 0, stream,
                     (const bf16_t*)table_bf16, ids, dst_rows, out, n, D, ldo);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// gather rows of an fp32 matrix (used to pick the label-bearing rows before lm_head) -> bf16
__global__ void gather_rows_f32_bf16_kernel(const float* __restrict__ src, long lds, const int* __restrict__ rows,
                                            bf16_t* __restrict__ dst, long n, int D) {
  const int per_row = D >> 2;
  const long total = n * per_row;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long i = it / per_row;
    const int c = (int)(it - i * per_row) * 4;
    const float4_t v = *reinterpret_cast<const float4_t*>(src + (long)rows[i] * lds + c);
    uint2 pk;
    pk.x = pack_bf2(v[0], v[1]);
    pk.y = pack_bf2(v[2], v[3]);
    *reinterpret_cast<uint2*>(dst + i * D + c) = pk;
  }
}
// scatter-add rows back: dst[rows[i]] += src[i]  (fp32); rows must be unique
__global__ void scatter_rows_f32_kernel(const float* __restrict__ src, const int* __restrict__ rows, float* dst,
                                        long ldd, long n, int D, int accumulate) {
  const int per_row = D >> 2;
  const long total = n * per_row;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long i = it / per_row;
    const int c = (int)(it - i * per_row) * 4;
    float4_t v = *reinterpret_cast<const float4_t*>(src + i * D + c);
    float* d = dst + (long)rows[i] * ldd + c;
    if (accumulate) {
      const float4_t o = *reinterpret_cast<const float4_t*>(d);
      v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
    }
    *reinterpret_cast<float4_t*>(d) = v;
  }
}
extern "C" int mh_gather_rows_f32_to_bf16(const float* src, long lds, const int* rows, void* dst, long n, int D,
                                          hipStream_t stream) {
  if (n <= 0) return MH_OK;
  if (D % 4 || lds % 4) return MH_ERR_ARG;
  hipLaunchKernelGGL(gather_rows_f32_bf16_kernel, dim3(ew_grid(n * (D / 4))), dim3(EW_NT), 0, stream, src, lds, rows,
                     (bf16_t*)dst, n, D);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
extern "C" int mh_scatter_rows_f32(const float* src, const int* rows, float* dst, long ldd, long n, int D,
                                   int accumulate, hipStream_t stream) {
  if (n <= 0) return MH_OK;
  if (D % 4 || ldd % 4) return MH_ERR_ARG;
  hipLaunchKernelGGL(scatter_rows_f32_kernel, dim3(ew_grid(n * (D / 4))), dim3(EW_NT), 0, stream, src, rows, dst, ldd,
                     n, D, accumulate);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ---- transpose to bf16 with zero padding: out[c][r] = in[r][c], out is [C, ldo], r >= R -> 0 ----
// 64x64 tiles through LDS, four elements per thread per access on both sides (8-byte bf16 / 16-byte fp32 loads, 8-byte
// stores) when strides and bases allow it: the 2-byte-per-lane form ran at 1.1 TB/s on the 210 MB VETokenizer head.
// IN_F32 selects fp32 or bf16 input; the tile holds bf16 bits (row stride 66 halves: column reads spread over the banks).
template <bool IN_F32>
__global__ __launch_bounds__(256) void transpose_kernel(const void* __restrict__ in_, long ldi, bf16_t* __restrict__ out,
                                                        long ldo, int R, int C, int vec_in, int vec_out) {
  __shared__ unsigned short tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int q = threadIdx.x; q < 64 * 16; q += 256) {
    const int i = q >> 4, c4 = (q & 15) * 4;
    const int r = r0 + i, c = c0 + c4;
    unsigned short v[4] = {0, 0, 0, 0};
    if (r < R) {
      if (vec_in && c + 3 < C) {
        if (IN_F32) {
          const float4_t f = *reinterpret_cast<const float4_t*>(reinterpret_cast<const float*>(in_) + (long)r * ldi + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (unsigned short)f2bf(f[e]);
        } else {
          const short4_t h = *reinterpret_cast<const short4_t*>(reinterpret_cast<const bf16_t*>(in_) + (long)r * ldi + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (unsigned short)h[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < C)
            v[e] = IN_F32 ? (unsigned short)f2bf(reinterpret_cast<const float*>(in_)[(long)r * ldi + c + e])
                          : (unsigned short)reinterpret_cast<const bf16_t*>(in_)[(long)r * ldi + c + e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[i][c4 + e] = v[e];
  }
  __syncthreads();
  for (int q = threadIdx.x; q < 64 * 16; q += 256) {
    const int i = q >> 4, r4 = (q & 15) * 4;
    const int c = c0 + i, r = r0 + r4;
    if (c >= C || r >= ldo) continue;
    if (vec_out && r + 3 < ldo) {
      short4_t h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (short)tile[r4 + e][i];
      *reinterpret_cast<short4_t*>(out + (long)c * ldo + r) = h;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (r + e < ldo) out[(long)c * ldo + r + e] = (bf16_t)tile[r4 + e][i];
    }
  }
}
extern "C" int mh_transpose_to_bf16(const void* in, int in_is_f32, long ldi, void* out, long ldo, int R, int C,
                                    hipStream_t stream) {
  if (R <= 0 || C <= 0) return MH_OK;
  if (ldo < R) return MH_ERR_ARG;
  // cover [0, ldo) along R so the padding columns are zero-filled
  const dim3 grid((C + 63) / 64, (int)((ldo + 63) / 64));
  const int vec_in = (ldi % 4 == 0) && ((uintptr_t)in % (in_is_f32 ? 16 : 8) == 0);
  const int vec_out = (ldo % 4 == 0) && ((uintptr_t)out % 8 == 0);
  if (in_is_f32)
    hipLaunchKernelGGL(transpose_kernel<true>, grid, dim3(256), 0, stream, in, ldi, (bf16_t*)out, ldo, R, C, vec_in, vec_out);
  else
    hipLaunchKernelGGL(transpose_kernel<false>, grid, dim3(256), 0, stream, in, ldi, (bf16_t*)out, ldo, R, C, vec_in, vec_out);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ---- column sums (bias gradients): out[c] = sum_r in[r][c], fp32 in -----------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, long ld, float* __restrict__ out,
                                                     long R, int C) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ty = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C)
    for (long r = ty; r < R; r += 4) s += in[r * ld + c];
  part[ty][threadIdx.x & 63] = s;
  __syncthreads();
  if (ty == 0 && c < C) out[c] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}
extern "C" int mh_colsum_f32(const float* in, long ld, float* out, long R, int C, hipStream_t stream) {
  if (C <= 0) return MH_OK;
  hipLaunchKernelGGL(colsum_kernel, dim3((C + 63) / 64), dim3(256), 0, stream, in, ld, out, R, C);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ---- scale / axpy on fp32 ----------------------------------------------------------------------
__global__ void scale_f32_kernel(float* x, float a, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= a;
}
extern "C" int mh_scale_f32(float* x, float a, long n, hipStream_t stream) {
  if (n <= 0) return MH_OK;
  hipLaunchKernelGGL(scale_f32_kernel, dim3(ew_grid(n)), dim3(EW_NT), 0, stream, x, a, n);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ---- gather rows f32 -> f32 --------------------------------------------------------------------
__global__ void gather_rows_f32_kernel(const float* __restrict__ src, long lds, const int* __restrict__ rows,
                                       float* __restrict__ dst, long n, int D) {
  const int per_row = D >> 2;
  const long total = n * per_row;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long i = it / per_row;
    const int c = (int)(it - i * per_row) * 4;
    *reinterpret_cast<float4_t*>(dst + i * D + c) = *reinterpret_cast<const float4_t*>(src + (long)rows[i] * lds + c);
  }
}
extern "C" int mh_gather_rows_f32(const float* src, long lds, const int* rows, float* dst, long n, int D,
                                  hipStream_t stream) {
  if (n <= 0) return MH_OK;
  if (D % 4 || lds % 4) return MH_ERR_ARG;
  hipLaunchKernelGGL(gather_rows_f32_kernel, dim3(ew_grid(n * (D / 4))), dim3(EW_NT), 0, stream, src, lds, rows, dst,
                     n, D);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ---- row gather of 16-byte units (any element type: bf16 rows of D % 8 == 0, f32 rows of D % 4 == 0) and its inverse with
// zero fill: dst[m, :] = inv[m] >= 0 ? src[inv[m], :] : 0 for every row m of dst -- one launch instead of a fill + a scatter.
// Used where only the label-bearing rows of a [B*S, D] tensor are non-zero / needed (llama.py: the last decoder layer's
// o_proj and MLP, modeling_llama.py:281-293, run on those rows only: every other row's output feeds nothing and its gradient is
// exactly zero).
__global__ void gather_rows16_kernel(const uint4* __restrict__ src, long lds16, const int* __restrict__ rows, uint4* __restrict__ dst,
                                     long n, int d16) {
  const long total = n * d16;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long i = it / d16;
    const int c = (int)(it - i * d16);
    dst[i * d16 + c] = src[(long)rows[i] * lds16 + c];
  }
}
__global__ void expand_rows16_kernel(const uint4* __restrict__ src, const int* __restrict__ inv, uint4* __restrict__ dst, long ldd16,
                                     long M, int d16) {
  const long total = M * d16;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long m = it / d16;
    const int c = (int)(it - m * d16);
    const int i = inv[m];
    dst[m * ldd16 + c] = i >= 0 ? src[(long)i * d16 + c] : make_uint4(0u, 0u, 0u, 0u);
  }
}
// src [*, lds] rows of D elements of elem_bytes (2 or 4) each -> dst [n, D] dense
extern "C" int mh_gather_rows(const void* src, long lds, const int* rows, void* dst, long n, int D, int elem_bytes,
                              hipStream_t stream) {
  if (n <= 0) return MH_OK;
  const int per16 = 16 / (elem_bytes > 0 ? elem_bytes : 1);
  if ((elem_bytes != 2 && elem_bytes != 4) || D % per16 || lds % per16 || (((uintptr_t)src | (uintptr_t)dst) & 15)) return MH_ERR_ARG;
  hipLaunchKernelGGL(gather_rows16_kernel, dim3(ew_grid(n * (D / per16))), dim3(EW_NT), 0, stream, (const uint4*)src, lds / per16, rows,
                     (uint4*)dst, n, D / per16);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
// dst [M, ldd] : row m = src row inv[m] (src dense [*, D]) or zeros when inv[m] < 0
extern "C" int mh_expand_rows(const void* src, const int* inv, void* dst, long ldd, long M, int D, int elem_bytes,
                              hipStream_t stream) {
  if (M <= 0) return MH_OK;
  const int per16 = 16 / (elem_bytes > 0 ? elem_bytes : 1);
  if ((elem_bytes != 2 && elem_bytes != 4) || D % per16 || ldd % per16 || (((uintptr_t)src | (uintptr_t)dst) & 15)) return MH_ERR_ARG;
  hipLaunchKernelGGL(expand_rows16_kernel, dim3(ew_grid(M * (D / per16))), dim3(EW_NT), 0, stream, (const uint4*)src, inv, (uint4*)dst,
                     ldd / per16, M, D / per16);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ---- batched 2-D copy of bf16 (KV-cache append / slicing); cols % 8 == 0 --------------------------
__global__ void copy3d_bf16_kernel(const bf16_t* __restrict__ src, long sb, long lds, bf16_t* __restrict__ dst,
                                   long db, long ldd, int nb, long rows, int cols8) {
  const long per_b = rows * cols8;
  const long total = per_b * nb;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long b = it / per_b;
    const long rem = it - b * per_b;
    const long r = rem / cols8;
    const int c = (int)(rem - r * cols8) * 8;
    *reinterpret_cast<short8_t*>(dst + b * db + r * ldd + c) =
        *reinterpret_cast<const short8_t*>(src + b * sb + r * lds + c);
  }
}
extern "C" int mh_copy3d_bf16(const void* src, long src_bstride, long lds, void* dst, long dst_bstride, long ldd,
                              int nb, long rows, int cols, hipStream_t stream) {
  if (rows <= 0 || cols <= 0 || nb <= 0) return MH_OK;
  if (cols % 8 || lds % 8 || ldd % 8 || src_bstride % 8 || dst_bstride % 8) return MH_ERR_ARG;
  hipLaunchKernelGGL(copy3d_bf16_kernel, dim3(ew_grid(nb * rows * (cols / 8))), dim3(EW_NT), 0, stream,
                     (const bf16_t*)src, src_bstride, lds, (bf16_t*)dst, dst_bstride, ldd, nb, rows, cols / 8);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ---- K14 patchify: NCHW f32 image -> [B*np, Kpad] bf16 rows in conv-weight order (c, iy, ix) --------
// reference eva_vit.py:196-204 (Conv2d kernel = stride = patch) expressed as a GEMM operand.
__global__ void patchify_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int B, int C, int H, int W,
                                int P, int K, int Kpad) {
  const int gw = W / P, gh = H / P;
  const long total = (long)B * gh * gw * Kpad;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const int k = (int)(it % Kpad);
    const long m = it / Kpad;
    float v = 0.f;
    if (k < K) {
      const int px = (int)(m % gw), py = (int)((m / gw) % gh), b = (int)(m / ((long)gw * gh));
      const int c = k / (P * P), r = k - c * P * P, iy = r / P, ix = r - iy * P;
      v = img[(((long)b * C + c) * H + py * P + iy) * W + px * P + ix];
    }
    out[it] = f2bf(v);
  }
}
extern "C" int mh_patchify_nchw(const float* img, void* out, int B, int C, int H, int W, int P, int Kpad,
                                hipStream_t stream) {
  const int K = C * P * P;
  if (H % P || W % P || Kpad < K) return MH_ERR_ARG;
  const long total = (long)B * (H / P) * (W / P) * Kpad;
  hipLaunchKernelGGL(patchify_kernel, dim3(ew_grid(total)), dim3(EW_NT), 0, stream, img, (bf16_t*)out, B, C, H, W, P, K,
                     Kpad);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ---- counter-based dropout (PEFT lora_dropout, reference myriad.py:171-178): keep-mask = hash(seed, index) >= p ----
// The mask is a pure function of (seed, flat element index), so backward regenerates it instead of storing it.
// (the keep-mask hash itself is dropout_keep in common.h, shared with lora.hip)
// y[r][c] = x[r][c] * mask/(1-p), mask index = r*cols + c
__global__ void dropout_bf16_kernel(const bf16_t* __restrict__ x, long ldx, bf16_t* __restrict__ y, long ldy, long rows,
                                    int cols8, float p, unsigned long long seed) {
  const float ik = 1.f / (1.f - p);
  const long total = rows * cols8;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long r = it / cols8;
    const int c = (int)(it - r * cols8) * 8;
    const short8_t v = *reinterpret_cast<const short8_t*>(x + r * ldx + c);
    short8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      o[e] = (short)f2bf(bf2f((bf16_t)v[e]) * dropout_keep(seed, (unsigned long long)(r * (long)cols8 * 8 + c + e), p, ik));
    *reinterpret_cast<short8_t*>(y + r * ldy + c) = o;
  }
}
// acc[r][c] += dy[r][c] * mask/(1-p), mask index = r*cols + c (contiguous index space of the forward tensor)
__global__ void dropout_add_f32_kernel(const float* __restrict__ dy, long lddy, float* acc, long ldacc, long rows,
                                       int cols4, float p, unsigned long long seed) {
  const float ik = 1.f / (1.f - p);
  const long total = rows * cols4;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long r = it / cols4;
    const int c = (int)(it - r * cols4) * 4;
    const float4_t g = *reinterpret_cast<const float4_t*>(dy + r * lddy + c);
    float4_t a = *reinterpret_cast<const float4_t*>(acc + r * ldacc + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] += g[e] * dropout_keep(seed, (unsigned long long)(r * (long)cols4 * 4 + c + e), p, ik);
    *reinterpret_cast<float4_t*>(acc + r * ldacc + c) = a;
  }
}
extern "C" int mh_dropout_bf16(const void* x, long ldx, void* y, long ldy, long rows, int cols, float p,
                               unsigned long long seed, hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return MH_OK;
  if (cols % 8 || ldx % 8 || ldy % 8 || p < 0.f || p >= 1.f) return MH_ERR_ARG;
  hipLaunchKernelGGL(dropout_bf16_kernel, dim3(ew_grid(rows * (cols / 8))), dim3(EW_NT), 0, stream, (const bf16_t*)x, ldx,
                     (bf16_t*)y, ldy, rows, cols / 8, p, seed);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
extern "C" int mh_dropout_add_f32(const float* dy, long lddy, float* acc, long ldacc, long rows, int cols, float p,
                                  unsigned long long seed, hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return MH_OK;
  if (cols % 4 || lddy % 4 || ldacc % 4 || p < 0.f || p >= 1.f) return MH_ERR_ARG;
  hipLaunchKernelGGL(dropout_add_f32_kernel, dim3(ew_grid(rows * (cols / 4))), dim3(EW_NT), 0, stream, dy, lddy, acc, ldacc,
                     rows, cols / 4, p, seed);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ---- KV-cache append at a DEVICE-resident position (so the decode step can be replayed from a hipGraph) ----------
// cache[b, pos[0], :] = src[b, :]   (reference modeling_llama.py:190-195 concatenates; we write in place)
__global__ void kv_append_kernel(const bf16_t* __restrict__ src, long ld_src, bf16_t* __restrict__ cache, long cache_bs,
                                 long ld_cache, const int* __restrict__ pos, int B, int cols8) {
  const long total = (long)B * cols8;
  const long p = pos[0];
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long b = it / cols8;
    const int c = (int)(it - b * cols8) * 8;
    *reinterpret_cast<short8_t*>(cache + b * cache_bs + p * ld_cache + c) =
        *reinterpret_cast<const short8_t*>(src + b * ld_src + c);
  }
}
extern "C" int mh_kv_append_bf16(const void* src, long ld_src, void* cache, long cache_bstride, long ld_cache,
                                 const int* pos_dev, int B, int cols, hipStream_t stream) {
  if (B <= 0 || cols <= 0) return MH_OK;
  if (cols % 8 || ld_src % 8 || ld_cache % 8 || cache_bstride % 8) return MH_ERR_ARG;
  hipLaunchKernelGGL(kv_append_kernel, dim3(ew_grid((long)B * (cols / 8))), dim3(EW_NT), 0, stream, (const bf16_t*)src,
                     ld_src, (bf16_t*)cache, cache_bstride, ld_cache, pos_dev, B, cols / 8);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ---- decode token: rotary on q (in place) and k, then k|v straight into the cache row pos_dev[0] -- one launch
// instead of rope + kv_append (modeling_llama.py:186-195).  qkv: [B, 3W] bf16 = [q | k | v], W = H*d; cache row
// layout [k | v].  Same arithmetic as rope_kernel (fp32 rotate-half, one rounding to bf16).
__global__ void rope_kv_append_kernel(bf16_t* qkv, long ld, int H, int d, const int* __restrict__ pos,
                                      const float* __restrict__ cs, const float* __restrict__ sn,
                                      bf16_t* __restrict__ cache, long cache_bs, long ld_cache,
                                      const int* __restrict__ pos_dev, int B) {
  const int half = d >> 1, W = H * d;
  const int rope_items = 2 * H * (half >> 2);      // q heads then k heads, 4 pairs per item
  const int per_tok = rope_items + (W >> 3);       // + v copy, 8 elements per item
  const long total = (long)B * per_tok;
  const long row = pos_dev[0];
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long b = it / per_tok;
    const int rem = (int)(it - b * per_tok);
    bf16_t* src = qkv + b * ld;
    bf16_t* crow = cache + b * cache_bs + row * ld_cache;
    if (rem >= rope_items) {
      const int c = (rem - rope_items) * 8;
      *reinterpret_cast<short8_t*>(crow + W + c) = *reinterpret_cast<const short8_t*>(src + 2 * W + c);
      continue;
    }
    const int h = rem / (half >> 2), i = (rem % (half >> 2)) * 4;   // h in [0, 2H): q heads, then k heads
    bf16_t* p = src + h * d + i;
    const int ps = pos[b];
    const float4_t c4 = *reinterpret_cast<const float4_t*>(cs + (size_t)ps * half + i);
    const float4_t s4 = *reinterpret_cast<const float4_t*>(sn + (size_t)ps * half + i);
    const short4_t a = *reinterpret_cast<const short4_t*>(p);
    const short4_t bb = *reinterpret_cast<const short4_t*>(p + half);
    short4_t oa, ob;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x1 = bf2f((bf16_t)a[e]), x2 = bf2f((bf16_t)bb[e]);
      oa[e] = (short)f2bf(x1 * c4[e] - x2 * s4[e]);
      ob[e] = (short)f2bf(x2 * c4[e] + x1 * s4[e]);
    }
    bf16_t* dst = h < H ? p : crow + (h - H) * d + i;               // q in place, k into the cache
    *reinterpret_cast<short4_t*>(dst) = oa;
    *reinterpret_cast<short4_t*>(dst + half) = ob;
  }
}
extern "C" int mh_rope_kv_append(void* qkv, long ld, int n_heads, int head_dim, const int* pos, const float* cos_tab,
                                 const float* sin_tab, void* cache, long cache_bstride, long ld_cache,
                                 const int* pos_dev, int B, hipStream_t stream) {
  if (B <= 0) return MH_OK;
  if (head_dim % 8 || ld % 8 || ld_cache % 8 || cache_bstride % 8) return MH_ERR_ARG;
  const long items = (long)B * (2L * n_heads * (head_dim / 8) + (long)n_heads * head_dim / 8);
  hipLaunchKernelGGL(rope_kv_append_kernel, dim3(ew_grid(items)), dim3(EW_NT), 0, stream, (bf16_t*)qkv, ld, n_heads,
                     head_dim, pos, cos_tab, sin_tab, (bf16_t*)cache, cache_bstride, ld_cache, pos_dev, B);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

__global__ void add_i32_kernel(int* x, int n, int delta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += delta;
}
extern "C" int mh_add_i32(int* x, int n, int delta, hipStream_t stream) {
  if (n <= 0) return MH_OK;
  hipLaunchKernelGGL(add_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x, n, delta);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// debug (tests): the hardware fp32 -> bf16 rounding next to the integer form it replaced (common.h), element by element
__global__ void bf16_round_check_kernel(const float* __restrict__ x, bf16_t* __restrict__ hw, bf16_t* __restrict__ hw_pk,
                                        bf16_t* __restrict__ sw, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const float a = x[2 * i], b = x[2 * i + 1];
  hw[2 * i] = f2bf(a);
  hw[2 * i + 1] = f2bf(b);
  const unsigned pk = pack_bf2(a, b);
  hw_pk[2 * i] = (bf16_t)(pk & 0xffffu);
  hw_pk[2 * i + 1] = (bf16_t)(pk >> 16);
  sw[2 * i] = f2bf_sw(a);
  sw[2 * i + 1] = f2bf_sw(b);
}
extern "C" int mh_bf16_round_check(const float* x, void* hw, void* hw_pk, void* sw, long n, hipStream_t stream) {
  if (n <= 0 || (n & 1)) return MH_ERR_ARG;
  hipLaunchKernelGGL(bf16_round_check_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, stream, x, (bf16_t*)hw,
                     (bf16_t*)hw_pk, (bf16_t*)sw, n);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

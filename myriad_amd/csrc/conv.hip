// K12: the vision-expert conv stacks (reference networks.py:98-127 VEInstructorV2, :159-189 VETokenizer)
// as im2col + the MFMA GEMM.  Activations are NHWC bf16, so an im2col row is kh*kw contiguous C-chunks.
// The bias rides along as one extra K column (im2col writes 1.0 there, the packed weight holds the bias),
// so the forward GEMM needs no bias epilogue and the wgrad GEMM produces the bias gradient for free.
// Master weights are fp32 in GEMM order [Cout, K], k = (ky*kw+kx)*Cin + ci; the checkpoint boundary
// permutes to/from the reference's [Cout,Cin,kh,kw].  pack/unpack add/strip the bias column + K padding.
#include "common.h"

#define CV_NT 256
static inline int cv_grid(long n) {
  long g = (n + CV_NT - 1) / CV_NT;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

__global__ void im2col_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ col, int B, int H, int W, int C,
                              int kh, int kw, int pad, int OH, int OW, int K, int Kpad) {
  // one item = one (row m, 4-element group of the Kpad columns)
  const int groups = Kpad >> 2;
  const long total = (long)B * OH * OW * groups;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long m = it / groups;
    const int k0 = (int)(it - m * groups) * 4;
    const int ox = (int)(m % OW);
    const int oy = (int)((m / OW) % OH);
    const int b = (int)(m / ((long)OW * OH));
    bf16_t v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + e;
      bf16_t val = 0;
      if (k < K) {
        const int tap = k / C, c = k - tap * C;
        const int ky = tap / kw, kx = tap - ky * kw;
        const int iy = oy + ky - pad, ix = ox + kx - pad;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) val = x[(((long)b * H + iy) * W + ix) * C + c];
      } else if (k == K) {
        val = 0x3F80;  // 1.0 -> bias column
      }
      v[e] = val;
    }
    uint2 pk;
    pk.x = (unsigned)v[0] | ((unsigned)v[1] << 16);
    pk.y = (unsigned)v[2] | ((unsigned)v[3] << 16);
    *reinterpret_cast<uint2*>(col + m * Kpad + k0) = pk;
  }
}

extern "C" int mh_im2col_nhwc(const void* x, void* col, int B, int H, int W, int C, int kh, int kw, int pad, int Kpad,
                              hipStream_t stream) {
  const int OH = H + 2 * pad - kh + 1, OW = W + 2 * pad - kw + 1, K = kh * kw * C;
  if (Kpad % 4 || Kpad < K || OH <= 0 || OW <= 0) return MH_ERR_ARG;   // Kpad == K: no bias (ones) column
  const long total = (long)B * OH * OW * (Kpad / 4);
  hipLaunchKernelGGL(im2col_kernel, dim3(cv_grid(total)), dim3(CV_NT), 0, stream, (const bf16_t*)x, (bf16_t*)col, B, H,
                     W, C, kh, kw, pad, OH, OW, K, Kpad);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// gather-form col2im: dx[b,y,x,c] = sum_taps dcol[(b, y+pad-ky, x+pad-kx)][tap*C + c]
__global__ void col2im_kernel(const bf16_t* __restrict__ dcol, float* __restrict__ dx, int B, int H, int W, int C,
                              int kh, int kw, int pad, int OH, int OW, int Kpad) {
  const long total = (long)B * H * W * C;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const int c = (int)(it % C);
    const long p = it / C;
    const int ix = (int)(p % W);
    const int iy = (int)((p / W) % H);
    const int b = (int)(p / ((long)W * H));
    float s = 0.f;
    for (int ky = 0; ky < kh; ++ky) {
      const int oy = iy + pad - ky;
      if (oy < 0 || oy >= OH) continue;
      for (int kx = 0; kx < kw; ++kx) {
        const int ox = ix + pad - kx;
        if (ox < 0 || ox >= OW) continue;
        s += bf2f(dcol[(((long)b * OH + oy) * OW + ox) * Kpad + (ky * kw + kx) * C + c]);
      }
    }
    dx[it] = s;
  }
}

extern "C" int mh_col2im_nhwc(const void* dcol, float* dx, int B, int H, int W, int C, int kh, int kw, int pad,
                              int Kpad, hipStream_t stream) {
  const int OH = H + 2 * pad - kh + 1, OW = W + 2 * pad - kw + 1;
  if (OH <= 0 || OW <= 0) return MH_ERR_ARG;
  hipLaunchKernelGGL(col2im_kernel, dim3(cv_grid((long)B * H * W * C)), dim3(CV_NT), 0, stream, (const bf16_t*)dcol,
                     dx, B, H, W, C, kh, kw, pad, OH, OW, Kpad);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ReLU -> MaxPool2d(2) on NHWC (reference networks.py:100-102 etc.).  The conv pre-activation y is kept in
// fp32 (bf16 would create arg-max ties inside the 2x2 windows and route gradients differently from the fp32
// reference); the pooled output feeding the next MFMA GEMM is bf16.  ldy = row stride of y in elements.
template <typename T>
__device__ __forceinline__ float ldf(const T* p, long i);
template <>
__device__ __forceinline__ float ldf<float>(const float* p, long i) { return p[i]; }
template <>
__device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p, long i) { return bf2f(p[i]); }

template <typename T>
__global__ void relu_pool_fwd_kernel(const T* __restrict__ y, long ldy, bf16_t* __restrict__ p, int B, int H,
                                     int W, int C) {
  const int PH = H >> 1, PW = W >> 1;
  const long total = (long)B * PH * PW * C;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const int c = (int)(it % C);
    const long q = it / C;
    const int px = (int)(q % PW);
    const int py = (int)((q / PW) % PH);
    const int b = (int)(q / ((long)PW * PH));
    const long base = ((long)b * H + 2 * py) * W + 2 * px;
    float m = 0.f;  // relu floor
    m = fmaxf(m, ldf<T>(y, base * ldy + c));
    m = fmaxf(m, ldf<T>(y, (base + 1) * ldy + c));
    m = fmaxf(m, ldf<T>(y, (base + W) * ldy + c));
    m = fmaxf(m, ldf<T>(y, (base + W + 1) * ldy + c));
    p[it] = f2bf(m);
  }
}

// gradient goes to the first arg-max of the window (scan order, like torch) iff that max is > 0
template <typename T>
__global__ void relu_pool_bwd_kernel(const float* __restrict__ dp, const T* __restrict__ y, long ldy,
                                     bf16_t* __restrict__ dy, long lddy, int B, int H, int W, int C) {
  const int PH = H >> 1, PW = W >> 1;
  const long total = (long)B * PH * PW * C;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const int c = (int)(it % C);
    const long q = it / C;
    const int px = (int)(q % PW);
    const int py = (int)((q / PW) % PH);
    const int b = (int)(q / ((long)PW * PH));
    const long base = ((long)b * H + 2 * py) * W + 2 * px;
    const long idx[4] = {base, base + 1, base + W, base + W + 1};
    float best = ldf<T>(y, idx[0] * ldy + c);
    int arg = 0;
#pragma unroll
    for (int e = 1; e < 4; ++e) {
      const float v = ldf<T>(y, idx[e] * ldy + c);
      if (v > best) { best = v; arg = e; }
    }
    const float g = best > 0.f ? dp[it] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) dy[idx[e] * lddy + c] = (e == arg) ? f2bf(g) : (bf16_t)0;
  }
}

extern "C" int mh_relu_maxpool2_fwd(const void* y, int y_is_f32, long ldy, void* p, int B, int H, int W, int C,
                                    hipStream_t stream) {
  if ((H & 1) || (W & 1)) return MH_ERR_ARG;
  const dim3 grid(cv_grid((long)B * (H / 2) * (W / 2) * C)), block(CV_NT);
  if (y_is_f32)
    hipLaunchKernelGGL(relu_pool_fwd_kernel<float>, grid, block, 0, stream, (const float*)y, ldy, (bf16_t*)p, B, H, W, C);
  else
    hipLaunchKernelGGL(relu_pool_fwd_kernel<bf16_t>, grid, block, 0, stream, (const bf16_t*)y, ldy, (bf16_t*)p, B, H, W, C);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
extern "C" int mh_relu_maxpool2_bwd(const float* dp, const void* y, int y_is_f32, long ldy, void* dy, long lddy, int B,
                                    int H, int W, int C, hipStream_t stream) {
  if ((H & 1) || (W & 1)) return MH_ERR_ARG;
  const dim3 grid(cv_grid((long)B * (H / 2) * (W / 2) * C)), block(CV_NT);
  if (y_is_f32)
    hipLaunchKernelGGL(relu_pool_bwd_kernel<float>, grid, block, 0, stream, dp, (const float*)y, ldy, (bf16_t*)dy, lddy, B,
                       H, W, C);
  else
    hipLaunchKernelGGL(relu_pool_bwd_kernel<bf16_t>, grid, block, 0, stream, dp, (const bf16_t*)y, ldy, (bf16_t*)dy, lddy,
                       B, H, W, C);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// Master conv weights are held in GEMM order  W[co][(ky*kw+kx)*Cin + ci]  (fp32; the checkpoint
// boundary permutes to/from the reference's [Cout,Cin,kh,kw] -- myriad_amd/networks.py).
// pack: Wp[co][k] = bf16(W[co][k]) k<K ; Wp[co][K] = bias[co] ; rest 0
__global__ void conv_pack_kernel(const float* __restrict__ Wt, const float* __restrict__ bias, bf16_t* __restrict__ Wp,
                                 int Cout, int K, int Kpad) {
  // one thread per 8 packed elements (Kpad % 64 == 0): a 16-byte store, and two 16-byte loads where the row allows it
  const int cpr = Kpad >> 3;
  const long total = (long)Cout * cpr;
  const bool vec = (K & 3) == 0;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const int co = (int)(it / cpr), k0 = (int)(it - (long)co * cpr) * 8;
    float v[8];
    if (vec && k0 + 8 <= K) {
      const float4_t a = *reinterpret_cast<const float4_t*>(Wt + (long)co * K + k0);
      const float4_t b = *reinterpret_cast<const float4_t*>(Wt + (long)co * K + k0 + 4);
      v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        v[e] = k < K ? Wt[(long)co * K + k] : (k == K && bias ? bias[co] : 0.f);
      }
    }
    uint4 pk;
    pk.x = pack_bf2(v[0], v[1]); pk.y = pack_bf2(v[2], v[3]); pk.z = pack_bf2(v[4], v[5]); pk.w = pack_bf2(v[6], v[7]);
    *reinterpret_cast<uint4*>(Wp + (long)co * Kpad + k0) = pk;
  }
}
// unpack gradients: dW[co][k] = dWp[co][k], db[co] = dWp[co][K]
__global__ void conv_unpack_kernel(const float* __restrict__ dWp, float* __restrict__ dW, float* __restrict__ db,
                                   int Cout, int K, int Kpad) {
  if ((K & 3) == 0) {                               // 16-byte copies; the bias column is element K of each packed row
    const int qpr = K >> 2;
    const long total = (long)Cout * (qpr + 1);
    for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
      const int co = (int)(it / (qpr + 1)), q = (int)(it - (long)co * (qpr + 1));
      if (q == qpr) {
        if (db) db[co] = dWp[(long)co * Kpad + K];
      } else {
        *reinterpret_cast<float4_t*>(dW + (long)co * K + q * 4) = *reinterpret_cast<const float4_t*>(dWp + (long)co * Kpad + q * 4);
      }
    }
    return;
  }
  const long total = (long)Cout * (K + 1);
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const int co = (int)(it / (K + 1)), j = (int)(it - (long)co * (K + 1));
    if (j == K) {
      if (db) db[co] = dWp[(long)co * Kpad + K];
    } else {
      dW[(long)co * K + j] = dWp[(long)co * Kpad + j];
    }
  }
}

extern "C" int mh_conv_pack_weight(const float* W, const float* bias, void* Wp, int Cout, int K, int Kpad,
                                   hipStream_t stream) {
  if (Kpad < K + 1 || (Kpad & 7) != 0) return MH_ERR_ARG;
  hipLaunchKernelGGL(conv_pack_kernel, dim3(cv_grid((long)Cout * (Kpad / 8))), dim3(CV_NT), 0, stream, W, bias,
                     (bf16_t*)Wp, Cout, K, Kpad);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
extern "C" int mh_conv_unpack_grad(const float* dWp, float* dW, float* db, int Cout, int K, int Kpad,
                                   hipStream_t stream) {
  if (Kpad < K + 1) return MH_ERR_ARG;
  hipLaunchKernelGGL(conv_unpack_kernel, dim3(cv_grid((K & 3) ? (long)Cout * (K + 1) : (long)Cout * (K / 4 + 1))), dim3(CV_NT), 0, stream, dWp, dW, db,
                     Cout, K, Kpad);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// K16: anomaly-map heads of the vision expert (reference adrefexpert_v2.py:243-301) -- small HBM-bound kernels around
// the MFMA GEMMs that do the heavy lifting (per-tap decoder Linear, query x reference similarity matrix).
//   l2norm_rows      y = x / max(||x||, eps)                       (F.cosine_similarity / `x / x.norm()` operands)
//   pair_logits      [scale * <p,t0>/||p||, scale * <p,t1>/||p||]   zero-shot logits against the [normal, abnormal] text pair
//   zs_accumulate    mask += softmax(pair)[1] at h x h ; map += softmax(bilinear_ac(pair))[1] at S x S   (one tap)
//   rowmax_skip      acc[row] += w * max over columns that are not class-token columns      (one-shot: best reference patch)
//   bilinear_ac      align_corners=True bilinear resize, optionally 1 - x                     (one-shot map)
// fp32 arithmetic throughout; a two-way softmax is evaluated as sigmoid(l1 - l0), and interpolating the two logit planes
// then taking the softmax equals the sigmoid of the interpolated difference (interpolation is linear).
#include "common.h"

#define EX_NT 256

__global__ __launch_bounds__(EX_NT) void l2norm_rows_kernel(const float* __restrict__ x, long ldx, bf16_t* __restrict__ yb,
                                                            float* __restrict__ yf, long ldy, int D, float eps) {
  __shared__ float red[EX_NT / 64];
  const size_t row = blockIdx.x;
  const float* xr = x + row * ldx;
  float ss = 0.f;
  for (int i = threadIdx.x * 4; i < D; i += EX_NT * 4) {
    const float4_t v = *reinterpret_cast<const float4_t*>(xr + i);
    ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  ss = block_sum<EX_NT / 64>(ss, red);
  const float inv = 1.f / fmaxf(sqrtf(ss), eps);
  for (int i = threadIdx.x * 4; i < D; i += EX_NT * 4) {
    const float4_t v = *reinterpret_cast<const float4_t*>(xr + i);
    if (yb) {
      uint2 pk;
      pk.x = pack_bf2(v[0] * inv, v[1] * inv);
      pk.y = pack_bf2(v[2] * inv, v[3] * inv);
      *reinterpret_cast<uint2*>(yb + row * ldy + i) = pk;
    }
    if (yf) *reinterpret_cast<float4_t*>(yf + row * ldy + i) = (float4_t){v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv};
  }
}

// one wave per token row; rows_per_batch rows share one text pair
__global__ __launch_bounds__(EX_NT) void pair_logits_kernel(const float* __restrict__ p, long ldp, const float* __restrict__ text,
                                                            float* __restrict__ out, long rows, int rows_per_batch, int C,
                                                            float scale) {
  const long row = (long)blockIdx.x * (EX_NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* pr = p + row * ldp;
  const float* t0 = text + (row / rows_per_batch) * 2L * C;
  const float* t1 = t0 + C;
  float d0 = 0.f, d1 = 0.f, nn = 0.f;
  for (int i = lane * 4; i < C; i += 256) {
    const float4_t v = *reinterpret_cast<const float4_t*>(pr + i);
    const float4_t a = *reinterpret_cast<const float4_t*>(t0 + i);
    const float4_t b = *reinterpret_cast<const float4_t*>(t1 + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      d0 += v[e] * a[e];
      d1 += v[e] * b[e];
      nn += v[e] * v[e];
    }
  }
  d0 = wave_sum(d0);
  d1 = wave_sum(d1);
  nn = wave_sum(nn);
  if (lane == 0) {
    const float inv = scale / sqrtf(nn);
    out[row * 2] = d0 * inv;
    out[row * 2 + 1] = d1 * inv;
  }
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + __expf(-x)); }

// logits [B, h*h, 2]; mask_acc [B, h, h] += w * sigmoid(l1 - l0); map_acc [B, S, S] += w * sigmoid(bilinear_ac(l1 - l0))
__global__ void zs_accumulate_kernel(const float* __restrict__ logits, float* __restrict__ mask_acc, float* __restrict__ map_acc,
                                     int B, int h, int S, float w) {
  const long total = (long)B * S * S;
  const float step = S > 1 ? (float)(h - 1) / (float)(S - 1) : 0.f;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const int b = (int)(it / ((long)S * S));
    const int rem = (int)(it - (long)b * S * S);
    const int oy = rem / S, ox = rem - oy * S;
    const float fy = oy * step, fx = ox * step;
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 < h - 1 ? y0 : (h > 1 ? h - 2 : 0);
    x0 = x0 < h - 1 ? x0 : (h > 1 ? h - 2 : 0);
    const int y1 = y0 + (h > 1), x1 = x0 + (h > 1);
    const float wy = fy - y0, wx = fx - x0;
    const float* lb = logits + (long)b * h * h * 2;
    auto diff = [&](int y, int x) { return lb[(y * h + x) * 2 + 1] - lb[(y * h + x) * 2]; };
    const float d = (1.f - wy) * ((1.f - wx) * diff(y0, x0) + wx * diff(y0, x1)) + wy * ((1.f - wx) * diff(y1, x0) + wx * diff(y1, x1));
    map_acc[it] += w * sigmoidf(d);
    if (oy < h && ox < h) mask_acc[((long)b * h + oy) * h + ox] += w * sigmoidf(diff(oy, ox));
  }
}

// acc[row] += w * max_{c < cols, c % period != 0} s[row][c]   (one wave per row)
__global__ __launch_bounds__(EX_NT) void rowmax_skip_kernel(const float* __restrict__ s, long lds, float* __restrict__ acc, long rows,
                                                            int cols, int period, float w) {
  const long row = (long)blockIdx.x * (EX_NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64)
    if (period <= 0 || (c % period) != 0) m = fmaxf(m, s[row * lds + c]);
  m = wave_max(m);
  if (lane == 0) acc[row] += w * m;
}

// out[b, oy, ox] = (one_minus ? 1 - v : v), v = bilinear(in[b]) with align_corners=True
__global__ void bilinear_ac_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int h, int w_in, int H, int W,
                                   int one_minus) {
  const long total = (long)B * H * W;
  const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w_in - 1) / (float)(W - 1) : 0.f;
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const int b = (int)(it / ((long)H * W));
    const int rem = (int)(it - (long)b * H * W);
    const int oy = rem / W, ox = rem - oy * W;
    const float fy = oy * sy, fx = ox * sx;
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 < h - 1 ? y0 : (h > 1 ? h - 2 : 0);
    x0 = x0 < w_in - 1 ? x0 : (w_in > 1 ? w_in - 2 : 0);
    const int y1 = y0 + (h > 1), x1 = x0 + (w_in > 1);
    const float wy = fy - y0, wx = fx - x0;
    const float* ib = in + (long)b * h * w_in;
    const float v = (1.f - wy) * ((1.f - wx) * ib[y0 * w_in + x0] + wx * ib[y0 * w_in + x1]) +
                    wy * ((1.f - wx) * ib[y1 * w_in + x0] + wx * ib[y1 * w_in + x1]);
    out[it] = one_minus ? 1.f - v : v;
  }
}

static inline int ex_grid(long n) {
  long g = (n + EX_NT - 1) / EX_NT;
  if (g > 4096) g = 4096;
  return (int)(g < 1 ? 1 : g);
}

extern "C" int mh_l2norm_rows(const float* x, long ldx, void* y_bf16, float* y_f32, long ldy, int M, int D, float eps,
                              hipStream_t stream) {
  if (M <= 0) return MH_OK;
  if ((D % 4) != 0 || (ldx % 4) != 0 || (ldy % 4) != 0) return MH_ERR_ARG;
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3(M), dim3(EX_NT), 0, stream, x, ldx, (bf16_t*)y_bf16, y_f32, ldy, D, eps);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_pair_logits(const float* p, long ldp, const float* text, float* out, long rows, int rows_per_batch, int C,
                              float scale, hipStream_t stream) {
  if (rows <= 0) return MH_OK;
  if ((C % 4) != 0 || (ldp % 4) != 0 || rows_per_batch <= 0) return MH_ERR_ARG;
  hipLaunchKernelGGL(pair_logits_kernel, dim3((int)((rows + 3) / 4)), dim3(EX_NT), 0, stream, p, ldp, text, out, rows,
                     rows_per_batch, C, scale);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_zs_accumulate(const float* logits, float* mask_acc, float* map_acc, int B, int h, int S, float w,
                                hipStream_t stream) {
  if (B <= 0) return MH_OK;
  if (h <= 0 || S < h) return MH_ERR_ARG;
  hipLaunchKernelGGL(zs_accumulate_kernel, dim3(ex_grid((long)B * S * S)), dim3(EX_NT), 0, stream, logits, mask_acc, map_acc,
                     B, h, S, w);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_rowmax_skip(const float* s, long lds, float* acc, long rows, int cols, int period, float w,
                              hipStream_t stream) {
  if (rows <= 0) return MH_OK;
  if (cols <= 0) return MH_ERR_ARG;
  hipLaunchKernelGGL(rowmax_skip_kernel, dim3((int)((rows + 3) / 4)), dim3(EX_NT), 0, stream, s, lds, acc, rows, cols, period,
                     w);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_bilinear_ac(const float* in, float* out, int B, int h, int w, int H, int W, int one_minus,
                              hipStream_t stream) {
  if (B <= 0) return MH_OK;
  if (h <= 0 || w <= 0 || H <= 0 || W <= 0) return MH_ERR_ARG;
  hipLaunchKernelGGL(bilinear_ac_kernel, dim3(ex_grid((long)B * H * W)), dim3(EX_NT), 0, stream, in, out, B, h, w, H, W,
                     one_minus);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

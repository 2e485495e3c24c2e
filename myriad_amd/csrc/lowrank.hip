// K10: rank-r residual adaptor  y = x + (x A^T) B^T  (reference networks.py:81-93 LoraAdaptorV2, r=4, D=1408),
// forward + dgrad + wgrad(A,B), all fp32 (the adaptor sits on the fp32 image-token stream in front of ln_vision).
// Wavefront-primitive kernels: the r dot products per row are wave/block reductions, the rank-r update is a
// broadcast FMA; wgrad reduces over rows in fixed-order partial slabs (deterministic, no atomics).
//   A: [R, D] (conv1.weight), Bm: [D, R] (conv2.weight), R in {1,2,4,8}.
#include "common.h"

#define LR_NT 256
#define LR_NW 4
#define LR_CHUNKS 16

template <int R>
__global__ __launch_bounds__(LR_NT) void lowrank_fwd_kernel(const float* __restrict__ x, const float* __restrict__ A,
                                                            const float* __restrict__ Bm, float* __restrict__ y,
                                                            float* __restrict__ t_out, int D) {
  __shared__ float red[LR_NW];
  __shared__ float ts[R];
  const long row = blockIdx.x;
  const float* xr = x + row * D;
  float part[R];
#pragma unroll
  for (int r = 0; r < R; ++r) part[r] = 0.f;
  for (int k = threadIdx.x; k < D; k += LR_NT) {
    const float xv = xr[k];
#pragma unroll
    for (int r = 0; r < R; ++r) part[r] += xv * A[(long)r * D + k];
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float s = block_sum<LR_NW>(part[r], red);
    if (threadIdx.x == 0) {
      ts[r] = s;
      if (t_out) t_out[row * R + r] = s;
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < D; k += LR_NT) {
    float v = xr[k];
#pragma unroll
    for (int r = 0; r < R; ++r) v += ts[r] * Bm[(long)k * R + r];
    y[row * D + k] = v;
  }
}

// per row: dt[r] = sum_k dy[k] B[k][r] ; optional dx = dy + dt.A
template <int R>
__global__ __launch_bounds__(LR_NT) void lowrank_bwd_row_kernel(const float* __restrict__ dy, const float* __restrict__ A,
                                                                const float* __restrict__ Bm, float* __restrict__ dt_out,
                                                                float* __restrict__ dx, int D) {
  __shared__ float red[LR_NW];
  __shared__ float ts[R];
  const long row = blockIdx.x;
  const float* gr = dy + row * D;
  float part[R];
#pragma unroll
  for (int r = 0; r < R; ++r) part[r] = 0.f;
  for (int k = threadIdx.x; k < D; k += LR_NT) {
    const float g = gr[k];
#pragma unroll
    for (int r = 0; r < R; ++r) part[r] += g * Bm[(long)k * R + r];
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float s = block_sum<LR_NW>(part[r], red);
    if (threadIdx.x == 0) {
      ts[r] = s;
      dt_out[row * R + r] = s;
    }
  }
  if (dx) {
    __syncthreads();
    for (int k = threadIdx.x; k < D; k += LR_NT) {
      float v = gr[k];
#pragma unroll
      for (int r = 0; r < R; ++r) v += ts[r] * A[(long)r * D + k];
      dx[row * D + k] = v;
    }
  }
}

// partial wgrad over a row chunk: pA[chunk][r][k] = sum_m dt[m][r] x[m][k]; pB[chunk][k][r] = sum_m dy[m][k] t[m][r]
template <int R>
__global__ __launch_bounds__(LR_NT) void lowrank_wgrad_partial_kernel(const float* __restrict__ dy,
                                                                      const float* __restrict__ x,
                                                                      const float* __restrict__ t,
                                                                      const float* __restrict__ dt, float* __restrict__ pA,
                                                                      float* __restrict__ pB, int M, int D) {
  __shared__ float sa[4][R][64];
  __shared__ float sb[4][R][64];
  const int k = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ty = threadIdx.x >> 6;
  const int chunk = blockIdx.y;
  const int rows_per = (M + LR_CHUNKS - 1) / LR_CHUNKS;
  const int m0 = chunk * rows_per;
  const int m1 = (m0 + rows_per) < M ? (m0 + rows_per) : M;
  float a[R], b[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    a[r] = 0.f;
    b[r] = 0.f;
  }
  if (k < D) {
    for (int m = m0 + ty; m < m1; m += 4) {
      const float xv = x[(long)m * D + k], gv = dy[(long)m * D + k];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        a[r] += dt[(long)m * R + r] * xv;
        b[r] += gv * t[(long)m * R + r];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    sa[ty][r][threadIdx.x & 63] = a[r];
    sb[ty][r][threadIdx.x & 63] = b[r];
  }
  __syncthreads();
  if (ty == 0 && k < D) {
    const int c = threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      pA[((long)chunk * R + r) * D + k] = sa[0][r][c] + sa[1][r][c] + sa[2][r][c] + sa[3][r][c];
      pB[((long)chunk * D + k) * R + r] = sb[0][r][c] + sb[1][r][c] + sb[2][r][c] + sb[3][r][c];
    }
  }
}
__global__ void lowrank_wgrad_reduce_kernel(const float* __restrict__ pA, const float* __restrict__ pB,
                                            float* __restrict__ dA, float* __restrict__ dB, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = 0.f, b = 0.f;
  for (int c = 0; c < LR_CHUNKS; ++c) {
    a += pA[(long)c * n + i];
    b += pB[(long)c * n + i];
  }
  dA[i] = a;
  dB[i] = b;
}

#define LR_DISPATCH(R_, CALL)                     \
  switch (R_) {                                   \
    case 1: { constexpr int R = 1; CALL; break; } \
    case 2: { constexpr int R = 2; CALL; break; } \
    case 4: { constexpr int R = 4; CALL; break; } \
    case 8: { constexpr int R = 8; CALL; break; } \
    default: return MH_ERR_UNSUPPORTED;           \
  }

extern "C" int mh_lowrank_fwd(const float* x, const float* A, const float* Bm, float* y, float* t, int M, int D, int R_,
                              hipStream_t stream) {
  if (M <= 0) return MH_OK;
  LR_DISPATCH(R_, hipLaunchKernelGGL(lowrank_fwd_kernel<R>, dim3(M), dim3(LR_NT), 0, stream, x, A, Bm, y, t, D));
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// workspace floats needed by mh_lowrank_bwd: M*R (dt) + 2 * LR_CHUNKS * R * D
extern "C" long mh_lowrank_bwd_ws_floats(int M, int D, int R_) { return (long)M * R_ + 2L * LR_CHUNKS * R_ * D; }

extern "C" int mh_lowrank_bwd(const float* dy, const float* x, const float* t, const float* A, const float* Bm,
                              float* dA, float* dB, float* dx, float* ws, int M, int D, int R_, hipStream_t stream) {
  if (M <= 0) return MH_OK;
  float* dt = ws;
  float* pA = ws + (long)M * R_;
  float* pB = pA + (long)LR_CHUNKS * R_ * D;
  LR_DISPATCH(R_, hipLaunchKernelGGL(lowrank_bwd_row_kernel<R>, dim3(M), dim3(LR_NT), 0, stream, dy, A, Bm, dt, dx, D));
  MH_CHECK_LAUNCH();
  LR_DISPATCH(R_, hipLaunchKernelGGL(lowrank_wgrad_partial_kernel<R>, dim3((D + 63) / 64, LR_CHUNKS), dim3(LR_NT), 0,
                                     stream, dy, x, t, dt, pA, pB, M, D));
  MH_CHECK_LAUNCH();
  const int n = R_ * D;
  hipLaunchKernelGGL(lowrank_wgrad_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, pA, pB, dA, dB, n);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// K5 (decode): skinny-M GEMM  C[M<=16, N] = alpha * A[M,K] . B[N,K]^T (+bias) (+residual)  -- weight streaming.
//
// One decode token multiplies 1..16 activation rows by every frozen weight matrix (13.2 GB bf16 per token for
// Vicuna-7B, reference modeling_llama.py:184-231 with the KV cache): HBM-bound, 2.1 ms/token at 6.3 TB/s.  The
// 128x128 training tile starves here (N = 4096 gives 32 workgroups for 256 CUs and nothing hides HBM latency), so
// this kernel makes the WEIGHT stream the only thing that matters:
//   * a workgroup owns 16 output columns (16 weight rows), its 4 waves split K (interleaved 32-deep steps) and are
//     reduced through LDS at the end  ->  N/16 workgroups (768 for qkv, 1376 for gate|up), ~1k waves streaming;
//   * every lane loads its weight fragment straight from global memory in MFMA B-operand layout (16 B/lane, deep
//     unroll, no LDS round trip -- the operand is read once and never shared between waves);
//   * the activation rows (<= 16 x K bf16, L2-resident) are read as the A operand of v_mfma_f32_16x16x32_bf16, so
//     one MFMA retires 1 KiB of weights: the matrix pipe is idle-cheap and exact-fp32 accumulation comes for free.
#include "common.h"

#define GV_NW 4
#define GV_UNROLL 8

__global__ __launch_bounds__(GV_NW * 64) void gemv_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                          void* __restrict__ Cv, const float* __restrict__ bias,
                                                          const float* res, int M, int N, int K, int lda, int ldb,
                                                          int ldc, int ldr, int out_f32, float alpha) {
  __shared__ float red[GV_NW][16 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.x * 16;
  int nrow = n0 + lr;
  nrow = nrow < N ? nrow : N - 1;
  const int mrow = lr < M ? lr : M - 1;          // rows >= M duplicate the last row; their results are never stored
  const bf16_t* wp = B + (size_t)nrow * ldb + lg * 8;
  const bf16_t* xp = A + (size_t)mrow * lda + lg * 8;
  float4_t acc = (float4_t){0.f, 0.f, 0.f, 0.f};
  const int nsteps = K / 32;
  // wave w takes steps w, w+4, w+8, ...; GV_UNROLL independent loads in flight per lane
  int s = wave;
  for (; s + (GV_UNROLL - 1) * GV_NW < nsteps; s += GV_UNROLL * GV_NW) {
    short8_t wv[GV_UNROLL], xv[GV_UNROLL];
#pragma unroll
    for (int u = 0; u < GV_UNROLL; ++u) {
      const int k = (s + u * GV_NW) * 32;
      wv[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + k));   // streamed once: don't pollute L2
      xv[u] = *reinterpret_cast<const short8_t*>(xp + k);
    }
#pragma unroll
    for (int u = 0; u < GV_UNROLL; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xv[u], wv[u], acc, 0, 0, 0);
  }
  for (; s < nsteps; s += GV_NW) {
    const int k = s * 32;
    const short8_t wv = *reinterpret_cast<const short8_t*>(wp + k);
    const short8_t xv = *reinterpret_cast<const short8_t*>(xp + k);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xv, wv, acc, 0, 0, 0);
  }
  // D layout: row m = 4*lg + r, col n = lr.  Cross-wave K reduction through LDS, then wave 0 finishes.
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][(4 * lg + r) * 16 + lr] = acc[r];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 4 * lg + r, n = n0 + lr;
      if (m < M && n < N) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < GV_NW; ++w) v += red[w][m * 16 + lr];
        v *= alpha;
        if (bias) v += bias[n];
        if (res) v += res[(size_t)m * ldr + n];
        if (out_f32) reinterpret_cast<float*>(Cv)[(size_t)m * ldc + n] = v;
        else reinterpret_cast<bf16_t*>(Cv)[(size_t)m * ldc + n] = f2bf(v);
      }
    }
  }
}

// called from mh_gemm_bf16_nt for M <= 16 (no GELU epilogue on this path)
int mh_launch_gemv(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                   const float* bias, const float* residual, int ldr, int out_f32, float alpha, hipStream_t stream) {
  hipLaunchKernelGGL(gemv_kernel, dim3((N + 15) / 16), dim3(GV_NW * 64), 0, stream, (const bf16_t*)A, (const bf16_t*)B,
                     C, bias, residual, M, N, K, lda, ldb, ldc, ldr, out_f32, alpha);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

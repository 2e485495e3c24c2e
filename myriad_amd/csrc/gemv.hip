// K5 (decode): skinny-M GEMM  C[M<=16, N] = alpha * A[M,K] . B[N,K]^T (+bias) (+residual)  -- weight streaming.
//
// One decode token multiplies 1..16 activation rows by every frozen weight matrix (13.2 GB bf16 per token for
// Vicuna-7B, reference modeling_llama.py:184-231 with the KV cache): HBM-bound, 2.1 ms/token at 6.3 TB/s.  The
// 128x128 training tile starves here (N = 4096 gives 32 workgroups for 256 CUs and nothing hides HBM latency), so
// this kernel makes the WEIGHT stream the only thing that matters:
//   * a workgroup owns 16 output columns (16 weight rows), its 4 waves split K and are reduced through LDS at the
//     end  ->  N/16 workgroups (768 for qkv, 1376 for gate|up), ~1k waves streaming;
//   * every lane loads its weight fragment straight from global memory in MFMA B-operand layout (no LDS round
//     trip -- the operand is read once and never shared between waves).  The dot product does not care which k a
//     lane holds as long as both operands agree, so lane (row, lg) takes 32 CONTIGUOUS bytes (k = 16*lg .. +15 of a
//     64-deep step, two MFMAs): the four lanes of a row then cover one whole 128-B line per step instead of half of
//     one (MODE 0, the first version, measured 3.9-4.2 TB/s);
//   * the activation rows (<= 16 x K bf16, L2-resident) are read as the A operand of v_mfma_f32_16x16x32_bf16, so
//     one MFMA retires 1 KiB of weights: the matrix pipe is idle-cheap and exact-fp32 accumulation comes for free.
#include "common.h"
#include <cstdlib>

template <int MODE, int UNROLL, int GV_NW>
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
  float4_t acc = (float4_t){0.f, 0.f, 0.f, 0.f};
  if (MODE == 0) {
    const bf16_t* wp = B + (size_t)nrow * ldb + lg * 8;
    const bf16_t* xp = A + (size_t)mrow * lda + lg * 8;
    const int nsteps = K / 32;
    // wave w takes 32-deep steps w, w+4, w+8, ...; UNROLL independent loads in flight per lane
    int s = wave;
    for (; s + (UNROLL - 1) * GV_NW < nsteps; s += UNROLL * GV_NW) {
      short8_t wv[UNROLL], xv[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int k = (s + u * GV_NW) * 32;
        wv[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + k));   // streamed once
        xv[u] = *reinterpret_cast<const short8_t*>(xp + k);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xv[u], wv[u], acc, 0, 0, 0);
    }
    for (; s < nsteps; s += GV_NW) {
      const int k = s * 32;
      const short8_t wv = *reinterpret_cast<const short8_t*>(wp + k);
      const short8_t xv = *reinterpret_cast<const short8_t*>(xp + k);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xv, wv, acc, 0, 0, 0);
    }
  } else if (MODE == 2) {
    // MODE 1's arithmetic on a pre-permuted weight copy (mh_gemv_pack): the bytes lane (lr, lg) of wave w needs at step t
    // sit at ((block * NW + w) * per + t) * 2 KiB + h * 1 KiB + lane * 16, so every wave-instruction reads one contiguous
    // KiB and a wave walks one contiguous region -- 6.8 TB/s against 5.8 TB/s for the row-strided order on a pure stream
    // (tools/micro/stream_pattern.hip), and bit-identical results (same k per lane, same reduction order).
    const bf16_t* xp = A + (size_t)mrow * lda + lg * 16;
    const int nsteps = K / 64;
    const int per = (nsteps + GV_NW - 1) / GV_NW;
    const bf16_t* wp = B + ((size_t)blockIdx.x * GV_NW + wave) * per * 1024 + lane * 8;
    int s = wave * per;
    const int s0 = s;
    const int s_end = (s + per) < nsteps ? (s + per) : nsteps;
    for (; s + UNROLL <= s_end; s += UNROLL) {
      short8_t w0[UNROLL], w1[UNROLL], x0[UNROLL], x1[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int k = (s + u) * 64;
        w0[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + (size_t)(s - s0 + u) * 1024));
        w1[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + (size_t)(s - s0 + u) * 1024 + 512));
        x0[u] = *reinterpret_cast<const short8_t*>(xp + k);
        x1[u] = *reinterpret_cast<const short8_t*>(xp + k + 8);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0[u], w0[u], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1[u], w1[u], acc, 0, 0, 0);
      }
    }
    if (s < s_end) {                                   // remainder as one partial batch (loads in flight together), same order
      const int rem = s_end - s;
      short8_t w0[UNROLL], w1[UNROLL], x0[UNROLL], x1[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        if (u < rem) {
          const int k = (s + u) * 64;
          w0[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + (size_t)(s - s0 + u) * 1024));
          w1[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + (size_t)(s - s0 + u) * 1024 + 512));
          x0[u] = *reinterpret_cast<const short8_t*>(xp + k);
          x1[u] = *reinterpret_cast<const short8_t*>(xp + k + 8);
        }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        if (u < rem) {
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0[u], w0[u], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1[u], w1[u], acc, 0, 0, 0);
        }
    }
  } else {
    // 64-deep steps, lane holds k = 16*lg .. 16*lg+15 (32 contiguous bytes); wave w owns the contiguous K quarter
    // [w*K/4, (w+1)*K/4) rounded to steps, so each wave walks 16 rows line by line
    const bf16_t* wp = B + (size_t)nrow * ldb + lg * 16;
    const bf16_t* xp = A + (size_t)mrow * lda + lg * 16;
    const int nsteps = K / 64;
    const int per = (nsteps + GV_NW - 1) / GV_NW;
    int s = wave * per;
    const int s_end = (s + per) < nsteps ? (s + per) : nsteps;
    for (; s + UNROLL <= s_end; s += UNROLL) {
      short8_t w0[UNROLL], w1[UNROLL], x0[UNROLL], x1[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int k = (s + u) * 64;
        w0[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + k));
        w1[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + k + 8));
        x0[u] = *reinterpret_cast<const short8_t*>(xp + k);
        x1[u] = *reinterpret_cast<const short8_t*>(xp + k + 8);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0[u], w0[u], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1[u], w1[u], acc, 0, 0, 0);
      }
    }
    for (; s < s_end; ++s) {
      const int k = s * 64;
      const short8_t w0 = *reinterpret_cast<const short8_t*>(wp + k), w1 = *reinterpret_cast<const short8_t*>(wp + k + 8);
      const short8_t x0 = *reinterpret_cast<const short8_t*>(xp + k), x1 = *reinterpret_cast<const short8_t*>(xp + k + 8);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0, w0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, w1, acc, 0, 0, 0);
    }
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
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("MYRIAD_GEMV_MODE");   // debug: A/B of the load patterns
    mode = e ? atoi(e) : 1;
  }
  static int nw_small = -1;
  if (nw_small < 0) {
    const char* e = getenv("MYRIAD_GEMV_NW");     // debug: waves per workgroup when N/16 under-fills the chip
    nw_small = e ? atoi(e) : 8;                   // N = 4096, K = 11008: 4.12 vs 3.86 TB/s with 8 waves
  }
  const dim3 grid((N + 15) / 16);
#define GV_LAUNCH(MODE, UNR, NW)                                                                                      \
  hipLaunchKernelGGL((gemv_kernel<MODE, UNR, NW>), grid, dim3(NW * 64), 0, stream, (const bf16_t*)A, (const bf16_t*)B, \
                     C, bias, residual, M, N, K, lda, ldb, ldc, ldr, out_f32, alpha)
  const bool small = (N + 15) / 16 < 512;
  if (mode == 0) GV_LAUNCH(0, 8, 4);
  else if (small && nw_small == 8) GV_LAUNCH(1, 8, 8);
  else if (small && nw_small == 16) GV_LAUNCH(1, 4, 16);
  else GV_LAUNCH(1, 8, 4);
#undef GV_LAUNCH
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ---- stream-ordered weight copy for decode (288 GB of HBM: a second, 13.5 GB copy of the frozen LLaMA weights is cheap) ----
static inline int gv_packed_nw(int N) { return ((N + 15) / 16 < 512) ? 8 : 4; }   // the launch rule above, without the env knob

__global__ void gemv_pack_kernel(const bf16_t* __restrict__ W, int ldb, int N, int K, bf16_t* __restrict__ out, int nw, int per,
                                 long chunks) {
  const int nsteps = K / 64;
  for (long c = blockIdx.x * (long)blockDim.x + threadIdx.x; c < chunks; c += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(c & 63), h = (int)((c >> 6) & 1);
    long r = c >> 7;
    const int t = (int)(r % per); r /= per;
    const int q = (int)(r % nw);
    const long nb = r / nw;
    const int lr = lane & 15, lg = lane >> 4;
    long row = nb * 16 + lr;
    row = row < N ? row : N - 1;
    const int step = q * per + t;
    short8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (step < nsteps) v = *reinterpret_cast<const short8_t*>(W + row * ldb + step * 64 + lg * 16 + h * 8);
    *reinterpret_cast<short8_t*>(out + c * 8) = v;
  }
}

extern "C" long mh_gemv_pack_elems(int N, int K) {
  if (N <= 0 || K <= 0 || (K % 64) != 0) return -1;
  const int nw = gv_packed_nw(N), per = (K / 64 + nw - 1) / nw;
  return (long)((N + 15) / 16) * nw * per * 1024;
}

extern "C" int mh_gemv_pack(const void* W, int ldb, int N, int K, void* out, hipStream_t stream) {
  if (N <= 0 || K <= 0 || (K % 64) != 0 || (ldb % 8) != 0 || ((uintptr_t)W & 15) || ((uintptr_t)out & 15)) return MH_ERR_ARG;
  const int nw = gv_packed_nw(N), per = (K / 64 + nw - 1) / nw;
  const long chunks = (long)((N + 15) / 16) * nw * per * 128;
  long grid = (chunks + 255) / 256;
  if (grid > 65536) grid = 65536;
  hipLaunchKernelGGL(gemv_pack_kernel, dim3((int)grid), dim3(256), 0, stream, (const bf16_t*)W, ldb, N, K, (bf16_t*)out, nw, per,
                     chunks);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// C[M <= 16, N] = alpha * A . W^T (+bias) (+residual) with W given as its mh_gemv_pack copy
extern "C" int mh_gemv_packed(const void* A, int lda, const void* P, void* C, int ldc, int M, int N, int K, const float* bias,
                              const float* residual, int ldr, int out_f32, float alpha, hipStream_t stream) {
  if (M <= 0 || N <= 0) return MH_OK;
  if (M > 16 || K <= 0 || (K % 64) != 0 || (lda % 8) != 0 || ((uintptr_t)A & 15) || ((uintptr_t)P & 15)) return MH_ERR_ARG;
  const dim3 grid((N + 15) / 16);
  if (gv_packed_nw(N) == 8)
    hipLaunchKernelGGL((gemv_kernel<2, 8, 8>), grid, dim3(512), 0, stream, (const bf16_t*)A, (const bf16_t*)P, C, bias, residual, M,
                       N, K, lda, 0, ldc, ldr, out_f32, alpha);
  else
    hipLaunchKernelGGL((gemv_kernel<2, 8, 4>), grid, dim3(256), 0, stream, (const bf16_t*)A, (const bf16_t*)P, C, bias, residual, M,
                       N, K, lda, 0, ldc, ldr, out_f32, alpha);
  MH_CHECK_LAUNCH();
  return MH_OK;
}


// ---- decode GEMV with the producer of its activation operand fused in (one launch instead of two per Linear) ---------------
// The single-token step runs ~290 launches of 3-25 us; the RMSNorm in front of the qkv / gate|up / lm_head products and the
// SiLU gate in front of the down projection are per-row elementwise work on <= 16 rows that every workgroup can redo for
// itself: the operand rows are built ONCE per workgroup into LDS (bf16, [M][K]) and the weight stream then runs exactly as
// gemv_kernel<2> (same k per lane, same reduction order: bit-identical to the two-launch form).
//   PRO 1 (SiLU gate, modeling_llama.py:139-140): A = gu [M, 2K] bf16 with gate / up interleaved in blocks of 128
//          (llama.py interleave_gate_up); operand = bf16(silu(g) * u), the expression of silu_mul_fwd_kernel.
//   PRO 2 (RMSNorm, modeling_llama.py:66-74): A = h [M, K] f32; operand = bf16(w * (h * rsqrt(mean(h^2) + eps))): the first
//          256 threads sum the squares in rmsnorm_fwd_kernel's order (thread t: elements 4t + 1024 j) and the block reduction
//          adds the same four wave sums first, so the scale and every operand element carry the same bits.
template <int UNROLL, int GV_NW, int PRO>
__global__ __launch_bounds__(GV_NW * 64) void gemv_pro_kernel(const void* __restrict__ Ain, long lda, const bf16_t* __restrict__ B,
                                                              void* __restrict__ Cv, const float* __restrict__ bias, const float* res,
                                                              int M, int N, int K, int ldc, int ldr, int out_f32, float alpha,
                                                              const float* __restrict__ norm_w, float eps) {
  extern __shared__ __attribute__((aligned(16))) char gsm[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(gsm);                  // [M][K] operand rows
  __shared__ float red[GV_NW][16 * 16];
  __shared__ float bred[GV_NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int nsteps = K / 64;
  const int per = (nsteps + GV_NW - 1) / GV_NW;
  const bf16_t* wp = B + ((size_t)blockIdx.x * GV_NW + wave) * per * 1024 + lane * 8;
  // The weight stream does not depend on the operand: the first UNROLL steps of it are put in flight BEFORE the rows are
  // built (every workgroup of a launch starts at the same time; without this the HBM pipe idles for the ~3 us the prologue
  // takes).  Vector loads retire in order, so what the prologue needs first -- the fp32 row and the norm weights -- is
  // requested ahead of the weights.
  int s = wave * per;
  const int s0 = s;
  const int s_end = (s + per) < nsteps ? (s + per) : nsteps;
  bool have = s + UNROLL <= s_end;
  short8_t w0[UNROLL], w1[UNROLL];
  float4_t hv[4];                                               // PRO 2: K <= 4096 (checked by the launcher)
  if (PRO == 2 && tid < 256) {
    const float* xr = reinterpret_cast<const float*>(Ain);
    int c = 0;
    for (int i = tid * 4; i < K; i += 1024, ++c) hv[c] = *reinterpret_cast<const float4_t*>(xr + i);
  }
  if (have) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      w0[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + (size_t)u * 1024));
      w1[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + (size_t)u * 1024 + 512));
    }
  }
  if (PRO == 1) {
    const bf16_t* gu = reinterpret_cast<const bf16_t*>(Ain);
    const int per_row = K >> 3;
    for (int it = tid; it < M * per_row; it += GV_NW * 64) {
      const int m = it / per_row, c = (it - m * per_row) * 8;
      const long gc = (long)(c >> 7) * 256 + (c & 127);
      const short8_t g = *reinterpret_cast<const short8_t*>(gu + (size_t)m * lda + gc);
      const short8_t u = *reinterpret_cast<const short8_t*>(gu + (size_t)m * lda + gc + 128);
      short8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gv = bf2f((bf16_t)g[e]), uv = bf2f((bf16_t)u[e]);
        o[e] = (short)f2bf(gv / (1.f + __expf(-gv)) * uv);
      }
      *reinterpret_cast<short8_t*>(xs + (size_t)m * K + c) = o;
    }
  } else {
    const float* h = reinterpret_cast<const float*>(Ain);
    for (int m = 0; m < M; ++m) {
      const float* xr = h + (size_t)m * lda;
      float ss = 0.f;
      int c = 0;
      if (tid < 256)
        for (int i = tid * 4; i < K; i += 1024, ++c) {
          if (m > 0) hv[c] = *reinterpret_cast<const float4_t*>(xr + i);
          ss += hv[c][0] * hv[c][0] + hv[c][1] * hv[c][1] + hv[c][2] * hv[c][2] + hv[c][3] * hv[c][3];
        }
      ss = block_sum<GV_NW>(ss, bred);                        // waves 4.. add exact zeros: the sum of rmsnorm_fwd_kernel
      const float r = rsqrtf(ss / K + eps);
      c = 0;
      if (tid < 256)
        for (int i = tid * 4; i < K; i += 1024, ++c) {
          const float4_t g = *reinterpret_cast<const float4_t*>(norm_w + i);   // queues behind the weights: they are needed first anyway
          uint2 pk;
          pk.x = pack_bf2(g[0] * (hv[c][0] * r), g[1] * (hv[c][1] * r));
          pk.y = pack_bf2(g[2] * (hv[c][2] * r), g[3] * (hv[c][3] * r));
          *reinterpret_cast<uint2*>(xs + (size_t)m * K + i) = pk;
        }
    }
  }
  __syncthreads();
  const int mrow = lr < M ? lr : M - 1;
  const bf16_t* xp = xs + (size_t)mrow * K + lg * 16;
  float4_t acc = (float4_t){0.f, 0.f, 0.f, 0.f};
  while (have) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int k = (s + u) * 64;
      const short8_t x0 = *reinterpret_cast<const short8_t*>(xp + k), x1 = *reinterpret_cast<const short8_t*>(xp + k + 8);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0, w0[u], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, w1[u], acc, 0, 0, 0);
    }
    s += UNROLL;
    have = s + UNROLL <= s_end;
    if (have) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        w0[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + (size_t)(s - s0 + u) * 1024));
        w1[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + (size_t)(s - s0 + u) * 1024 + 512));
      }
    }
  }
  if (s < s_end) {
    // the remainder of this wave's K range (K = 11008: 22 steps = 2 batches + 6) as ONE partial batch: its loads fly together
    // instead of one load latency per step; same accumulation order
    const int rem = s_end - s;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (u < rem) {
        w0[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + (size_t)(s - s0 + u) * 1024));
        w1[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + (size_t)(s - s0 + u) * 1024 + 512));
      }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (u < rem) {
        const int k = (s + u) * 64;
        const short8_t x0 = *reinterpret_cast<const short8_t*>(xp + k), x1 = *reinterpret_cast<const short8_t*>(xp + k + 8);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0, w0[u], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, w1[u], acc, 0, 0, 0);
      }
  }
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

#define GV_PRO_LDS_MAX (64 * 1024)
#define GV_PRO_MAX_ROWS 2
template <int PRO>
static int launch_gemv_pro(const void* A, long lda, const void* P, void* C, int ldc, int M, int N, int K, const float* bias,
                           const float* residual, int ldr, int out_f32, float alpha, const float* norm_w, float eps,
                           hipStream_t stream) {
  if (M <= 0 || N <= 0) return MH_OK;
  if (M > 16 || K <= 0 || (K % 64) != 0 || ((uintptr_t)A & 15) || ((uintptr_t)P & 15)) return MH_ERR_ARG;
  if (PRO == 1 && ((K % 128) != 0 || (lda % 8) != 0 || lda < 2L * K)) return MH_ERR_ARG;
  if (PRO == 2 && (!norm_w || (K % 4) != 0 || (lda % 4) != 0)) return MH_ERR_ARG;
  if (PRO == 2 && K > 4096) return MH_ERR_UNSUPPORTED;
  const size_t sh = (size_t)M * K * 2;
  // every workgroup rebuilds all M rows: measured at batch 8 (decode, M = 8) the fused step costs 6.5 ms per token against
  // 4.1 ms with the separate launches, at batch 1 it saves 0.3 ms -- fused for up to GV_PRO_MAX_ROWS rows only
  if (sh > GV_PRO_LDS_MAX || M > GV_PRO_MAX_ROWS) return MH_ERR_UNSUPPORTED;        // the caller falls back to the two-launch form
  const dim3 grid((N + 15) / 16);
  static bool attr8 = false, attr4 = false;                   // once per instantiation, outside any stream capture
  if (gv_packed_nw(N) == 8) {
    if (!attr8) { (void)hipFuncSetAttribute((const void*)gemv_pro_kernel<8, 8, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, GV_PRO_LDS_MAX); attr8 = true; }
    hipLaunchKernelGGL((gemv_pro_kernel<8, 8, PRO>), grid, dim3(512), sh, stream, A, lda, (const bf16_t*)P, C, bias, residual, M, N, K,
                       ldc, ldr, out_f32, alpha, norm_w, eps);
  } else {
    if (!attr4) { (void)hipFuncSetAttribute((const void*)gemv_pro_kernel<8, 4, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, GV_PRO_LDS_MAX); attr4 = true; }
    hipLaunchKernelGGL((gemv_pro_kernel<8, 4, PRO>), grid, dim3(256), sh, stream, A, lda, (const bf16_t*)P, C, bias, residual, M, N, K,
                       ldc, ldr, out_f32, alpha, norm_w, eps);
  }
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// C[M <= 16, N] = alpha * rmsnorm(H; norm_w, eps) . W^T (+bias) (+residual), W as its mh_gemv_pack copy; H [M, K] f32.
// MH_ERR_UNSUPPORTED when M * K * 2 bytes of operand do not fit the kernel's LDS budget (64 KiB).
extern "C" int mh_gemv_packed_rmsnorm(const float* H, long ldh, const float* norm_w, float eps, const void* P, void* C, int ldc,
                                      int M, int N, int K, const float* bias, const float* residual, int ldr, int out_f32,
                                      float alpha, hipStream_t stream) {
  return launch_gemv_pro<2>(H, ldh, P, C, ldc, M, N, K, bias, residual, ldr, out_f32, alpha, norm_w, eps, stream);
}

// C[M <= 16, N] = alpha * (silu(g) * u) . W^T (+bias) (+residual): gu [M, >= 2K] bf16, gate / up interleaved in blocks of 128.
extern "C" int mh_gemv_packed_silu(const void* gu, long ldgu, const void* P, void* C, int ldc, int M, int N, int K,
                                   const float* bias, const float* residual, int ldr, int out_f32, float alpha, hipStream_t stream) {
  return launch_gemv_pro<1>(gu, ldgu, P, C, ldc, M, N, K, bias, residual, ldr, out_f32, alpha, nullptr, 0.f, stream);
}

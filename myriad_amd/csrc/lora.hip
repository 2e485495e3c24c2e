// K10 (PEFT form): LoRA on q_proj / v_proj,  y = W x + (alpha/r) B (A dropout(x))   (reference myriad.py:170-180;
// arithmetic = the published peft LoRA layer).  The rank-r UP projection rides the qkv MFMA GEMM as a 64-column K
// border (myriad_amd/lora.py); these wavefront-primitive kernels do the skinny parts without materialising the
// dropout mask, transposes or [M,4096] temporaries:
//   lora_down : border[m, j] = s * sum_d keep(m,d) x[m,d] A[j,d]                  (block-per-row reductions)
//   lora_dx   : dxn[m,d] = dx_base[m,d] + keep(m,d) * s * sum_j dborder[m,j] bf16(A[j,d])   (elementwise; the forward multiplied by bf16(A))
//   lora_wgrad: dA[j,d] = sum_m s dborder[m,j] keep(m,d) x[m,d] ; dB_q[d,j] = sum_m dq[m,d] border[m,j] ; dB_v likewise
//               (partial sums over row chunks -- MFMA products at r = 8, else thread-per-column -- + fixed-order reduce: deterministic)
//   lora_refresh_border: writes bf16(B_q), bf16(B_v) into the borders of W_ext and W_ext^T.
// keep(m,d) = hash(seed, m*D+d) >= p ? 1/(1-p) : 0  -- regenerated, never stored.  R2 = 2r (q then v), r in {8,16}.
// peft gives q_proj and v_proj their own nn.Dropout, i.e. independent masks: rows j < r of A (q) see keep(seed, .), rows
// j >= r (v) see the second draw of the same hash (seed with bit 63 set; common.h dropout_keep_pair).
#include "common.h"
#include <stdint.h>
#include <stdlib.h>

// The 64-column border holds G = 64 / R2 GROUPS of R2 columns.  lora_down splits D into G ranges, one workgroup column per
// range, and writes each range's partial product into its own group; the weight border repeats [B_q | B_v] G times, so the qkv
// GEMM adds the partial products up by itself.  With 74 row blocks the kernel was a chain of load latencies on 74 CUs; 4
// groups put 296 workgroups on the chip (lora_down 24.9 -> ~9 us per layer).
#define LORA_BORDER 64
#define LR_NT 256
#define LR_NW 4
#define LR_CH 64   // row chunks of the wgrad partial sums
#define LR_WG_NT 128   // wgrad: 128 threads x 4 columns = 512 columns per workgroup

// border[m, j] = s * sum_d keep(m,d) x[m,d] A[j,d] as MFMA work: a workgroup owns 16 token rows, its 8 waves split D,
// lane (row, lg) feeds 8 consecutive k of its token row (dropped elements zeroed -- exact -- and 1/(1-p) folded into the
// output scale) against A[j, k..k+7] converted to bf16 on the fly; fp32 accumulation, cross-wave sum through LDS.
// A is read once per 16 rows (the first version: a block per row, all 256 KiB of A per row, 16 block reductions, 24 us).
// A enters the MFMA in bf16 like every other GEMM operand on the path (the border it produces is itself a bf16 operand).
#define LD_NW 8
// NORM = 1 (the single-token decode step with LoRA attached, M <= 2 rows, p = 0): x is not given -- every workgroup first builds the
// RMS-normalised rows of the f32 residual stream h itself (LDS, [M][D] bf16; rmsnorm_fwd_kernel's summation order and expression,
// as gemv.hip's gemv_pro_kernel does), writes ITS range of columns to x_ext (the operand of the bordered qkv product) and then
// runs the same products on the LDS copy: the bits of mh_rmsnorm_fwd + mh_lora_down in one launch.
template <int R2, int NORM = 0>
__global__ __launch_bounds__(LD_NW * 64) void lora_down_kernel(const bf16_t* __restrict__ x, long ldx,
                                                                const float* __restrict__ A, bf16_t* __restrict__ out,
                                                                long ldo, int M, int D, float s, float p,
                                                                unsigned long long seed, const float* __restrict__ hres = nullptr,
                                                                long ldh = 0, const float* __restrict__ norm_w = nullptr,
                                                                float eps = 0.f, bf16_t* __restrict__ xn_out = nullptr) {
  constexpr int NJ = R2 / 16;
  __shared__ float red[LD_NW][NJ][256];
  extern __shared__ __attribute__((aligned(16))) char ld_dyn[];      // NORM: the normalised rows
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * 16;
  int mrow = m0 + lr;
  mrow = mrow < M ? mrow : M - 1;
  if (NORM) {
    __shared__ float bred[LD_NW];
    bf16_t* xs = reinterpret_cast<bf16_t*>(ld_dyn);
    const int tid = threadIdx.x;
    const int c0 = blockIdx.y * (D / gridDim.y), c1 = c0 + D / gridDim.y;     // the columns this workgroup publishes
    for (int m = 0; m < M; ++m) {
      const float* xr = hres + (size_t)m * ldh;
      float4_t hv[4];                                              // D <= 4096 (checked by the launcher)
      float ss = 0.f;
      int c = 0;
      if (tid < 256)
        for (int i = tid * 4; i < D; i += 1024, ++c) {
          hv[c] = *reinterpret_cast<const float4_t*>(xr + i);
          ss += hv[c][0] * hv[c][0] + hv[c][1] * hv[c][1] + hv[c][2] * hv[c][2] + hv[c][3] * hv[c][3];
        }
      ss = block_sum<LD_NW>(ss, bred);                             // waves 4.. add exact zeros: the sum of rmsnorm_fwd_kernel
      const float rr = rsqrtf(ss / D + eps);
      c = 0;
      if (tid < 256)
        for (int i = tid * 4; i < D; i += 1024, ++c) {
          const float4_t g = *reinterpret_cast<const float4_t*>(norm_w + i);
          uint2 pk;
          pk.x = pack_bf2(g[0] * (hv[c][0] * rr), g[1] * (hv[c][1] * rr));
          pk.y = pack_bf2(g[2] * (hv[c][2] * rr), g[3] * (hv[c][3] * rr));
          *reinterpret_cast<uint2*>(xs + (size_t)m * D + i) = pk;
          if (i >= c0 && i < c1) *reinterpret_cast<uint2*>(xn_out + (size_t)m * ldx + i) = pk;
        }
    }
    __syncthreads();
    x = xs;
    ldx = D;
  }
  const float ik = 1.f / (1.f - p);
  constexpr int r = R2 / 2;
  float4_t acc[NJ], accv[NJ];                      // acc: x under the q mask, accv: x under the v mask
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) { acc[jj] = (float4_t){0.f, 0.f, 0.f, 0.f}; accv[jj] = (float4_t){0.f, 0.f, 0.f, 0.f}; }
  const int G = gridDim.y, grp = blockIdx.y;       // this workgroup's range of D: steps [gs0, gs0 + steps)
  const int steps = D / 32 / G, gs0 = grp * steps;
  const int per = (steps + LD_NW - 1) / LD_NW;
  const int s0 = gs0 + wave * per, s1 = (s0 + per) < (gs0 + steps) ? (s0 + per) : (gs0 + steps);
  // several steps of loads in flight per lane: the kernel is a chain of load latencies, not bandwidth
  constexpr int UN = 4;
  for (int st = s0; st < s1; st += UN) {
    short8_t xv[UN];
    float4_t a0[UN][NJ], a1[UN][NJ];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = (st + u < s1 ? st + u : s1 - 1) * 32 + lg * 8;     // tail steps re-read the last one, weight 0 below
      xv[u] = *reinterpret_cast<const short8_t*>(x + (long)mrow * ldx + k);
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const float* ap = A + (long)(jj * 16 + lr) * D + k;
        a0[u][jj] = *reinterpret_cast<const float4_t*>(ap);
        a1[u][jj] = *reinterpret_cast<const float4_t*>(ap + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (st + u >= s1) xv[u] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
      const int k = (st + u) * 32 + lg * 8;
      short8_t xq = xv[u], xw = xv[u];
      if (p > 0.f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float kq, kv;
          dropout_keep_pair(seed, (unsigned long long)((long)mrow * D + k + e), p, ik, kq, kv);
          if (kq == 0.f) xq[e] = 0;
          if (kv == 0.f) xw[e] = 0;
        }
      }
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const short8_t av = {(short)f2bf(a0[u][jj][0]), (short)f2bf(a0[u][jj][1]), (short)f2bf(a0[u][jj][2]), (short)f2bf(a0[u][jj][3]),
                             (short)f2bf(a1[u][jj][0]), (short)f2bf(a1[u][jj][1]), (short)f2bf(a1[u][jj][2]), (short)f2bf(a1[u][jj][3])};
        // D[m = 4*lg + r][j = lr]; a 16-row block of A that is all q (or all v) needs only its own product
        if (jj * 16 < r) acc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xq, av, acc[jj], 0, 0, 0);
        if (jj * 16 + 16 > r) accv[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xw, av, accv[jj], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave][jj][(4 * lg + q) * 16 + lr] = (jj * 16 + lr < r) ? acc[jj][q] : accv[jj][q];
  __syncthreads();
  const float scale = p > 0.f ? s * ik : s;
  for (int i = threadIdx.x; i < NJ * 256; i += LD_NW * 64) {
    const int jj = i >> 8, rem = i & 255, row = rem >> 4, j = rem & 15;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < LD_NW; ++w) v += red[w][jj][rem];
    if (m0 + row < M) out[(long)(m0 + row) * ldo + grp * R2 + jj * 16 + j] = f2bf(scale * v);
  }
}

// A thread keeps ITS four columns of all R2 rows of A in registers and walks a chunk of token rows: A is read once
// per workgroup (64 KiB) instead of once per token row (the first version re-read the 256 KiB of A for every one of the
// 1184 rows: 300 MB of L2 traffic, 19 us for a 39 MB kernel).  The R2 per-row scalars are wave-uniform (scalar loads).
// nslab > 1: dx_ext is the first of nslab split-K slabs (fp32, or bf16 with sbf; stride `slab` elements) of the dgrad GEMM: summed here, in
// slab order, exactly as splitk_reduce_kernel would have (one launch and one pass over [M, D+64] less); the summed border
// d(s*t) [M, 64] is written to border_out for the weight-gradient kernel.
template <int R2, int sbf>
__global__ __launch_bounds__(256) void lora_dx_kernel(const void* __restrict__ dx_ext, long ld, const float* __restrict__ A,
                                                      float* __restrict__ out, int M, int D, float s, float p,
                                                      unsigned long long seed, int rows_per, int nslab, long slab,
                                                      float* __restrict__ border_out) {
  const float ik = 1.f / (1.f - p);
  const int d = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (d >= D) return;
  float4_t a[R2];                                  // rounded to bf16: the A the forward's LoRA-down product multiplied by
#pragma unroll
  for (int j = 0; j < R2; ++j) {
    a[j] = *reinterpret_cast<const float4_t*>(A + (long)j * D + d);
#pragma unroll
    for (int e = 0; e < 4; ++e) a[j][e] = bf2f(f2bf(a[j][e]));
  }
  const int m0 = blockIdx.y * rows_per;
  const int m1 = (m0 + rows_per) < M ? (m0 + rows_per) : M;
  for (int m = m0; m < m1; ++m) {
    const long r0 = (long)m * ld;                   // element offset of the row in a slab (fp32, or bf16: sbf)
    // the R2 border values of a row are wave-uniform: plain indexed reads off a uniform pointer become scalar loads
    float g[R2];
#pragma unroll
    for (int j = 0; j < R2; ++j) g[j] = 0.f;
    float4_t base = (float4_t){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nslab; ++k) {
      const long rk = r0 + (long)k * slab;
      if (sbf) {
        const unsigned* bp = reinterpret_cast<const unsigned*>(reinterpret_cast<const bf16_t*>(dx_ext) + rk + D);
#pragma unroll
        for (int j = 0; j < R2; j += 2) {
          const unsigned u = bp[j >> 1];
          g[j] += bf2f((bf16_t)(u & 0xffffu));
          g[j + 1] += bf2f((bf16_t)(u >> 16));
        }
      } else {
        const float* fp = reinterpret_cast<const float*>(dx_ext) + rk + D;
#pragma unroll
        for (int j = 0; j < R2; ++j) g[j] += fp[j];
      }
      const float4_t bk = slab_load4(dx_ext, rk + d, sbf);
      base[0] += bk[0]; base[1] += bk[1]; base[2] += bk[2]; base[3] += bk[3];
    }
    if (border_out && blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
      for (int j = 0; j < R2; j += 4)
        *reinterpret_cast<float4_t*>(border_out + (long)m * 64 + j) = (float4_t){g[j], g[j + 1], g[j + 2], g[j + 3]};
    }
    float4_t acc = (float4_t){0.f, 0.f, 0.f, 0.f}, accv = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < R2 / 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(g[j], a[j][e], acc[e]);
#pragma unroll
    for (int j = R2 / 2; j < R2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) accv[e] = __builtin_fmaf(g[j], a[j][e], accv[e]);
    float4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float kq, kv;
      dropout_keep_pair(seed, (unsigned long long)((long)m * D + d + e), p, ik, kq, kv);
      o[e] = lora_dx_value(base[e], s, acc[e], kq, accv[e], kv);
    }
    *reinterpret_cast<float4_t*>(out + (long)m * D + d) = o;
  }
}

// lora_dx and the RMSNorm backward that consumes it (the layer's input norm, modeling_llama.py:66-74 under autograd) as ONE
// kernel: the [M, D] fp32 d(xn) that lora_dx wrote and rmsnorm_bwd_kernel read back never exists, and one dependent launch
// leaves the LLaMA backward chain per layer.
// Shape of the kernel: a 1024-thread workgroup (16 waves, one per CU) owns ROWS token rows; thread T owns columns 4T..4T+3 of
// every row, so ITS four columns of all R2 rows of A sit in 64 registers, loaded once per workgroup (a one-row-per-workgroup
// form re-reads the 256 KiB of A for every row: 300 MB of L2 traffic, measured 44 us; a 256-thread workgroup with four rows
// in registers has too few waves per CU to keep HBM busy, measured 55 us; the two separate launches 21.8 + 19.9 us).
// Bit-identical to lora_dx_kernel followed by rmsnorm_bwd_kernel: the same expressions (common.h: lora_dx_value / rms_bwd_sums /
// rms_bwd_value) and the SAME summation order -- rmsnorm_bwd_kernel's thread t adds its chunks c = 0..3 (columns c * 1024 + 4t..)
// one after the other, then a wave butterfly, then the four wave sums in order; here chunk c of that thread is thread
// c * 256 + t, so the running sums are handed from wave group c to c + 1 through LDS (three hand-offs for all rows at once)
// and group 3 finishes with the same butterfly and the same four-term sum.
#define LXN_NT 1024
template <int R2, int sbf, int ROWS>
__global__ __launch_bounds__(LXN_NT) void lora_dx_rmsnorm_bwd_kernel(
    const void* __restrict__ dx_ext, long ld, int nslab, long slab, const float* __restrict__ A, const float* __restrict__ x,
    const float* __restrict__ w, const float* dres, float* dx, bf16_t* dx_bf, float* __restrict__ border_out, int M, int D,
    float s, float p, unsigned long long seed, float eps) {
  __shared__ float s_g[ROWS][R2];
  __shared__ float2 s_chain[ROWS][256];
  __shared__ float s_red[2 * ROWS][4];
  __shared__ float4_t s_dxn[ROWS][LXN_NT];                     // 16 KiB per row
  const int T = threadIdx.x, lane = T & 63, wave = T >> 6, grp = T >> 8, t = T & 255;
  const int d = T * 4;
  const bool live = d < D;
  const int m0 = blockIdx.x * ROWS;
  const float ik = 1.f / (1.f - p);
  // d(s*t): the R2 border values of each row, summed over the slabs in slab order
  if (T < ROWS * R2) {
    const int r = T / R2, j = T - r * R2, m = m0 + r;
    float g = 0.f;
    if (m < M) {
      for (int k = 0; k < nslab; ++k) {
        const long idx = (long)m * ld + (long)k * slab + D + j;
        g += sbf ? bf2f(reinterpret_cast<const bf16_t*>(dx_ext)[idx]) : reinterpret_cast<const float*>(dx_ext)[idx];
      }
      if (border_out) border_out[(long)m * 64 + j] = g;
    }
    s_g[r][j] = g;
  }
  // Register budget: 16 waves per workgroup leave 128 registers per thread.  A is held as packed bf16 pairs (32 registers for
  // 16 x 4 values) -- the values lora_dx_kernel multiplies by: the dx correction uses the bf16-rounded A the forward's
  // LoRA-down product multiplied by.  The rows' x values and slabs are all requested before the first is used.
  float4_t ww = (float4_t){0.f, 0.f, 0.f, 0.f};
  float4_t xv[ROWS], gv[ROWS];
  if (live) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int m = (m0 + r) < M ? (m0 + r) : (M - 1);        // rows past M redo the last row; nothing of theirs is stored
      xv[r] = *reinterpret_cast<const float4_t*>(x + (long)m * D + d);
    }
    ww = *reinterpret_cast<const float4_t*>(w + d);
    // The LoRA pass walks the rows in a real loop (unrolled, the compiler interleaves all rows and spills 150 registers) with the
    // slabs of the next two rows in flight, and parks each row's d(xn) in LDS.
    float4_t nb0[3], nb1[3];
    auto slab_req = [&](int r, int k) -> float4_t {
      const int m = (m0 + r) < M ? (m0 + r) : (M - 1);
      return slab_load4(dx_ext, (long)m * ld + (long)k * slab + d, sbf);
    };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      nb0[k] = k < nslab ? slab_req(0, k) : (float4_t){0.f, 0.f, 0.f, 0.f};
      nb1[k] = (k < nslab && ROWS > 1) ? slab_req(1, k) : (float4_t){0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("" ::: "memory");                             // ... in flight before A: 32 CUs per XCD queue on the same L2 lines there
    uint2 apk[R2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {                           // two batches of loads: 32 transient registers, not 64
#pragma unroll
      for (int j = h2 * (R2 / 2); j < (h2 + 1) * (R2 / 2); ++j) {
        const float4_t af = *reinterpret_cast<const float4_t*>(A + (long)j * D + d);
        apk[j].x = pack_bf2(af[0], af[1]);
        apk[j].y = pack_bf2(af[2], af[3]);
      }
      asm volatile("" ::: "memory");
    }
    __syncthreads();                                           // s_g
#pragma unroll 1
    for (int r = 0; r < ROWS; ++r) {
      const int m = (m0 + r) < M ? (m0 + r) : (M - 1);
      float4_t base = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < nslab) { base[0] += nb0[k][0]; base[1] += nb0[k][1]; base[2] += nb0[k][2]; base[3] += nb0[k][3]; }
      for (int k = 3; k < nslab; ++k) {                        // more than three slabs (small M): in order, as they come
        const float4_t bk = slab_req(r, k);
        base[0] += bk[0]; base[1] += bk[1]; base[2] += bk[2]; base[3] += bk[3];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        nb0[k] = nb1[k];
        if (k < nslab && r + 2 < ROWS) nb1[k] = slab_req(r + 2, k);
      }
      float4_t acc = (float4_t){0.f, 0.f, 0.f, 0.f}, accv = (float4_t){0.f, 0.f, 0.f, 0.f};
      unsigned hi, sh;                                         // 0xffff0000 and 16 the compiler cannot see through: the unpack of A is
      asm volatile("v_mov_b32 %0, 0xffff0000\n\tv_mov_b32 %1, 16" : "=v"(hi), "=v"(sh));   // redone per row, not hoisted out of
#pragma unroll                                                 // the loop into 64 more registers
      for (int j = 0; j < R2; ++j) {
        const float g = s_g[r][j];
        const float a0 = __uint_as_float(apk[j].x << sh), a1 = __uint_as_float(apk[j].x & hi);
        const float a2 = __uint_as_float(apk[j].y << sh), a3 = __uint_as_float(apk[j].y & hi);
        if (j < R2 / 2) {
          acc[0] = __builtin_fmaf(g, a0, acc[0]); acc[1] = __builtin_fmaf(g, a1, acc[1]);
          acc[2] = __builtin_fmaf(g, a2, acc[2]); acc[3] = __builtin_fmaf(g, a3, acc[3]);
        } else {
          accv[0] = __builtin_fmaf(g, a0, accv[0]); accv[1] = __builtin_fmaf(g, a1, accv[1]);
          accv[2] = __builtin_fmaf(g, a2, accv[2]); accv[3] = __builtin_fmaf(g, a3, accv[3]);
        }
      }
      float4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float kq, kv;
        dropout_keep_pair(seed, (unsigned long long)((long)m * D + d + e), p, ik, kq, kv);
        o[e] = lora_dx_value(base[e], s, acc[e], kq, accv[e], kv);     // = d(xn), what lora_dx_kernel stores
      }
      s_dxn[r][T] = o;
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) gv[r] = s_dxn[r][T];        // each thread reads back its own values: no barrier needed
  } else {
    __syncthreads();
  }
  // the residual gradient of the output pass is requested now: it lands while the row sums are handed from group to group
  float4_t dd[ROWS];
  if (live && dres) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const long m = (m0 + r) < M ? (m0 + r) : (M - 1);
      dd[r] = *reinterpret_cast<const float4_t*>(dres + m * D + d);
    }
  }
  // row sums in rmsnorm_bwd_kernel's order: wave group c continues the running sums of group c - 1
  float ss[ROWS], dot[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) ss[r] = dot[r] = 0.f;
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    if (grp == h) {
      if (h > 0) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { const float2 c2 = s_chain[r][t]; ss[r] = c2.x; dot[r] = c2.y; }
      }
      if (live) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
          for (int e = 0; e < 4; ++e) rms_bwd_sums(xv[r][e], ww[e], gv[r][e], ss[r], dot[r]);
      }
      if (h < 3) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) s_chain[r][t] = make_float2(ss[r], dot[r]);
      }
    }
    if (h < 3) __syncthreads();
  }
  if (grp == 3) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const float sa = wave_sum(ss[r]), sb = wave_sum(dot[r]);
      if (lane == 0) { s_red[2 * r][wave - 12] = sa; s_red[2 * r + 1][wave - 12] = sb; }
    }
  }
  __syncthreads();
  if (!live) return;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const long m = m0 + r;
    if (m < M) {
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { sa += s_red[2 * r][i]; sb += s_red[2 * r + 1][i]; }
      const float rr = rsqrtf(sa / D + eps);
      const float cc = rr * rr * rr * sb / D;
      float4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rms_bwd_value(rr, ww[e], gv[r][e], xv[r][e], cc);
      if (dres) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += dd[r][e];
      }
      if (dx) *reinterpret_cast<float4_t*>(dx + m * D + d) = o;
      if (dx_bf) {
        uint2 pk;
        pk.x = pack_bf2(o[0], o[1]);
        pk.y = pack_bf2(o[2], o[3]);
        *reinterpret_cast<uint2*>(dx_bf + m * D + d) = pk;
      }
    }
  }
}

// partial sums over a row chunk; one thread per FOUR adjacent columns (8-byte bf16 loads instead of 2-byte ones: the
// first version, one column per thread, ran at 40 us for ~45 MB of traffic)
template <int R2>
__global__ __launch_bounds__(LR_WG_NT) void lora_wgrad_partial_kernel(
    const bf16_t* __restrict__ x, long ldx, const float* __restrict__ dx_ext, long ldg, const bf16_t* __restrict__ dq,
    const bf16_t* __restrict__ dv, long ldq, const bf16_t* __restrict__ border, long ldb, float* __restrict__ pA,
    float* __restrict__ pBq, float* __restrict__ pBv, int M, int D, float s, float p, unsigned long long seed) {
  constexpr int r = R2 / 2;
  const int d = (blockIdx.x * LR_WG_NT + threadIdx.x) * 4;
  const int chunk = blockIdx.y;
  const int rows_per = (M + LR_CH - 1) / LR_CH;
  const int m0 = chunk * rows_per;
  const int m1 = (m0 + rows_per) < M ? (m0 + rows_per) : M;
  const float ik = 1.f / (1.f - p);
  const bool live = d < D;                        // D % 4 == 0 (checked by the launcher)
  float a[R2][4], bq[r][4], bv[r][4];
#pragma unroll
  for (int j = 0; j < R2; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) a[j][e] = 0.f;
#pragma unroll
  for (int j = 0; j < r; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { bq[j][e] = 0.f; bv[j][e] = 0.f; }
  // The 2*R2 per-row scalars (s * d(border), border summed over its groups) are staged through LDS for 32 rows at a time: as
  // wave-uniform global reads inside the row loop they were one exposed scalar-load latency per row (19 per workgroup).
  // Three token rows per trip so that their vector loads overlap.
  __shared__ float s_sg[32][R2], s_st[32][R2];
  constexpr int RU = 3;
  for (int mb0 = m0; mb0 < m1; mb0 += 32) {
    const int nrow = (m1 - mb0) < 32 ? (m1 - mb0) : 32;
    __syncthreads();
    for (int i = threadIdx.x; i < nrow * R2; i += LR_WG_NT) {
      const int rr = i / R2, j = i - rr * R2;
      const long m = mb0 + rr;
      s_sg[rr][j] = s * dx_ext[m * ldg + D + j];
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < LORA_BORDER / R2; ++g) t += bf2f(border[m * ldb + g * R2 + j]);   // the border's groups add up to s * t
      s_st[rr][j] = t;
    }
    __syncthreads();
    // software pipeline over trips of RU rows: the loads of trip t+1 are in flight while trip t is multiplied
    short4_t xn[RU], qn[RU], vn[RU];
    auto load_trip = [&](int mb) {
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int m = (mb + u) < m1 ? (mb + u) : (m1 - 1);
        xn[u] = qn[u] = vn[u] = (short4_t){0, 0, 0, 0};
        if (live) {
          xn[u] = *reinterpret_cast<const short4_t*>(x + (long)m * ldx + d);
          qn[u] = *reinterpret_cast<const short4_t*>(dq + (long)m * ldq + d);
          vn[u] = *reinterpret_cast<const short4_t*>(dv + (long)m * ldq + d);
        }
      }
    };
    load_trip(mb0);
    for (int mb = mb0; mb < mb0 + nrow; mb += RU) {
      short4_t xv[RU], qv[RU], vv[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) { xv[u] = xn[u]; qv[u] = qn[u]; vv[u] = vn[u]; }
      if (mb + RU < mb0 + nrow) load_trip(mb + RU);
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int m = mb + u;
        if (m >= mb0 + nrow) break;
        const float* sg = s_sg[m - mb0];
        const float* st = s_st[m - mb0];
        if (live) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float kq, kv;
            dropout_keep_pair(seed, (unsigned long long)((long)m * D + d + e), p, ik, kq, kv);
            const float x0 = bf2f((bf16_t)xv[u][e]);
            const float xd = x0 * kq, xw = x0 * kv;
            const float gq = bf2f((bf16_t)qv[u][e]), gv = bf2f((bf16_t)vv[u][e]);
#pragma unroll
            for (int j = 0; j < r; ++j) { a[j][e] += sg[j] * xd; a[r + j][e] += sg[r + j] * xw; }
#pragma unroll
            for (int j = 0; j < r; ++j) {
              bq[j][e] += gq * st[j];
              bv[j][e] += gv * st[r + j];
            }
          }
        }
      }
    }
  }
  if (live) {
#pragma unroll
    for (int j = 0; j < R2; ++j)
      *reinterpret_cast<float4_t*>(pA + ((long)chunk * R2 + j) * D + d) = (float4_t){a[j][0], a[j][1], a[j][2], a[j][3]};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int j = 0; j < r; j += 4) {           // r floats of one column are contiguous: 16-byte stores
        *reinterpret_cast<float4_t*>(pBq + ((long)chunk * D + d + e) * r + j) = (float4_t){bq[j][e], bq[j + 1][e], bq[j + 2][e], bq[j + 3][e]};
        *reinterpret_cast<float4_t*>(pBv + ((long)chunk * D + d + e) * r + j) = (float4_t){bv[j][e], bv[j + 1][e], bv[j + 2][e], bv[j + 3][e]};
      }
  }
}
// The same partial sums as MFMA work (r = 8, D % 128 == 0): the sums over token rows are [D x rows] . [rows x 16] products.  A
// workgroup owns 128 columns of one row chunk; 32 rows at a time it stages x (twice: with q's and with v's dropped elements
// zeroed -- exact; 1/(1-p) multiplies the finished sum), dq and dv key-major in LDS, reads them as the A operand through the
// gfx950 transpose read, and multiplies by the 16 per-row scalars (s d(border); the border summed over its groups), each split
// into a bf16 head and a bf16 remainder so that the products carry ~16 mantissa bits of the scalar (fp32 accumulation).  The
// thread-per-column form above spent its time in 2 * R2 dependent FMAs per element; here the kernel is the loads.
#define LWM_CB 128                 // columns per workgroup
#define LWM_RS (LWM_CB + 16)       // image row stride (elements): 288 B = 72 dwords, consecutive rows 8 banks apart
#define LWM_SS 40                  // row stride of the scalar images [16][32]
#define LWM_RC 16                  // row chunks (LWM_RC * D / LWM_CB = 512 workgroups at D = 4096)
typedef __attribute__((address_space(3))) short4_t lw_lds_s4;
__device__ __forceinline__ short8_t lw_frag_tr(const bf16_t* img, int jd, int lr, int lg) {
  const bf16_t* p = img + (4 * lg + (lr >> 2)) * LWM_RS + 16 * jd + 4 * (lr & 3);
  const short4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lw_lds_s4*)p);
  const short4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lw_lds_s4*)(p + 16 * LWM_RS));
  return (short8_t){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
// the B operand that goes with it: lane (j, lg) holds rows {4 lg .. +3} U {16 + 4 lg .. +3} of scalar column j
__device__ __forceinline__ short8_t lw_frag_sc(const bf16_t* sc, int lr, int lg) {
  const short4_t a = *reinterpret_cast<const short4_t*>(sc + lr * LWM_SS + 4 * lg);
  const short4_t b = *reinterpret_cast<const short4_t*>(sc + lr * LWM_SS + 16 + 4 * lg);
  return (short8_t){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__global__ __launch_bounds__(256) void lora_wgrad_mfma_kernel(
    const bf16_t* __restrict__ x, long ldx, const float* __restrict__ dx_ext, long ldg, const bf16_t* __restrict__ dq,
    const bf16_t* __restrict__ dv, long ldq, const bf16_t* __restrict__ border, long ldb, float* __restrict__ pA,
    float* __restrict__ pBq, float* __restrict__ pBv, int M, int D, float s, float p, unsigned long long seed) {
  constexpr int R2 = 16, r = 8;
  __shared__ __attribute__((aligned(16))) bf16_t img[4][32 * LWM_RS];    // x (q mask), x (v mask), dq, dv
  __shared__ __attribute__((aligned(16))) bf16_t sc[4][16 * LWM_SS];     // sg head / remainder, st head / remainder, [j][row]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
  const int d0 = blockIdx.x * LWM_CB;
  const int chunk = blockIdx.y;
  const int rows_per = (M + LWM_RC - 1) / LWM_RC;
  const int m0 = chunk * rows_per;
  const int m1 = (m0 + rows_per) < M ? (m0 + rows_per) : M;
  const float ik = 1.f / (1.f - p);
  // staging role: row (tid >> 3) of the 32, 16 columns from 16 * (tid & 7); scalars: the same row, columns 2 * (tid & 7), +1
  const int srow = tid >> 3, scol = (tid & 7) * 16, sj = (tid & 7) * 2;
  short8_t xr[2], qr[2], vr[2];
  float sgr[2], str[2];
  auto load_step = [&](int mb) {
    const int m = mb + srow;
    const short8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    xr[0] = xr[1] = qr[0] = qr[1] = vr[0] = vr[1] = z;
    sgr[0] = sgr[1] = str[0] = str[1] = 0.f;
    if (m < m1) {
      const bf16_t* xp = x + (long)m * ldx + d0 + scol;
      const bf16_t* qp = dq + (long)m * ldq + d0 + scol;
      const bf16_t* vp = dv + (long)m * ldq + d0 + scol;
      xr[0] = *reinterpret_cast<const short8_t*>(xp);
      xr[1] = *reinterpret_cast<const short8_t*>(xp + 8);
      qr[0] = *reinterpret_cast<const short8_t*>(qp);
      qr[1] = *reinterpret_cast<const short8_t*>(qp + 8);
      vr[0] = *reinterpret_cast<const short8_t*>(vp);
      vr[1] = *reinterpret_cast<const short8_t*>(vp + 8);
      sgr[0] = dx_ext[(long)m * ldg + D + sj];
      sgr[1] = dx_ext[(long)m * ldg + D + sj + 1];
#pragma unroll
      for (int g = 0; g < LORA_BORDER / R2; ++g) {                 // the border's groups add up to s * t
        str[0] += bf2f(border[(long)m * ldb + g * R2 + sj]);
        str[1] += bf2f(border[(long)m * ldb + g * R2 + sj + 1]);
      }
    }
  };
  auto store_step = [&](int mb) {
    const int m = mb + srow;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      short8_t xq = xr[h], xv = xr[h];
      if (p > 0.f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float kq, kv;
          dropout_keep_pair(seed, (unsigned long long)((long)m * D + d0 + scol + 8 * h + e), p, ik, kq, kv);
          if (kq == 0.f) xq[e] = 0;
          if (kv == 0.f) xv[e] = 0;
        }
      }
      const int o = srow * LWM_RS + scol + 8 * h;
      *reinterpret_cast<short8_t*>(&img[0][o]) = xq;
      *reinterpret_cast<short8_t*>(&img[1][o]) = xv;
      *reinterpret_cast<short8_t*>(&img[2][o]) = qr[h];
      *reinterpret_cast<short8_t*>(&img[3][o]) = vr[h];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float g = s * sgr[h], t = str[h];
      const bf16_t gh = f2bf(g), th = f2bf(t);
      const int o = (sj + h) * LWM_SS + srow;
      sc[0][o] = gh;
      sc[1][o] = f2bf(g - bf2f(gh));
      sc[2][o] = th;
      sc[3][o] = f2bf(t - bf2f(th));
    }
  };
  constexpr int NJD = LWM_CB / 16 / 4;             // 16-column blocks per wave
  float4_t aq[NJD], av[NJD], bq[NJD], bv[NJD];
#pragma unroll
  for (int i = 0; i < NJD; ++i) aq[i] = av[i] = bq[i] = bv[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
  load_step(m0);
  for (int mb = m0; mb < m1; mb += 32) {
    __syncthreads();                               // the previous step's operand reads are done
    store_step(mb);
    __syncthreads();
    if (mb + 32 < m1) load_step(mb + 32);          // in flight under the products
    const short8_t gh = lw_frag_sc(sc[0], lr, lg), gl = lw_frag_sc(sc[1], lr, lg);
    const short8_t th = lw_frag_sc(sc[2], lr, lg), tl = lw_frag_sc(sc[3], lr, lg);
#pragma unroll
    for (int i = 0; i < NJD; ++i) {
      const int jd = wave * NJD + i;
      const short8_t fq = lw_frag_tr(img[0], jd, lr, lg), fv = lw_frag_tr(img[1], jd, lr, lg);
      const short8_t fdq = lw_frag_tr(img[2], jd, lr, lg), fdv = lw_frag_tr(img[3], jd, lr, lg);
      aq[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq, gh, aq[i], 0, 0, 0);
      aq[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq, gl, aq[i], 0, 0, 0);
      av[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, gh, av[i], 0, 0, 0);
      av[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, gl, av[i], 0, 0, 0);
      bq[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fdq, th, bq[i], 0, 0, 0);
      bq[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fdq, tl, bq[i], 0, 0, 0);
      bv[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fdv, th, bv[i], 0, 0, 0);
      bv[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fdv, tl, bv[i], 0, 0, 0);
    }
  }
  // accumulator of lane (j = lr, lg): columns d = 16 jd + 4 lg + e of scalar column j.  Rows j < r of A belong to q (x under q's
  // mask), the others to v; the border column j < r multiplies dq, the others dv.
#pragma unroll
  for (int i = 0; i < NJD; ++i) {
    const int d = d0 + 16 * (wave * NJD + i) + 4 * lg;
    const float4_t a = lr < r ? aq[i] : av[i];
    *reinterpret_cast<float4_t*>(pA + ((long)chunk * R2 + lr) * D + d) = (float4_t){a[0] * ik, a[1] * ik, a[2] * ik, a[3] * ik};
    float* pb = lr < r ? pBq : pBv;
    const float4_t b = lr < r ? bq[i] : bv[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) pb[((long)chunk * D + d + e) * r + (lr & (r - 1))] = b[e];
  }
}

__global__ void lora_wgrad_reduce_kernel(const float* __restrict__ pA, const float* __restrict__ pBq,
                                         const float* __restrict__ pBv, float* __restrict__ dA, float* __restrict__ dBq,
                                         float* __restrict__ dBv, int nA, int nB, int nch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // chunks in ascending order, eight loads in flight at a time (nch is 16 or 64)
  if (i < nA) {
    float s = 0.f;
    for (int c = 0; c < nch; c += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pA[(long)(c + u) * nA + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    dA[i] = s;
  }
  if (i < nB) {
    float s = 0.f, t = 0.f;
    for (int c = 0; c < nch; c += 8) {
      float v[8], x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { v[u] = pBq[(long)(c + u) * nB + i]; x[u] = pBv[(long)(c + u) * nB + i]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s += v[u]; t += x[u]; }
    }
    dBq[i] = s;
    dBv[i] = t;
  }
}

__global__ void lora_refresh_kernel(const float* __restrict__ Bq, const float* __restrict__ Bv, bf16_t* __restrict__ ext,
                                    long ld_ext, bf16_t* __restrict__ extT, long ld_extT, int W, int D, int r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // i = n*r + j
  if (i >= W * r) return;
  const int n = i / r, j = i - n * r;
  const bf16_t q = f2bf(Bq[i]), v = f2bf(Bv[i]);
  for (int g = 0; g < LORA_BORDER / (2 * r); ++g) {
    const int c = D + g * 2 * r;
    ext[(long)n * ld_ext + c + j] = q;
    ext[(long)(2 * W + n) * ld_ext + c + r + j] = v;
    if (extT) {   // absent when the model was built without backward support (inference)
      extT[(long)(c + j) * ld_extT + n] = q;
      extT[(long)(c + r + j) * ld_extT + 2 * W + n] = v;
    }
  }
}

// all layers in one launch: tab[layer] = {B_q, B_v, W_ext, W_ext^T (or NULL)} device pointers, blockIdx.y = layer
struct LoraRefreshEntry { const float* Bq; const float* Bv; bf16_t* ext; bf16_t* extT; };
__global__ void lora_refresh_all_kernel(const LoraRefreshEntry* __restrict__ tab, long ld_ext, long ld_extT, int W, int D, int r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // i = n*r + j
  if (i >= W * r) return;
  const LoraRefreshEntry e = tab[blockIdx.y];
  const int n = i / r, j = i - n * r;
  const bf16_t q = f2bf(e.Bq[i]), v = f2bf(e.Bv[i]);
  for (int g = 0; g < LORA_BORDER / (2 * r); ++g) {      // every group of the border carries [B_q | B_v]
    const int c = D + g * 2 * r;
    e.ext[(long)n * ld_ext + c + j] = q;
    e.ext[(long)(2 * W + n) * ld_ext + c + r + j] = v;
    if (e.extT) {
      e.extT[(long)(c + j) * ld_extT + n] = q;
      e.extT[(long)(c + r + j) * ld_extT + 2 * W + n] = v;
    }
  }
}

#define LORA_DISPATCH(R2_, CALL)                      \
  switch (R2_) {                                      \
    case 16: { constexpr int R2 = 16; CALL; break; }  \
    case 32: { constexpr int R2 = 32; CALL; break; }  \
    default: return MH_ERR_UNSUPPORTED;               \
  }

extern "C" int mh_lora_down(const void* x, long ldx, const float* A, void* border, long ldo, int M, int D, int R2_,
                            float s, float p, unsigned long long seed, hipStream_t stream) {
  if (M <= 0) return MH_OK;
  if (D % 4 || ldx % 4 || p < 0.f || p >= 1.f) return MH_ERR_ARG;
  if (D % 32 || ldx % 8 || R2_ <= 0 || R2_ > LORA_BORDER) return MH_ERR_ARG;
  int G = LORA_BORDER / R2_;                       // groups of the border; an unused group stays zero and contributes nothing
  while (G > 1 && (D % (32 * G)) != 0) G >>= 1;
  LORA_DISPATCH(R2_, hipLaunchKernelGGL(lora_down_kernel<R2>, dim3((M + 15) / 16, G), dim3(LD_NW * 64), 0, stream,
                                        (const bf16_t*)x, ldx, A, (bf16_t*)border, ldo, M, D, s, p, seed));
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// x_ext[:, :D] = bf16(rmsnorm(h; w, eps)) and x_ext[:, D:D+64] = the LoRA border of those rows (no dropout), M <= 2 rows: the head of a
// single-token decode step with LoRA attached.  MH_ERR_UNSUPPORTED for more rows / wider models: the caller runs the two launches.
extern "C" int mh_rmsnorm_lora_down(const float* h, long ldh, const float* norm_w, float eps, const float* A, void* x_ext, long ldx,
                                    int M, int D, int R2_, float s, hipStream_t stream) {
  if (M <= 0) return MH_OK;
  if (!h || !norm_w || !A || !x_ext || D % 32 || ldx % 8 || ldh % 4 || ldx < D + LORA_BORDER || R2_ <= 0 || R2_ > LORA_BORDER)
    return MH_ERR_ARG;
  if (M > 2 || D > 4096) return MH_ERR_UNSUPPORTED;
  int G = LORA_BORDER / R2_;
  while (G > 1 && (D % (32 * G)) != 0) G >>= 1;
  bf16_t* xe = (bf16_t*)x_ext;
  LORA_DISPATCH(R2_, hipLaunchKernelGGL((lora_down_kernel<R2, 1>), dim3(1, G), dim3(LD_NW * 64), (size_t)M * D * 2, stream,
                                        (const bf16_t*)nullptr, ldx, A, xe + D, ldx, M, D, s, 0.f, 0ULL, h, ldh, norm_w, eps, xe));
  MH_CHECK_LAUNCH();
  return MH_OK;
}

int mh_launch_lora_dx(const void* dx_ext, int slab_bf16, long ld, int nslab, long slab, const float* A, float* out, float* border_out,
                      int M, int D, int R2_, float s, float p, unsigned long long seed, hipStream_t stream) {
  if (M <= 0) return MH_OK;
  if (D % 4 || ld % 4 || ld < D + R2_ || p < 0.f || p >= 1.f || nslab < 1 || R2_ > 64) return MH_ERR_ARG;
  const int col_blocks = (D / 4 + 255) / 256;
  int row_chunks = 512 / col_blocks;               // ~512 workgroups
  row_chunks = row_chunks < 1 ? 1 : (row_chunks > M ? M : row_chunks);
  const int rows_per = (M + row_chunks - 1) / row_chunks;
  const dim3 grid(col_blocks, (M + rows_per - 1) / rows_per);
  if (slab_bf16) {
    LORA_DISPATCH(R2_, hipLaunchKernelGGL((lora_dx_kernel<R2, 1>), grid, dim3(256), 0, stream, dx_ext, ld, A, out, M, D, s, p, seed,
                                          rows_per, nslab, slab, border_out));
  } else {
    LORA_DISPATCH(R2_, hipLaunchKernelGGL((lora_dx_kernel<R2, 0>), grid, dim3(256), 0, stream, dx_ext, ld, A, out, M, D, s, p, seed,
                                          rows_per, nslab, slab, border_out));
  }
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// lora_dx + rmsnorm_bwd in one launch (D <= 4096); MH_ERR_UNSUPPORTED sends the caller to the two-launch form
int mh_launch_lora_dx_rmsnorm_bwd(const void* dx_ext, int slab_bf16, long ld, int nslab, long slab, const float* A, const float* x,
                                  const float* w, const float* dres, float* dx, void* dx_bf16, float* border_out, int M, int D,
                                  int R2_, float s, float p, unsigned long long seed, float eps, hipStream_t stream) {
  if (M <= 0) return MH_OK;
  if (D % 4 || ld % 4 || ld < D + R2_ || p < 0.f || p >= 1.f || nslab < 1 || R2_ > 64) return MH_ERR_ARG;
  if (D > 4 * LXN_NT) return MH_ERR_UNSUPPORTED;
  // a wave owns 256 consecutive columns and the kernel's barriers sit in both arms of `if (live)`: every wave has to be all live
  // or all dead (at D % 256 != 0, e.g. 3200, one wave would arrive at both arms' barriers)
  if (D % 256) return MH_ERR_UNSUPPORTED;
  if (R2_ != 16) return MH_ERR_UNSUPPORTED;          // r = 8 (the shipped config); r = 16 would need 128 registers of A per thread
  // rows per workgroup: as few as give one round of <= 256 workgroups (A is read once per workgroup), five at most
  int rows = M <= 256 ? 1 : (M <= 512 ? 2 : (M <= 768 ? 3 : (M <= 1024 ? 4 : 5)));
#ifdef MH_DEBUG_HOOKS
  { const char* e = getenv("MYRIAD_LXN_ROWS"); if (e && atoi(e) >= 1 && atoi(e) <= 5) rows = atoi(e); }
#endif
#define LXN_LAUNCH(SBF, ROWS)                                                                                                   \
  hipLaunchKernelGGL((lora_dx_rmsnorm_bwd_kernel<16, SBF, ROWS>), dim3((M + ROWS - 1) / ROWS), dim3(LXN_NT), 0, stream, dx_ext, ld,  \
                     nslab, slab, A, x, w, dres, dx, (bf16_t*)dx_bf16, border_out, M, D, s, p, seed, eps)
#define LXN_ROWS_SWITCH(SBF)                                                                          \
  switch (rows) {                                                                                     \
    case 1: LXN_LAUNCH(SBF, 1); break;                                                                \
    case 2: LXN_LAUNCH(SBF, 2); break;                                                                \
    case 3: LXN_LAUNCH(SBF, 3); break;                                                                \
    case 4: LXN_LAUNCH(SBF, 4); break;                                                                \
    default: LXN_LAUNCH(SBF, 5); break;                                                               \
  }
  if (slab_bf16) { LXN_ROWS_SWITCH(1) } else { LXN_ROWS_SWITCH(0) }
#undef LXN_ROWS_SWITCH
#undef LXN_LAUNCH
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_lora_dx(const float* dx_ext, long ld, const float* A, float* out, int M, int D, int R2_, float s,
                          float p, unsigned long long seed, hipStream_t stream) {
  return mh_launch_lora_dx(dx_ext, 0, ld, 1, 0, A, out, nullptr, M, D, R2_, s, p, seed, stream);
}

extern "C" long mh_lora_wgrad_ws_floats(int D, int R2_) { return (long)LR_CH * (2L * R2_ * D); }

extern "C" int mh_lora_wgrad(const void* x, long ldx, const float* dx_ext, long ldg, const void* dq, const void* dv,
                             long ldq, const void* border, long ldb, float* dA, float* dBq, float* dBv, float* ws, int M,
                             int D, int R2_, float s, float p, unsigned long long seed, hipStream_t stream) {
  if (M <= 0) return MH_OK;
  if (p < 0.f || p >= 1.f) return MH_ERR_ARG;
  const int r = R2_ / 2;
  float* pA = ws;
  float* pBq = pA + (long)LR_CH * R2_ * D;
  float* pBv = pBq + (long)LR_CH * D * r;
  if ((D % 4) != 0 || (ldx % 4) != 0 || (ldq % 4) != 0) return MH_ERR_ARG;
  int nch = LR_CH;
  const bool al16 = (((uintptr_t)x | (uintptr_t)dq | (uintptr_t)dv) & 15) == 0 && (ldx % 8) == 0 && (ldq % 8) == 0 && (ldg % 2) == 0;
  if (R2_ == 16 && (D % LWM_CB) == 0 && al16 && mh_opt(MH_OPT_LORA_WGRAD_MFMA)) {
    nch = LWM_RC;
    pBq = pA + (long)nch * R2_ * D;
    pBv = pBq + (long)nch * D * r;
    hipLaunchKernelGGL(lora_wgrad_mfma_kernel, dim3(D / LWM_CB, LWM_RC), dim3(256), 0, stream, (const bf16_t*)x, ldx, dx_ext, ldg,
                       (const bf16_t*)dq, (const bf16_t*)dv, ldq, (const bf16_t*)border, ldb, pA, pBq, pBv, M, D, s, p, seed);
  } else {
    const dim3 grid((D / 4 + LR_WG_NT - 1) / LR_WG_NT, LR_CH);
    LORA_DISPATCH(R2_, hipLaunchKernelGGL(lora_wgrad_partial_kernel<R2>, grid, dim3(LR_WG_NT), 0, stream, (const bf16_t*)x, ldx,
                                          dx_ext, ldg, (const bf16_t*)dq, (const bf16_t*)dv, ldq, (const bf16_t*)border, ldb,
                                          pA, pBq, pBv, M, D, s, p, seed));
  }
  MH_CHECK_LAUNCH();
  const int nA = R2_ * D, nB = D * r;
  hipLaunchKernelGGL(lora_wgrad_reduce_kernel, dim3((nA + 255) / 256), dim3(256), 0, stream, pA, pBq, pBv, dA, dBq, dBv, nA,
                     nB, nch);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_lora_refresh_borders(const void* table, int n_layers, long ld_ext, long ld_extT, int W, int D, int r,
                                       hipStream_t stream) {
  if (n_layers <= 0) return MH_OK;
  if (!table || W <= 0 || r <= 0) return MH_ERR_ARG;
  hipLaunchKernelGGL(lora_refresh_all_kernel, dim3((W * r + 255) / 256, n_layers), dim3(256), 0, stream,
                     (const LoraRefreshEntry*)table, ld_ext, ld_extT, W, D, r);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

extern "C" int mh_lora_refresh_border(const float* Bq, const float* Bv, void* ext, long ld_ext, void* extT, long ld_extT,
                                      int W, int D, int r, hipStream_t stream) {
  const int n = W * r;
  hipLaunchKernelGGL(lora_refresh_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, Bq, Bv, (bf16_t*)ext, ld_ext,
                     (bf16_t*)extT, ld_extT, W, D, r);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

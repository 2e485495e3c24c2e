// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libmyriad_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MH_OK 0
#define MH_ERR_ARG (-1)
#define MH_ERR_LAUNCH (-2)
#define MH_ERR_UNSUPPORTED (-3)

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;
typedef __attribute__((ext_vector_type(16))) float float16_t;

#define MH_CHECK_LAUNCH()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return MH_ERR_LAUNCH;            \
  } while (0)

// split-K scratch record (gemm.hip; one per mh_ctx + a process default)
#define MH_MAX_ALT_WS 4
struct MhScratch {
  float* ws;
  size_t bytes;
  float* alt[MH_MAX_ALT_WS];
  hipStream_t alt_stream[MH_MAX_ALT_WS];
};
extern MhScratch g_default_scratch;
extern MhScratch* g_scratch;

// Library options (version.hip: mh_set_option / mh_get_option; include/myriad_hip.h lists them).  Process-wide A/B switches, each
// defaulting to the measured-best setting and to its MYRIAD_* environment variable when that is set; read through mh_opt().
enum MhOpt {
  MH_OPT_SLAB_BF16 = 0,     // bf16 split-K slabs of the 256x256 kernel (0: fp32)
  MH_OPT_GEMM_SKINNY,       // 160-row tiles for 128 < M <= 320
  MH_OPT_SWIGLU_FUSED,      // SiLU gate in the gate|up / down-dgrad GEMM epilogues
  MH_OPT_GELU_FUSED,        // erf-GELU in the Q-Former MLP GEMM epilogues
  MH_OPT_ATTN_BWD_SPLIT,    // two workgroups per (batch, head) in the LLaMA attention backward at small batch
  MH_OPT_GEMM_ZERO_PAD,     // rows past M / N of the 256x256 tile read as zeros (0: copies of the last row)
  MH_OPT_GEMM256_IMPL,      // 1: hand-scheduled 64-deep loop (gemm_x4.hip), 0: the eight-wave fallback kernel (gemm_256.hip)
  MH_OPT_LORA_NORM_FUSED,   // LoRA dx correction + input-norm backward / LoRA down + norm forward as one kernel each
  MH_OPT_ATTN_FULL,         // whole-sequence forward attention (attn_full.hip) for unmasked Sk <= 288, head dim <= 96
  MH_OPT_LORA_WGRAD_MFMA,   // LoRA weight gradients as MFMA products (r = 8, D % 128 == 0); 0: the thread-per-column kernel
  MH_OPT_GEMM_SKIP_PAD,     // 256x256 kernel: waves whose rows lie past M (but for <= 2 fragments) issue no MFMAs for them
  MH_OPT_GEMM_SPLIT_XCD,    // 256x256 kernel, K-split launches: an XCD owns one split of a band of tile columns (0: all splits of a tile block)
  MH_OPT_COUNT
};
int mh_opt(int id);

// launch profiler (prof.hip): no-ops unless mh_prof_start() was called
extern bool g_mh_prof_on;
void mh_prof_pre(hipStream_t s, int kernel, int M, int N, int K, int splits, int flags);
void mh_prof_post(hipStream_t s);

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }

// round-to-nearest-even fp32 -> bf16.  gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two elements per instruction);
// the integer form (add 0x7fff + lsb, shift: ~8 VALU instructions per element) made every bf16 epilogue instruction-bound -- the
// 256 x 256 GEMM's read-out of 128 outputs per lane took 7 us of its workgroup's time (profiles/r04_gemm_x4.md).  Same bits for
// every finite input incl. denormals and ties (tests/test_kernels_gpu.py::test_bf16_rounding_in_hardware_equals_the_integer_form);
// a NaN stays a quiet NaN.  f2bf_sw keeps the integer form for that test.
typedef __bf16 mh_bf2_t __attribute__((ext_vector_type(2)));
typedef float mh_f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf_sw(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ bf16_t f2bf(float f) {
  const __bf16 b = (__bf16)f;
  return (bf16_t)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  const mh_f2_t v = {lo, hi};
  const mh_bf2_t b = __builtin_convertvector(v, mh_bf2_t);
  return __builtin_bit_cast(unsigned, b);
}

// Counter-based dropout keep-mask (PEFT lora_dropout, reference myriad.py:171-178): a pure function of (seed, flat
// element index), so backward regenerates it instead of storing it.  32-bit arithmetic only (murmur3 finaliser over a
// seed/index mix): the first version used a 64-bit splitmix and its multi-word multiplies made the LoRA kernels
// ALU-bound.  Returns 1/(1-p) for kept elements, 0 for dropped ones; u is a 24-bit uniform in [0,1).
__device__ __forceinline__ unsigned dropout_hash(unsigned long long seed, unsigned long long idx) {
  unsigned h = (unsigned)idx * 0x9E3779B1u + ((unsigned)seed ^ ((unsigned)(idx >> 32) * 0x85EBCA77u));
  h ^= (unsigned)(seed >> 32) & 0x7FFFFFFFu;
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
// second, independent draw derived from the first hash (one more multiply-xorshift round instead of a second full hash)
__device__ __forceinline__ unsigned dropout_hash2(unsigned h) {
  h = (h ^ 0x68E31DA4u) * 0x2C1B3C6Du;
  return h ^ (h >> 15);
}
// Bit 63 of the seed selects the second draw of the same (seed, index): two masks for the price of ~one hash
// (dropout_keep_pair), e.g. peft's separate nn.Dropout on q_proj and v_proj.
__device__ __forceinline__ float dropout_keep(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
  if (p <= 0.f) return 1.f;
  unsigned h = dropout_hash(seed, idx);
  if (seed >> 63) h = dropout_hash2(h);
  const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
  return u >= p ? inv_keep : 0.f;
}
__device__ __forceinline__ void dropout_keep_pair(unsigned long long seed, unsigned long long idx, float p, float inv_keep,
                                                  float& k0, float& k1) {
  if (p <= 0.f) { k0 = k1 = 1.f; return; }
  const unsigned h = dropout_hash(seed, idx);
  k0 = (float)(h >> 8) * (1.0f / 16777216.0f) >= p ? inv_keep : 0.f;
  k1 = (float)(dropout_hash2(h) >> 8) * (1.0f / 16777216.0f) >= p ? inv_keep : 0.f;
}

// Split-K partial slabs are fp32, or bf16 when the 256x256 kernel produced them (gemm.hip run_splitk): four consecutive
// elements at element index idx of a slab base, as fp32.
__device__ __forceinline__ float4_t slab_load4(const void* base, long idx, int slab_bf16) {
  if (slab_bf16) {
    const short4_t h = *reinterpret_cast<const short4_t*>(reinterpret_cast<const bf16_t*>(base) + idx);
    return (float4_t){bf2f((bf16_t)h[0]), bf2f((bf16_t)h[1]), bf2f((bf16_t)h[2]), bf2f((bf16_t)h[3])};
  }
  return *reinterpret_cast<const float4_t*>(reinterpret_cast<const float*>(base) + idx);
}

// Expression forms shared by kernels that must agree bit for bit (a fused kernel and the launches it replaces): explicit FMAs,
// no implicit contraction, so the same bits come out whatever code surrounds the call.
// LoRA dx (lora.hip): base + s * (acc_q * keep_q + acc_v * keep_v)
__device__ __forceinline__ float lora_dx_value(float base, float s, float acc, float kq, float accv, float kv) {
#pragma clang fp contract(off)
  return __builtin_fmaf(s, __builtin_fmaf(acc, kq, accv * kv), base);
}
// RMSNorm backward (norm.hip): the two row sums and the output  r * w * g - x * cc
__device__ __forceinline__ void rms_bwd_sums(float x, float w, float g, float& ss, float& dot) {
#pragma clang fp contract(off)
  ss = __builtin_fmaf(x, x, ss);
  dot = __builtin_fmaf(x * w, g, dot);
}
__device__ __forceinline__ float rms_bwd_value(float r, float w, float g, float x, float cc) {
#pragma clang fp contract(off)
  return __builtin_fmaf(-x, cc, (r * w) * g);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x = 64 * NW (NW <= 16). `red` is NW floats of LDS.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  if (NW == 1) return v;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  if (NW == 1) return v;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
  return t;
}

// erf for the GELU forms (eva_vit.py:54-61 / Qformer.py FFN: nn.GELU(), erf form): Abramowitz & Stegun 7.1.26 on |x| with the
// Gaussian factor passed in (the GELU derivative needs the same exp), absolute error <= 1.5e-7 -- three orders below the bf16
// rounding of every tensor these results are stored in, and inside the fp32 parity bounds (1e-4) of the kernel tests.  Round 4:
// the library erff() is ~40 VALU instructions with branches; in the fc1 GEMM's epilogue (one workgroup per CU, 128 outputs per
// thread, nothing overlapping the store tail) it cost 13 us of a 52 us launch (ViT fc1, 39 launches per step).
// The approximation is used in its erfc form, erfc(|u|) = poly(t) * exp(-u^2): the cdf of a negative argument is 0.5 * erfc(|u|)
// directly, not 1 - (1 - ...), so the negative tail keeps its sign and magnitude (relative error <= 1 % down to x = -8, where
// gelu is 1e-15; the 1 - erf form lost everything below x = -5: ADVICE r4).
__device__ __forceinline__ float mh_erfc_abs(float ax, float gauss /* exp(-ax * ax) */) {
  const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  return poly * gauss;
}
__device__ __forceinline__ float mh_norm_cdf(float u /* x / sqrt(2) */, float gauss) {
  const float half = 0.5f * mh_erfc_abs(fabsf(u), gauss);
  return u < 0.f ? half : 1.f - half;
}
__device__ __forceinline__ float gelu_erf(float x) {
  const float u = x * 0.70710678118654752440f;
  return x * mh_norm_cdf(u, __expf(-u * u));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float u = x * 0.70710678118654752440f;
  const float g = __expf(-u * u);                       // = exp(-x^2 / 2): the Gaussian of the pdf and of the erfc formula
  const float cdf = mh_norm_cdf(u, g);
  const float pdf = 0.39894228040143267794f * g;
  return cdf + x * pdf;
}

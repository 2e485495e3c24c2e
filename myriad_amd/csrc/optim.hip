// K15: AdamW on the flat fp32 trainable-parameter buffer (reference runner_base.py:104-139: torch.optim.AdamW,
// betas (0.9, 0.999), eps 1e-8, decoupled weight decay 0.05 on the wd group / 0 on the no-wd group), with an
// optional bf16 shadow copy written in the same pass (the copy the next step's MFMA GEMMs read).
// HBM-bound: 28 B/param algorithmic (read p,g,m,v ; write p,m,v) + 2 B for the shadow.
#include "common.h"

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, bf16_t* __restrict__ shadow, long n4, float lr, float beta1,
                             float beta2, float omb1, float omb2, float eps, float wd, float bc1, float bc2_sqrt,
                             float gscale) {
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < n4; it += (long)gridDim.x * blockDim.x) {
    float4_t pp = *reinterpret_cast<const float4_t*>(p + it * 4);
    const float4_t gg = *reinterpret_cast<const float4_t*>(g + it * 4);
    float4_t mm = *reinterpret_cast<const float4_t*>(m + it * 4);
    float4_t vv = *reinterpret_cast<const float4_t*>(v + it * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = gg[e] * gscale;
      pp[e] *= (1.f - lr * wd);
      mm[e] = beta1 * mm[e] + omb1 * gr;
      vv[e] = beta2 * vv[e] + omb2 * gr * gr;
      const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
      pp[e] -= (lr / bc1) * (mm[e] / denom);
    }
    *reinterpret_cast<float4_t*>(p + it * 4) = pp;
    *reinterpret_cast<float4_t*>(m + it * 4) = mm;
    *reinterpret_cast<float4_t*>(v + it * 4) = vv;
    if (shadow) {
      uint2 pk;
      pk.x = pack_bf2(pp[0], pp[1]);
      pk.y = pack_bf2(pp[2], pp[3]);
      *reinterpret_cast<uint2*>(shadow + it * 4) = pk;
    }
  }
}

// step is 1-based.  grad_scale multiplies the gradient first (e.g. 1/world after a sum all-reduce).
extern "C" int mh_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, long n, double lr,
                             double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale,
                             hipStream_t stream) {
  if (n <= 0) return MH_OK;
  if (n % 4 || step < 1) return MH_ERR_ARG;
  // bias corrections and (1-beta) in double on the host, like torch.optim.AdamW
  const float bc1 = (float)(1.0 - pow(beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow(beta2, (double)step));
  const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
  long grid = (n / 4 + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(adamw_kernel, dim3((int)grid), dim3(256), 0, stream, p, g, m, v, (bf16_t*)shadow_bf16, n / 4,
                     (float)lr, (float)beta1, (float)beta2, omb1, omb2, (float)eps, (float)weight_decay, bc1, bc2s,
                     (float)grad_scale);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ---- gated form: torch.optim.AdamW skips a parameter whose .grad is None (no decay, no moment update, its own step
// count) -- what happens to a module that no rank used in a step (reference myriad.py:378: random prompt stage;
// runner_base.py:96-98 find_unused_parameters).  `used` is a device float (the module's use count summed over ranks by
// the gradient all-reduce it rides on), `steps` the device-resident number of updates applied to the module so far:
// nothing is read back by the host.  mh_adamw_gated updates one contiguous range if *used > 0 with step = *steps + 1;
// mh_adamw_bump then advances the counters of the used modules (launch it after the module's last range).
__global__ void adamw_gated_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                   float* __restrict__ v, long n4, float lr, float beta1, float beta2, float eps, float wd,
                                   float gscale, const float* __restrict__ used, const int* __restrict__ steps) {
  if (*used <= 0.f) return;
  const double st = (double)(*steps + 1);
  const float bc1 = (float)(1.0 - pow((double)beta1, st));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, st));
  const float omb1 = (float)(1.0 - (double)beta1), omb2 = (float)(1.0 - (double)beta2);
  for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < n4; it += (long)gridDim.x * blockDim.x) {
    float4_t pp = *reinterpret_cast<const float4_t*>(p + it * 4);
    const float4_t gg = *reinterpret_cast<const float4_t*>(g + it * 4);
    float4_t mm = *reinterpret_cast<const float4_t*>(m + it * 4);
    float4_t vv = *reinterpret_cast<const float4_t*>(v + it * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = gg[e] * gscale;
      pp[e] *= (1.f - lr * wd);
      mm[e] = beta1 * mm[e] + omb1 * gr;
      vv[e] = beta2 * vv[e] + omb2 * gr * gr;
      const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
      pp[e] -= (lr / bc1) * (mm[e] / denom);
    }
    *reinterpret_cast<float4_t*>(p + it * 4) = pp;
    *reinterpret_cast<float4_t*>(m + it * 4) = mm;
    *reinterpret_cast<float4_t*>(v + it * 4) = vv;
  }
}
__global__ void adamw_bump_kernel(const float* __restrict__ used, int* __restrict__ steps, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && used[i] > 0.f) steps[i] += 1;
}

extern "C" int mh_adamw_gated(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2,
                              double eps, double weight_decay, double grad_scale, const float* used, const int* steps,
                              hipStream_t stream) {
  if (n <= 0) return MH_OK;
  if (n % 4 || !used || !steps) return MH_ERR_ARG;
  long grid = (n / 4 + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(adamw_gated_kernel, dim3((int)grid), dim3(256), 0, stream, p, g, m, v, n / 4, (float)lr, (float)beta1,
                     (float)beta2, (float)eps, (float)weight_decay, (float)grad_scale, used, steps);
  MH_CHECK_LAUNCH();
  return MH_OK;
}
extern "C" int mh_adamw_bump(const float* used, int* steps, int n, hipStream_t stream) {
  if (n <= 0) return MH_OK;
  hipLaunchKernelGGL(adamw_bump_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, used, steps, n);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

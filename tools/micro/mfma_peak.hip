// Microbenchmark: sustained bf16 MFMA rate of the two instruction shapes on random register data, 2 waves per SIMD
// (the residency of the 8-wave GEMM).  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((__vector_size__(8 * sizeof(short)))) short short8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float float4_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float float16_t;

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(const short8_t* in, float* out, int iters) {
  short8_t a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + 4 + i) & 4095]; }
  float s = 0.f;
  if (SHAPE == 16) {
    float4_t acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (float4_t){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  } else {
    float16_t acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + r) & 3], acc[i], 0, 0, 0);
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  short8_t* in; float* out;
  hipMalloc(&in, 4096 * sizeof(short8_t)); hipMalloc(&out, 256 * 512 * sizeof(float));
  short* h = (short*)malloc(4096 * 16);
  for (int i = 0; i < 4096 * 8; ++i) { float f = (rand() / (float)RAND_MAX - 0.5f) * 2.f; unsigned u; memcpy(&u, &f, 4); h[i] = (short)(u >> 16); }
  hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
  const int iters = 20000;
  for (int shape : {16, 32, 16, 32}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 0, 0, in, out, iters);
      else hipLaunchKernelGGL(k<32>, dim3(256), dim3(512), 0, 0, in, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per wave per iteration: 16 MFMAs x 16384 flop (16x16x32) or 8 MFMAs x 32768 flop (32x32x16) = 262144 flop
    double fl = 256.0 * 8 * iters * 262144.0;
    printf("mfma %dx%d: %.2f ms  %.0f TFLOP/s\n", shape, shape, ms, fl / (ms * 1e-3) / 1e12);
  }
  return 0;
}

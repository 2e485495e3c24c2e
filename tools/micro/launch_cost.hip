// Microbenchmark: what a back-to-back launch of a kernel with the 256x256 GEMM's footprint costs before it does anything --
// 512-thread workgroups, 128 KiB of dynamic LDS (one workgroup per CU), grids of 216 / 240 / 256 / 430 workgroups, body = one
// LDS store per thread.  The difference to the GEMM's fitted fixed cost per launch (profiles/r04_gemm_x4.md) is what its own
// prologue + epilogue + drain take.  hipcc --offload-arch=gfx950 -O3 launch_cost.hip -o launch_cost && ./launch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(float* out) {
  extern __shared__ float sm[];
  sm[threadIdx.x] = (float)blockIdx.x;
  if (out && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = sm[1];
}
__global__ __launch_bounds__(256) void k_small(float* out) {
  if (out && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = 1.f;
}
int main() {
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grids[] = {1, 216, 240, 256, 430, 512};
  for (int lds : {131072, 1024}) for (int g : grids) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(g), dim3(512), lds, s, (float*)nullptr);
    hipStreamSynchronize(s);
    const int n = 400;
    hipEventRecord(e0, s);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(g), dim3(512), lds, s, (float*)nullptr);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("512 threads, %6d B LDS, grid %3d: %.2f us per back-to-back launch\n", lds, g, ms * 1e3 / n);
  }
  for (int g : {256, 1024}) {
    const int n = 400;
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_small, dim3(g), dim3(256), 0, s, (float*)nullptr);
    hipEventRecord(e0, s);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_small, dim3(g), dim3(256), 0, s, (float*)nullptr);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("256 threads, no LDS, grid %4d: %.2f us per back-to-back launch\n", g, ms * 1e3 / n);
  }
  // the same chain of launches replayed from a hipGraph (stream capture): does a graph shorten the launch-to-launch time of
  // DEPENDENT kernels?
  for (int lds : {131072, 0}) for (int g : {240, 256}) {
    const int n = 400;
    hipGraph_t graph; hipGraphExec_t exec;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n; ++i) {
      if (lds) hipLaunchKernelGGL(k, dim3(g), dim3(512), lds, s, (float*)nullptr);
      else hipLaunchKernelGGL(k_small, dim3(g), dim3(256), 0, s, (float*)nullptr);
    }
    hipStreamEndCapture(s, &graph);
    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphLaunch(exec, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    hipGraphLaunch(exec, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("hipGraph of %d dependent launches, %s, grid %d: %.2f us per launch\n", n, lds ? "512 thr / 128 KiB LDS" : "256 thr / no LDS", g, ms * 1e3 / n);
    hipGraphExecDestroy(exec); hipGraphDestroy(graph);
  }
  return 0;
}

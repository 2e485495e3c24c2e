// HBM read rate vs. the ORDER in which a wave asks for its weight rows (tools/micro, not part of the library).
//   pipelined<DEP>: every step a wave asks for one 128-B line of each of its 32 rows, DEP steps ahead (the gemm_stream /
//                   tile-kernel order: a row's consecutive lines are requested one step apart)
//   burst<BUR>:     every BUR steps a wave asks for BUR consecutive lines of each row at once (double-buffered)
//   linear:         each wave streams a private contiguous region (upper bound)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef short short8_t __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int DEP>
__global__ __launch_bounds__(512) void pipelined(const short* __restrict__ B, int ldb, int nsteps, int* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const short* p0 = B + (size_t)(blockIdx.x * 256 + wave * 32 + lr) * ldb + lg * 16;
  const short* p1 = p0 + (size_t)16 * ldb;
  short8_t w[DEP][4];
  short8_t acc = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int d = 0; d < DEP; ++d) {
    w[d][0] = __builtin_nontemporal_load((const short8_t*)(p0 + d * 64));
    w[d][1] = __builtin_nontemporal_load((const short8_t*)(p0 + d * 64 + 8));
    w[d][2] = __builtin_nontemporal_load((const short8_t*)(p1 + d * 64));
    w[d][3] = __builtin_nontemporal_load((const short8_t*)(p1 + d * 64 + 8));
  }
  for (int t0 = 0; t0 < nsteps; t0 += DEP) {
#pragma unroll
    for (int d = 0; d < DEP; ++d) {
      acc ^= w[d][0] ^ w[d][1] ^ w[d][2] ^ w[d][3];
      int tn = t0 + d + DEP; tn = tn < nsteps ? tn : nsteps - 1;
      w[d][0] = __builtin_nontemporal_load((const short8_t*)(p0 + tn * 64));
      w[d][1] = __builtin_nontemporal_load((const short8_t*)(p0 + tn * 64 + 8));
      w[d][2] = __builtin_nontemporal_load((const short8_t*)(p1 + tn * 64));
      w[d][3] = __builtin_nontemporal_load((const short8_t*)(p1 + tn * 64 + 8));
      __builtin_amdgcn_s_sleep(SLEEP);
    }
  }
  if (acc[0] == 12345 && acc[3] == 77) sink[0] = 1;
}

template <int BUR>
__global__ __launch_bounds__(512) void burst(const short* __restrict__ B, int ldb, int nsteps, int* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const short* p0 = B + (size_t)(blockIdx.x * 256 + wave * 32 + lr) * ldb + lg * 16;
  const short* p1 = p0 + (size_t)16 * ldb;
  short8_t w[2][BUR][4];
  short8_t acc = {0, 0, 0, 0, 0, 0, 0, 0};
  auto load = [&](int buf, int t) {
    t = t < nsteps ? t : nsteps - BUR;
#pragma unroll
    for (int d = 0; d < BUR; ++d) {
      w[buf][d][0] = __builtin_nontemporal_load((const short8_t*)(p0 + (t + d) * 64));
      w[buf][d][1] = __builtin_nontemporal_load((const short8_t*)(p0 + (t + d) * 64 + 8));
    }
#pragma unroll
    for (int d = 0; d < BUR; ++d) {
      w[buf][d][2] = __builtin_nontemporal_load((const short8_t*)(p1 + (t + d) * 64));
      w[buf][d][3] = __builtin_nontemporal_load((const short8_t*)(p1 + (t + d) * 64 + 8));
    }
  };
  load(0, 0);
  load(1, BUR);
  for (int t0 = 0; t0 < nsteps; t0 += 2 * BUR) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int d = 0; d < BUR; ++d) {
        acc ^= w[b][d][0] ^ w[b][d][1] ^ w[b][d][2] ^ w[b][d][3];
        __builtin_amdgcn_s_sleep(SLEEP);
      }
      load(b, t0 + (b + 2) * BUR);
    }
  }
  if (acc[0] == 12345 && acc[3] == 77) sink[0] = 1;
}

__global__ __launch_bounds__(512) void linear(const short* __restrict__ B, long per_wave_elems, int* sink) {
  const int lane = threadIdx.x & 63;
  const long wv = (long)blockIdx.x * 8 + (threadIdx.x >> 6);
  const short* p = B + wv * per_wave_elems + lane * 8;
  short8_t acc = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long i = 0; i < per_wave_elems; i += 512 * 8) {
    short8_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load((const short8_t*)(p + i + u * 512));
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u];
  }
  if (acc[0] == 12345 && acc[3] == 77) sink[0] = 1;
}

// the decode gemv's order: a workgroup owns 16 rows, its 4 waves own K quarters, lane (lr, lg) reads 32 contiguous bytes of
// row lr per 64-deep step, 8 steps in flight.  PACKED: the same bytes per wave, but laid out so that every wave-instruction
// reads 1 KiB contiguous and a wave's instructions walk one contiguous region (what a pre-permuted weight copy would give).
template <bool PACKED>
__global__ __launch_bounds__(256) void gemvlike(const short* __restrict__ B, int K, int* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const int per = K / 64 / 4;                       // steps per wave
  short8_t acc = {0, 0, 0, 0, 0, 0, 0, 0};
  const short* base = B + (size_t)blockIdx.x * 16 * K;
  for (int s = 0; s < per; s += 8) {
    short8_t w0[8], w1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (PACKED) {
        const short* p = base + ((size_t)(wave * per + s + u) * 2) * 512 + lane * 8;
        w0[u] = __builtin_nontemporal_load((const short8_t*)p);
        w1[u] = __builtin_nontemporal_load((const short8_t*)(p + 512));
      } else {
        const short* p = base + (size_t)lr * K + (wave * per + s + u) * 64 + lg * 16;
        w0[u] = __builtin_nontemporal_load((const short8_t*)p);
        w1[u] = __builtin_nontemporal_load((const short8_t*)(p + 8));
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= w0[u] ^ w1[u];
  }
  if (acc[0] == 12345 && acc[3] == 77) sink[0] = 1;
}

int main() {
  const int K = 4096, nsteps = K / 64, NB = 256, N = NB * 256;
  const size_t bytes = (size_t)N * K * 2;           // 512 MiB per matrix
  const int NMAT = 3;
  short* B[NMAT]; int* sink;
  for (int i = 0; i < NMAT; ++i) { CHECK(hipMalloc(&B[i], bytes)); CHECK(hipMemset(B[i], i + 1, bytes)); }
  CHECK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < NMAT; ++i) launch(B[i]);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms / NMAT < best) best = ms / NMAT;
    }
    printf("%-28s %8.1f us  %.2f TB/s\n", name, best * 1e3, bytes / (best * 1e-3) / 1e12);
  };
  time("linear", [&](short* b) { hipLaunchKernelGGL(linear, dim3(NB), dim3(512), 0, 0, b, (long)(bytes / 2 / (NB * 8)), sink); });
  time("pipelined<3>", [&](short* b) { hipLaunchKernelGGL(pipelined<3>, dim3(NB), dim3(512), 0, 0, b, K, nsteps, sink); });
  time("pipelined<6>", [&](short* b) { hipLaunchKernelGGL(pipelined<6>, dim3(NB), dim3(512), 0, 0, b, K, nsteps, sink); });
  time("burst<2> (2x2 steps)", [&](short* b) { hipLaunchKernelGGL(burst<2>, dim3(NB), dim3(512), 0, 0, b, K, nsteps, sink); });
  time("burst<4> (2x4 steps)", [&](short* b) { hipLaunchKernelGGL(burst<4>, dim3(NB), dim3(512), 0, 0, b, K, nsteps, sink); });
  time("burst<8> (2x8 steps)", [&](short* b) { hipLaunchKernelGGL(burst<8>, dim3(NB), dim3(512), 0, 0, b, K, nsteps, sink); });
  time("gemv order (16 rows/WG)", [&](short* b) { hipLaunchKernelGGL(gemvlike<false>, dim3(N / 16), dim3(256), 0, 0, b, K, sink); });
  time("gemv order, packed copy", [&](short* b) { hipLaunchKernelGGL(gemvlike<true>, dim3(N / 16), dim3(256), 0, 0, b, K, sink); });
  return 0;
}

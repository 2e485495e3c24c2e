// Launch profiler (measurement hook of bench.py): HIP events recorded on the LAUNCH stream directly around
// single kernel launches of the GEMM kernels, so a launch's own duration is known whatever else the caller's op does
// around it (split-K launches: the partial-product kernel alone, without the launch that sums its slabs).  Off unless
// mh_prof_start() was called; launches inside a stream capture are not recorded (an event record would become a graph
// node).  Not thread-safe: one host thread drives the launches it profiles.
#include "common.h"
#include <cstdlib>

struct MhProfRec { hipEvent_t e0, e1; int meta[6]; };   // kernel id (mh_gemm_plan numbering), M, N, K, splits, flags
static MhProfRec* g_rec = nullptr;
static int g_cap = 0, g_n = 0, g_open = -1;
bool g_mh_prof_on = false;
static float g_overhead_ms = 0.f;
extern "C" double mh_prof_overhead_ms(void) { return (double)g_overhead_ms; }

// A named no-op: the two ends of the profiled region in a rocprofv3 kernel trace of the same run (tools/rocpd_step.py)
__global__ void mh_prof_marker_kernel(int which) { (void)which; }

extern "C" int mh_prof_start(int capacity, hipStream_t stream) {
  if (capacity <= 0) return MH_ERR_ARG;
  if (g_cap < capacity) {
    MhProfRec* r = (MhProfRec*)realloc(g_rec, sizeof(MhProfRec) * (size_t)capacity);
    if (!r) return MH_ERR_ARG;
    g_rec = r;
    for (int i = g_cap; i < capacity; ++i) {
      if (hipEventCreate(&g_rec[i].e0) != hipSuccess || hipEventCreate(&g_rec[i].e1) != hipSuccess) return MH_ERR_LAUNCH;
    }
    g_cap = capacity;
  }
  g_n = 0;
  g_open = -1;
  // calibration: an event pair with NOTHING between its two records still measures the queue's marker-to-marker time; the
  // median of 33 such pairs is what every bracketed launch carries on top of the kernel's own duration (mh_prof_overhead_ms)
  {
    hipEvent_t c0[33], c1[33];
    float t[33];
    bool ok = true;
    for (int i = 0; i < 33 && ok; ++i) ok = hipEventCreate(&c0[i]) == hipSuccess && hipEventCreate(&c1[i]) == hipSuccess;
    for (int i = 0; i < 33 && ok; ++i) ok = hipEventRecord(c0[i], stream) == hipSuccess && hipEventRecord(c1[i], stream) == hipSuccess;
    if (ok) ok = hipStreamSynchronize(stream) == hipSuccess;
    for (int i = 0; i < 33 && ok; ++i) ok = hipEventElapsedTime(&t[i], c0[i], c1[i]) == hipSuccess;
    if (ok) {
      for (int i = 1; i < 33; ++i) { float v = t[i]; int j = i - 1; while (j >= 0 && t[j] > v) { t[j + 1] = t[j]; --j; } t[j + 1] = v; }
      g_overhead_ms = t[16];
    }
    for (int i = 0; i < 33; ++i) { (void)hipEventDestroy(c0[i]); (void)hipEventDestroy(c1[i]); }
  }
  hipLaunchKernelGGL(mh_prof_marker_kernel, dim3(1), dim3(64), 0, stream, 0);
  MH_CHECK_LAUNCH();
  g_mh_prof_on = true;
  return MH_OK;
}

void mh_prof_pre(hipStream_t s, int kernel, int M, int N, int K, int splits, int flags) {
  g_open = -1;
  if (!g_mh_prof_on || g_n >= g_cap) return;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return;
  MhProfRec& r = g_rec[g_n];
  r.meta[0] = kernel; r.meta[1] = M; r.meta[2] = N; r.meta[3] = K; r.meta[4] = splits; r.meta[5] = flags;
  if (hipEventRecord(r.e0, s) != hipSuccess) return;
  g_open = g_n;
}

void mh_prof_post(hipStream_t s) {
  if (g_open < 0) return;
  if (hipEventRecord(g_rec[g_open].e1, s) == hipSuccess) ++g_n;
  g_open = -1;
}

// Stops recording, waits for the device, writes up to `cap` records: meta[i*6 .. i*6+5] and ms[i].  Returns the count.
extern "C" int mh_prof_stop(int* meta, float* ms, int cap, hipStream_t stream) {
  g_mh_prof_on = false;
  hipLaunchKernelGGL(mh_prof_marker_kernel, dim3(1), dim3(64), 0, stream, 1);
  if (hipDeviceSynchronize() != hipSuccess) return MH_ERR_LAUNCH;
  int n = g_n < cap ? g_n : cap;
  for (int i = 0; i < n; ++i) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_rec[i].e0, g_rec[i].e1) != hipSuccess) return MH_ERR_LAUNCH;
    ms[i] = t;
    for (int j = 0; j < 6; ++j) meta[i * 6 + j] = g_rec[i].meta[j];
  }
  g_n = 0;
  return n;
}

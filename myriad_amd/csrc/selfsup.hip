// K18 (data path): the image half of the NSA / CutPaste self-supervised anomaly augmentation on decoded
// uint8 crops resident in HBM -- reference minigpt4/datasets/self_sup_tasks.py:11-113 (label), :254-268 (arithmetic blends).
// The patch geometry (np.random draws, object-mask search loops) stays on the host (myriad_amd/self_sup.py) exactly as the
// reference samples it; what the host hands over per image is a short list of patch operations.  Integer / byte work,
// HBM-bound and tiny (a 224 x 224 x 3 crop is 150 KB): one launch per stage over the whole batch.
//
//   patch_blend:  for each operation (applied in order): dest box (y0, x0, h, w), source top-left (sy, sx), a byte mask
//                 [h][w] and the interpolation factor.  'swap' (:258-262):  out = m ? src : out.   'uniform' (:263-268):
//                 out = uint8(floor(x - (f m) x + (f m) s)) in float64, the reference's expression order.
//   label:        label_mask = medianBlur5(mean_c |M dest - M out| > tol)  (:97-98; M = union patch mask; the mean over 3
//                 channels compares as  sum > 3 tol),  intensity = median_disk5(mean_c |lm dest - lm out|) (:103-105; the
//                 median of 81 neighbours taken on the integer channel sums), logistic (:106-107), binary, continuous (:100-101).
#include "common.h"

struct PatchOp {          // 48 bytes; device array [n_ops]
  int image, y0, x0, h, w, sy, sx, mode;    // mode 0 = swap, 1 = uniform
  long mask_off;                            // offset of this operation's [h][w] byte mask in the mask pool
  double factor;
};

// one workgroup column per operation is wasteful for tiny patches; operations of one image must apply in order, so the
// grid is (pixels of the largest patch, 1, 1) and every thread walks the operation list (<= a few per image, sorted by image)
__global__ void patch_blend_kernel(unsigned char* __restrict__ out, const unsigned char* __restrict__ src,
                                   const unsigned char* __restrict__ masks, const PatchOp* __restrict__ ops, int n_ops, int H,
                                   int W, unsigned char* __restrict__ union_mask) {
  for (int o = 0; o < n_ops; ++o) {
    const PatchOp op = ops[o];
    const long npx = (long)op.h * op.w;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npx; i += (long)gridDim.x * blockDim.x) {
      const int py = (int)(i / op.w), px = (int)(i - (long)py * op.w);
      const unsigned char m = masks[op.mask_off + i];
      const long d = ((long)op.image * H + op.y0 + py) * W + op.x0 + px;
      const long s = ((long)op.image * H + op.sy + py) * W + op.sx + px;
      if (union_mask) union_mask[d] = m;                 // mask[a1:b1, a2:b2] = patch_mask (assignment, :88)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const unsigned char x = out[d * 3 + c], sv = src[s * 3 + c];
        if (op.mode == 0) {
          out[d * 3 + c] = m ? sv : x;
        } else {
          const double fm = op.factor * (double)m;
          double v = (double)x;
          v -= fm * (double)x;
          v += fm * (double)sv;
          out[d * 3 + c] = (unsigned char)floor(v);
        }
      }
    }
    // operations may overlap: the next one must see this one's pixels.  One launch per operation keeps that order without a
    // grid barrier -- the host loops (mh_patch_blend_u8), n_ops here is always 1.
  }
}

__global__ void label_sum_kernel(const unsigned char* __restrict__ dest, const unsigned char* __restrict__ out,
                                 const unsigned char* __restrict__ m, int* __restrict__ sums, long npx) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npx; i += (long)gridDim.x * blockDim.x) {
    int s = 0;
    if (m[i]) {
#pragma unroll
      for (int c = 0; c < 3; ++c) s += abs((int)dest[i * 3 + c] - (int)out[i * 3 + c]);
    }
    sums[i] = s;
  }
}

// binary 5 x 5 median with replicated borders of (sums > 3 tol): majority of 25
__global__ void label_mask_kernel(const int* __restrict__ sums, unsigned char* __restrict__ lm, int B, int H, int W, int tol3) {
  const long npx = (long)B * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npx; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long base = (i / ((long)W * H)) * (long)W * H;
    int cnt = 0;
    for (int dy = -2; dy <= 2; ++dy) {
      const int yy = min(max(y + dy, 0), H - 1);
      for (int dx = -2; dx <= 2; ++dx) {
        const int xx = min(max(x + dx, 0), W - 1);
        cnt += sums[base + (long)yy * W + xx] > tol3;
      }
    }
    lm[i] = cnt >= 13;
  }
}

// mode 0 binary: label = lm.  1 continuous: lm * factor[image].  2 intensity / 3 logistic-intensity: median over the radius-5
// disk (81 taps, replicated borders) of v = lm * sum_c |dest - out| (integers; the mean is v / 3), then for 3 the logistic map
__global__ void label_value_kernel(const int* __restrict__ sums, const unsigned char* __restrict__ lm, float* __restrict__ label,
                                   const double* __restrict__ factor, int B, int H, int W, int mode, double k, double x0) {
  const long npx = (long)B * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npx; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long img = i / ((long)W * H), base = img * (long)W * H;
    if (mode == 0) { label[i] = (float)lm[i]; continue; }
    if (mode == 1) { label[i] = (float)((double)lm[i] * factor[img]); continue; }
    int v[81];
    int n = 0;
    for (int dy = -5; dy <= 5; ++dy) {
      const int yy = min(max(y + dy, 0), H - 1);
      for (int dx = -5; dx <= 5; ++dx) {
        if (dx * dx + dy * dy > 25) continue;
        const int xx = min(max(x + dx, 0), W - 1);
        const long j = base + (long)yy * W + xx;
        v[n++] = lm[j] ? sums[j] : 0;
      }
    }
    // 41st smallest of 81: most windows are all zero -- count first, select only when needed
    int nz = 0;
    for (int t = 0; t < 81; ++t) nz += v[t] != 0;
    int med = 0;
    if (nz > 40) {
      for (int a = 0; a <= 40; ++a) {                    // partial selection sort up to the median position
        int mi = a;
        for (int t = a + 1; t < 81; ++t)
          if (v[t] < v[mi]) mi = t;
        const int tmp = v[a]; v[a] = v[mi]; v[mi] = tmp;
      }
      med = v[40];
    }
    const double val = (double)med / 3.0;
    if (mode == 2) label[i] = (float)val;
    else label[i] = (float)((double)lm[i] / (1.0 + exp(-k * (val - x0))));
  }
}

static inline int ss_grid(long n) {
  long g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

// out [B,H,W,3] u8 (starts as a copy of dest), src [B,H,W,3] u8, ops: device array of n_ops PatchOp (48 bytes each, see
// myriad_amd/self_sup.py for the packing), masks: device byte pool, union_mask [B,H,W] u8 (zero-filled by the caller) or NULL.
// hs/ws: host copies of every operation's h and w (grid sizing); operations run in order, one launch each.
extern "C" int mh_patch_blend_u8(void* out, const void* src, const void* masks, const void* ops, const int* hs, const int* ws,
                                 int n_ops, int B, int H, int W, void* union_mask, hipStream_t stream) {
  if (n_ops <= 0) return MH_OK;
  if (!out || !src || !masks || !ops || !hs || !ws || B <= 0 || H <= 0 || W <= 0) return MH_ERR_ARG;
  for (int o = 0; o < n_ops; ++o) {
    if (hs[o] <= 0 || ws[o] <= 0) continue;
    hipLaunchKernelGGL(patch_blend_kernel, dim3(ss_grid((long)hs[o] * ws[o])), dim3(256), 0, stream, (unsigned char*)out,
                       (const unsigned char*)src, (const unsigned char*)masks, (const PatchOp*)ops + o, 1, H, W,
                       (unsigned char*)union_mask);
  }
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// label [B,H,W] f32 from dest / out / union_mask; sums_ws [B,H,W] int32 and lm_ws [B,H,W] u8 are scratch.  mode: 0 binary,
// 1 continuous (factor [B] f64 device), 2 intensity, 3 logistic-intensity (k, x0)
extern "C" int mh_patch_label(const void* dest, const void* out, const void* union_mask, void* sums_ws, void* lm_ws, float* label,
                              const double* factor, int B, int H, int W, int mode, int tol, double k, double x0,
                              hipStream_t stream) {
  if (B <= 0) return MH_OK;
  if (!dest || !out || !union_mask || !sums_ws || !lm_ws || !label || mode < 0 || mode > 3 || (mode == 1 && !factor))
    return MH_ERR_ARG;
  const long npx = (long)B * H * W;
  hipLaunchKernelGGL(label_sum_kernel, dim3(ss_grid(npx)), dim3(256), 0, stream, (const unsigned char*)dest,
                     (const unsigned char*)out, (const unsigned char*)union_mask, (int*)sums_ws, npx);
  hipLaunchKernelGGL(label_mask_kernel, dim3(ss_grid(npx)), dim3(256), 0, stream, (const int*)sums_ws, (unsigned char*)lm_ws, B, H,
                     W, 3 * tol);
  hipLaunchKernelGGL(label_value_kernel, dim3(ss_grid(npx)), dim3(256), 0, stream, (const int*)sums_ws, (const unsigned char*)lm_ws,
                     label, factor, B, H, W, mode, k, x0);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

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

struct PatchOp {          // 56 bytes; device array [n_ops]
  int image, y0, x0, h, w, sy, sx, mode;    // mode & 3: 0 = swap, 1 = uniform, 2 = union mask only (the pixels were written by
                                            // the Poisson solve);  mode & 4: the source pixels are the resampled patch
                                            // [h][w][3] at patch_off of the patch pool instead of src[image] at (sy, sx)
  long mask_off;                            // offset of this operation's [h][w] byte mask in the mask pool
  double factor;
  long patch_off;
};

// one workgroup column per operation is wasteful for tiny patches; operations of one image must apply in order, so the
// grid is (pixels of the largest patch, 1, 1) and every thread walks the operation list (<= a few per image, sorted by image)
__global__ void patch_blend_kernel(unsigned char* __restrict__ out, const unsigned char* __restrict__ src,
                                   const unsigned char* __restrict__ masks, const PatchOp* __restrict__ ops, int n_ops, int H,
                                   int W, unsigned char* __restrict__ union_mask, const unsigned char* __restrict__ patches) {
  for (int o = 0; o < n_ops; ++o) {
    const PatchOp op = ops[o];
    const long npx = (long)op.h * op.w;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npx; i += (long)gridDim.x * blockDim.x) {
      const int py = (int)(i / op.w), px = (int)(i - (long)py * op.w);
      const unsigned char m = masks[op.mask_off + i];
      const long d = ((long)op.image * H + op.y0 + py) * W + op.x0 + px;
      const long s = ((long)op.image * H + op.sy + py) * W + op.sx + px;
      if (union_mask) union_mask[d] = m;                 // mask[a1:b1, a2:b2] = patch_mask (assignment, :88)
      if ((op.mode & 3) == 2) continue;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const unsigned char x = out[d * 3 + c], sv = (op.mode & 4) ? patches[op.patch_off + i * 3 + c] : src[s * 3 + c];
        if ((op.mode & 3) == 0) {
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
  // low half: the channel sum of |dest - out| under the union mask (what label_mask thresholds, self_sup_tasks.py:97); high half:
  // the same sum WITHOUT the mask -- the intensity label (:101-103) multiplies by label_mask, the median-blurred mask, which can
  // reach pixels the union mask leaves out (holes of a background-trimmed patch that a Poisson clone nevertheless changed).
  // Round 4 fix: both stages used the masked sum; the 'mix_b' golden (clone + skip_background, no resize) is the case that
  // tells them apart.
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npx; i += (long)gridDim.x * blockDim.x) {
    int s = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) s += abs((int)dest[i * 3 + c] - (int)out[i * 3 + c]);
    sums[i] = (m[i] ? s : 0) | (s << 16);
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
        cnt += (sums[base + (long)yy * W + xx] & 0xFFFF) > tol3;
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
        v[n++] = lm[j] ? (sums[j] >> 16) : 0;
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
                                 int n_ops, int B, int H, int W, void* union_mask, const void* patches, hipStream_t stream) {
  if (n_ops <= 0) return MH_OK;
  if (!out || !src || !masks || !ops || !hs || !ws || B <= 0 || H <= 0 || W <= 0) return MH_ERR_ARG;
  for (int o = 0; o < n_ops; ++o) {
    if (hs[o] <= 0 || ws[o] <= 0) continue;
    hipLaunchKernelGGL(patch_blend_kernel, dim3(ss_grid((long)hs[o] * ws[o])), dim3(256), 0, stream, (unsigned char*)out,
                       (const unsigned char*)src, (const unsigned char*)masks, (const PatchOp*)ops + o, 1, H, W,
                       (unsigned char*)union_mask, (const unsigned char*)patches);
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


// ---- the two OpenCV steps of the shipped recipe (self_sup_tasks.py:213-227 cv2.resize, :269-288 cv2.seamlessClone), built
// from their published algorithms; PARITY UNPINNED (OpenCV is absent: myriad_amd/self_sup.py) ------------------------
//
// patch_resize: 8-bit INTER_LINEAR of resize.cpp.  The host builds, per axis, the left source index and the two 11-bit
// weights (linear_resize_tables); horizontal pass in int32, vertical pass ((b0 (S0 >> 4)) >> 16) + ((b1 (S1 >> 4)) >> 16) + 2
// >> 2.  area2: the exact 2 x 2 decimation OpenCV hands to INTER_AREA (rounded mean of 4).  One thread per output pixel.
__global__ void patch_resize_kernel(const unsigned char* __restrict__ src, int H, int W, int image, int sy, int sx, int sh, int sw,
                                    const int* __restrict__ xi, const int* __restrict__ xw, const int* __restrict__ yi,
                                    const int* __restrict__ yw, int area2, unsigned char* __restrict__ out, int h, int w) {
  const long npx = (long)h * w;
  const unsigned char* base = src + ((long)image * H + sy) * W * 3 + (long)sx * 3;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npx; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i / w), x = (int)(i - (long)y * w);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int v;
      if (area2) {
        const unsigned char* p = base + ((long)(2 * y) * W + 2 * x) * 3 + c;
        v = ((int)p[0] + (int)p[3] + (int)p[(long)W * 3] + (int)p[(long)W * 3 + 3] + 2) >> 2;
      } else {
        const int x0 = xi[x], x1 = min(x0 + 1, sw - 1), y0 = yi[y], y1 = min(y0 + 1, sh - 1);
        const int a0 = xw[2 * x], a1 = xw[2 * x + 1], b0 = yw[2 * y], b1 = yw[2 * y + 1];
        const int r0 = (int)base[((long)y0 * W + x0) * 3 + c] * a0 + (int)base[((long)y0 * W + x1) * 3 + c] * a1;
        const int r1 = (int)base[((long)y1 * W + x0) * 3 + c] * a0 + (int)base[((long)y1 * W + x1) * 3 + c] * a1;
        v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
      }
      out[i * 3 + c] = (unsigned char)v;
    }
  }
}

extern "C" int mh_patch_resize_u8(const void* src, int image, int H, int W, int sy, int sx, int sh, int sw, const int* xi,
                                  const int* xw, const int* yi, const int* yw, void* out, int h, int w, hipStream_t stream) {
  if (h <= 0 || w <= 0) return MH_OK;
  if (!src || !out || sh <= 0 || sw <= 0 || sy < 0 || sx < 0 || sy + sh > H || sx + sw > W) return MH_ERR_ARG;
  const int area2 = (sw == 2 * w && sh == 2 * h) ? 1 : 0;
  if (!area2 && (!xi || !xw || !yi || !yw)) return MH_ERR_ARG;
  hipLaunchKernelGGL(patch_resize_kernel, dim3(ss_grid((long)h * w)), dim3(256), 0, stream, (const unsigned char*)src, H, W, image,
                     sy, sx, sh, sw, xi, xw, yi, yw, area2, (unsigned char*)out, h, w);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// Poisson cloning of one patch, Cloning::normalClone with NORMAL_CLONE (seamless_cloning_impl.cpp), float64.
//   ROI: h x w, at (y0s, x0s) of the patch / its mask and at (dy0, dx0) of the destination crop (the bounding box of the
//   non-zero mask; myriad_amd/self_sup.clone_roi).  D = destination ROI, P = the patch where the mask is set (0 elsewhere),
//   e = the mask eroded 3 x (host), m = e / 255.
//   rhs(y, x) = div( (1 - m) grad D + m grad P ) - Laplacian(boundary ring of D)         on the (h-2) x (w-2) interior
//   (mixed != 0, cv2.MIXED_CLONE: grad P is replaced element-wise by grad D where |Px - Py| <= |Dx - Dy|)
//   u = DST^-1( DST(rhs) / (2 cos(pi (i+1)/(w-1)) + 2 cos(pi (j+1)/(h-1)) - 4) )         (Cloning::solve, as dense sine products)
//   out = floor(clamp(u, 0, 255) + 1e-6) on the interior; the ring keeps the destination's pixels.
__global__ void poisson_rhs_kernel(const unsigned char* __restrict__ img, int H, int W, int image, const unsigned char* __restrict__ patch,
                                   int wp, const unsigned char* __restrict__ pms, const unsigned char* __restrict__ er, int y0s,
                                   int x0s, int dy0, int dx0, int h, int w, double* __restrict__ rhs, int mixed) {
  const int nh = h - 2, nw = w - 2;
  const long n = (long)nh * nw;
  const unsigned char* D0 = img + ((long)image * H + dy0) * W * 3 + (long)dx0 * 3;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i / nw) + 1, x = (int)(i % nw) + 1;
    auto mfv = [&](int yy, int xx) { return (double)er[(long)yy * w + xx] / 255.0; };
    auto miv = [&](int yy, int xx) { return (255.0 - (double)er[(long)yy * w + xx]) / 255.0; };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      auto Dv = [&](int yy, int xx) { return (double)D0[((long)yy * W + xx) * 3 + c]; };
      auto Pv = [&](int yy, int xx) {
        const long q = (long)(y0s + yy) * wp + (x0s + xx);
        return pms[q] ? (double)patch[q * 3 + c] : 0.0;
      };
      // MIXED_CLONE (Cloning::mixedClone): the patch's gradient pair at an element is kept where |Px - Py| > |Dx - Dy|, else the
      // destination's pair stands in for it (both components switch together)
      auto keep = [&](int yy, int xx) {
        if (!mixed) return true;
        const double px = Pv(yy, xx + 1) - Pv(yy, xx), py = Pv(yy + 1, xx) - Pv(yy, xx);
        const double dx = Dv(yy, xx + 1) - Dv(yy, xx), dy = Dv(yy + 1, xx) - Dv(yy, xx);
        return fabs(px - py) > fabs(dx - dy);
      };
      auto gx = [&](int yy, int xx) {
        const double d = Dv(yy, xx + 1) - Dv(yy, xx);
        const double pg = keep(yy, xx) ? Pv(yy, xx + 1) - Pv(yy, xx) : d;
        return d * miv(yy, xx) + pg * mfv(yy, xx);
      };
      auto gy = [&](int yy, int xx) {
        const double d = Dv(yy + 1, xx) - Dv(yy, xx);
        const double pg = keep(yy, xx) ? Pv(yy + 1, xx) - Pv(yy, xx) : d;
        return d * miv(yy, xx) + pg * mfv(yy, xx);
      };
      const double lap = (gx(y, x) - gx(y, x - 1)) + (gy(y, x) - gy(y - 1, x));
      double ring = 0.0;                                           // 4-neighbour Laplacian of the ROI's boundary ring
      if (x - 1 == 0) ring += Dv(y, 0);
      if (x + 1 == w - 1) ring += Dv(y, w - 1);
      if (y - 1 == 0) ring += Dv(0, x);
      if (y + 1 == h - 1) ring += Dv(h - 1, x);
      rhs[(long)c * n + i] = lap - ring;
    }
  }
}

// C[b] = scale * (A[b] . B[b]) / (rowv[i] + colv[j] + c0)   (the divisor only when rowv != NULL); row-major, f64, 16 x 16 tiles
__global__ void dmatmul_kernel(const double* __restrict__ A, int lda, long sA, const double* __restrict__ Bm, int ldb, long sB,
                               double* __restrict__ C, int ldc, long sC, int M, int N, int K, const double* __restrict__ rowv,
                               const double* __restrict__ colv, double c0, double scale) {
  __shared__ double ta[16][17], tb[16][17];
  const int b = blockIdx.z;
  A += b * sA; Bm += b * sB; C += b * sC;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int row = blockIdx.y * 16 + ty, col = blockIdx.x * 16 + tx;
  double acc = 0.0;
  for (int k0 = 0; k0 < K; k0 += 16) {
    ta[ty][tx] = (row < M && k0 + tx < K) ? A[(long)row * lda + k0 + tx] : 0.0;
    tb[ty][tx] = (k0 + ty < K && col < N) ? Bm[(long)(k0 + ty) * ldb + col] : 0.0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += ta[ty][k] * tb[k][tx];
    __syncthreads();
  }
  if (row < M && col < N) {
    double v = acc * scale;
    if (rowv) v /= (rowv[row] + colv[col] + c0);
    C[(long)row * ldc + col] = v;
  }
}

__global__ void poisson_write_kernel(const double* __restrict__ u, unsigned char* __restrict__ img, int H, int W, int image, int dy0,
                                     int dx0, int h, int w) {
  const int nh = h - 2, nw = w - 2;
  const long n = (long)nh * nw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i / nw) + 1, x = (int)(i % nw) + 1;
    unsigned char* p = img + (((long)image * H + dy0 + y) * W + dx0 + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double v = u[(long)c * n + i];
      v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
      p[c] = (unsigned char)floor(v + 1e-6);
    }
  }
}

extern "C" long mh_patch_poisson_ws_doubles(int h, int w) { return h < 3 || w < 3 ? 0 : 6L * (h - 2) * (w - 2); }

// out [B,H,W,3] u8 is edited in place (image `image`).  patch [hp][wp][3] u8, pms [hp][wp] u8 (the mask handed to
// seamlessClone, border cleared), eroded [h][w] u8, Sh [nh][nh] / Sw [nw][nw] f64 sine matrices and cy [nh] / cx [nw] their
// 2 cos terms (nh = h - 2, nw = w - 2; myriad_amd/self_sup.dst_tables), ws: mh_patch_poisson_ws_doubles(h, w) doubles.
extern "C" int mh_patch_poisson_u8(void* out, int image, int H, int W, const void* patch, int hp, int wp, const void* pms,
                                   const void* eroded, int y0s, int x0s, int dy0, int dx0, int h, int w, const double* Sh,
                                   const double* cy, const double* Sw, const double* cx, double* ws, int mixed,
                                   hipStream_t stream) {
  if (h < 3 || w < 3) return MH_OK;
  if (!out || !patch || !pms || !eroded || !Sh || !cy || !Sw || !cx || !ws) return MH_ERR_ARG;
  if (y0s < 0 || x0s < 0 || y0s + h > hp || x0s + w > wp || dy0 < 0 || dx0 < 0 || dy0 + h > H || dx0 + w > W) return MH_ERR_ARG;
  const int nh = h - 2, nw = w - 2;
  const long n = (long)nh * nw;
  double* r = ws;                 // [3][nh][nw]
  double* t = ws + 3 * n;
  hipLaunchKernelGGL(poisson_rhs_kernel, dim3(ss_grid(n)), dim3(256), 0, stream, (const unsigned char*)out, H, W, image,
                     (const unsigned char*)patch, wp, (const unsigned char*)pms, (const unsigned char*)eroded, y0s, x0s, dy0, dx0, h,
                     w, r, mixed);
  const dim3 grid((nw + 15) / 16, (nh + 15) / 16, 3), block(256);
  // t = Sh . r ;  r = (t . Sw) / (cy[j] + cx[i] - 4) ;  t = Sh . r ;  r = (t . Sw) * 4 / ((nh + 1)(nw + 1))
  hipLaunchKernelGGL(dmatmul_kernel, grid, block, 0, stream, Sh, nh, 0L, r, nw, n, t, nw, n, nh, nw, nh, nullptr, nullptr, 0.0, 1.0);
  hipLaunchKernelGGL(dmatmul_kernel, grid, block, 0, stream, t, nw, n, Sw, nw, 0L, r, nw, n, nh, nw, nw, cy, cx, -4.0, 1.0);
  hipLaunchKernelGGL(dmatmul_kernel, grid, block, 0, stream, Sh, nh, 0L, r, nw, n, t, nw, n, nh, nw, nh, nullptr, nullptr, 0.0, 1.0);
  hipLaunchKernelGGL(dmatmul_kernel, grid, block, 0, stream, t, nw, n, Sw, nw, 0L, r, nw, n, nh, nw, nw, nullptr, nullptr, 0.0,
                     4.0 / ((double)(nh + 1) * (double)(nw + 1)));
  hipLaunchKernelGGL(poisson_write_kernel, dim3(ss_grid(n)), dim3(256), 0, stream, r, (unsigned char*)out, H, W, image, dy0, dx0, h, w);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// K17: image front-end of the data path (image side) -- HBM-bound byte work, bit-exact with Pillow.
// reference: datasets/datasets/anomaly_detection.py:118-122,246 (torchvision Resize(224, BICUBIC) + CenterCrop(224) on the
// PIL image), processors/blip_processors.py:21-29,120-147,189-203 (ToTensor + Normalize; eval: Resize((224, 224))).
// torchvision's Resize on a PIL image IS PIL.Image.resize, i.e. Pillow's ImagingResample for 8-bit channels:
//   pass 1 (horizontal) out1[y][x][c] = clip8((2^21 + sum_t kh[x][t] * in[y][bh[x].first + t][c]) >> 22)   -> uint8
//   pass 2 (vertical)   out2[y][x][c] = clip8((2^21 + sum_t kv[y][t] * out1[bv[y].first + t][x][c]) >> 22) -> uint8
// with 22-bit fixed-point weights of the support-scaled cubic (computed on the host in double precision exactly as
// Resample.c does: myriad_amd/image_frontend.py, restated in oracle/image_ref.py).  Only the pixels the centre crop keeps are
// computed: pass 1 for the crop's columns and the input rows its output rows tap, pass 2 for the crop window, which also
// applies ToTensor + Normalize as a 3 x 256 float table (((v / 255) - mean) / std evaluated in float32 on the host, so the
// float32 output is bit-identical to torch's) and writes planar [3, S, S].
// Integer arithmetic end to end: |sum| < 2^8 * 2^22 * (sum |k|) fits int32 only for small tap counts (Pillow accumulates in
// int as well and relies on the same bound: sum |k| <= ~1.3 for the cubic), so int32 it is.
#include "common.h"

#define IMG_PB 22

__device__ __forceinline__ unsigned char clip8_fx(int v) {
  v >>= IMG_PB;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// tmp[r][x][c], r = input row y0 + r (rows the crop's output rows tap), x = crop column
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ in, long row_stride, int y0, int rows,
                                                         const int* __restrict__ kh, const int* __restrict__ bh, int ksz,
                                                         int crop_x0, int out_w, unsigned char* __restrict__ tmp) {
  const long total = (long)rows * out_w;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / out_w), x = (int)(i - (long)r * out_w);
    const int xx = crop_x0 + x;
    const int first = bh[2 * xx], n = bh[2 * xx + 1];
    const int* k = kh + (long)xx * ksz;
    const unsigned char* p = in + (long)(y0 + r) * row_stride + (long)first * 3;
    int s0 = 1 << (IMG_PB - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < n; ++t) {
      const int w = k[t];
      s0 += w * p[3 * t];
      s1 += w * p[3 * t + 1];
      s2 += w * p[3 * t + 2];
    }
    unsigned char* o = tmp + i * 3;
    o[0] = clip8_fx(s0); o[1] = clip8_fx(s1); o[2] = clip8_fx(s2);
  }
}

// out[c][y][x] = lut[c][pass-2 value]; u8_out (optional) [S_h][S_w][3] is the uint8 crop the NSA augmentation works on
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const unsigned char* __restrict__ tmp, int y0, int out_w,
                                                              const int* __restrict__ kv, const int* __restrict__ bv, int ksz,
                                                              int crop_y0, int out_h, const float* __restrict__ lut,
                                                              float* __restrict__ out, unsigned char* __restrict__ u8_out) {
  const long total = (long)out_h * out_w;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i / out_w), x = (int)(i - (long)y * out_w);
    const int yy = crop_y0 + y;
    const int first = bv[2 * yy], n = bv[2 * yy + 1];
    const int* k = kv + (long)yy * ksz;
    const unsigned char* p = tmp + ((long)(first - y0) * out_w + x) * 3;
    int s0 = 1 << (IMG_PB - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < n; ++t) {
      const int w = k[t];
      const unsigned char* q = p + (long)t * out_w * 3;
      s0 += w * q[0];
      s1 += w * q[1];
      s2 += w * q[2];
    }
    const unsigned char v0 = clip8_fx(s0), v1 = clip8_fx(s1), v2 = clip8_fx(s2);
    if (out) {
      out[i] = lut[v0];
      out[total + i] = lut[256 + v1];
      out[2 * total + i] = lut[512 + v2];
    }
    if (u8_out) {
      u8_out[i * 3] = v0; u8_out[i * 3 + 1] = v1; u8_out[i * 3 + 2] = v2;
    }
  }
}

// uint8 HWC [S_h][S_w][3] -> normalised planar float (the second half of the training path: after the augmentation)
__global__ __launch_bounds__(256) void u8_normalize_kernel(const unsigned char* __restrict__ u8, long n_pix,
                                                           const float* __restrict__ lut, float* __restrict__ out) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_pix; i += (long)gridDim.x * blockDim.x) {
    out[i] = lut[u8[i * 3]];
    out[n_pix + i] = lut[256 + u8[i * 3 + 1]];
    out[2 * n_pix + i] = lut[512 + u8[i * 3 + 2]];
  }
}

// One image: img [H][row_stride] uint8 RGB interleaved (device) -> out [3][out_h][out_w] float32 and/or u8_out
// [out_h][out_w][3].  kh/bh, kv/bv: the fixed-point weight and bounds tables of the full resize (device, built by
// myriad_amd/image_frontend.py); (crop_y0, crop_x0): the crop window inside the resized image; (y0, rows): the input rows the
// window's output rows tap (bv[crop_y0].first .. bv[crop_y0+out_h-1].first + count); tmp: rows * out_w * 3 bytes.
extern "C" int mh_image_resize_crop_norm(const void* img, int H, int W, long row_stride, const int* kh, const int* bh, int ksz_h,
                                         const int* kv, const int* bv, int ksz_v, int crop_y0, int crop_x0, int out_h, int out_w,
                                         int y0, int rows, void* tmp, const float* lut, float* out, void* u8_out,
                                         hipStream_t stream) {
  if (!img || !kh || !bh || !kv || !bv || !tmp || (!out && !u8_out) || (out && !lut)) return MH_ERR_ARG;
  if (H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0 || crop_y0 < 0 || crop_x0 < 0 || ksz_h <= 0 || ksz_v <= 0 || y0 < 0 ||
      rows <= 0 || y0 + rows > H || row_stride < (long)W * 3)
    return MH_ERR_ARG;
  long n1 = ((long)rows * out_w + 255) / 256, n2 = ((long)out_h * out_w + 255) / 256;
  if (n1 > 65536) n1 = 65536;
  if (n2 > 65536) n2 = 65536;
  hipLaunchKernelGGL(resample_h_kernel, dim3((int)n1), dim3(256), 0, stream, (const unsigned char*)img, row_stride, y0, rows, kh, bh,
                     ksz_h, crop_x0, out_w, (unsigned char*)tmp);
  MH_CHECK_LAUNCH();
  hipLaunchKernelGGL(resample_v_norm_kernel, dim3((int)n2), dim3(256), 0, stream, (const unsigned char*)tmp, y0, out_w, kv, bv,
                     ksz_v, crop_y0, out_h, lut, out, (unsigned char*)u8_out);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

// ToTensor + Normalize of an (augmented) uint8 HWC crop: [n_pix][3] -> [3][n_pix] float32 through the 3 x 256 table
extern "C" int mh_image_u8_normalize(const void* u8_hwc, long n_pix, const float* lut, float* out, hipStream_t stream) {
  if (n_pix <= 0) return MH_OK;
  if (!u8_hwc || !lut || !out) return MH_ERR_ARG;
  long n = (n_pix + 255) / 256;
  if (n > 65536) n = 65536;
  hipLaunchKernelGGL(u8_normalize_kernel, dim3((int)n), dim3(256), 0, stream, (const unsigned char*)u8_hwc, n_pix, lut, out);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

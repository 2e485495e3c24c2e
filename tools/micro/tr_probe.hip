// Probe of ds_read_b64_tr_b16 (gfx950): which LDS element lands in which lane/slot.  LDS holds lds[i] = i (16-bit).
// Convention A: lane i of each 16-lane group points at row (i>>2), columns 4*(i&3).. of a row-major [4][16] block
// (block g = 16-lane group g, 64 elements apart).  Convention B: every lane of a group points at its own column:
// address = block + (i&15) (the "no internal lane offset" reading).  Prints what lanes 0..63 receive.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((address_space(3))) short4_t lds_s4;
__global__ void probe(short* out, int mode, int rowstride) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  int off;
  if (mode == 0) off = g * 4 * rowstride + (i >> 2) * rowstride + (i & 3) * 4;
  else off = g * 4 * rowstride + (i & 3) * rowstride + (i >> 2) * 4;
  short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + off));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int mode = 0; mode < 2; ++mode)
    for (int rs : {16, 64}) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode, rs);
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("mode %d rowstride %d (element = row*%d + col within group block; group block starts at g*4*rs)\n", mode, rs, rs);
      for (int l = 0; l < 64; ++l) {
        printf("  lane %2d:", l);
        for (int j = 0; j < 4; ++j) { int e = h[l * 4 + j] - (l >> 4) * 4 * rs; printf(" (r%d,c%2d)", e / rs, e % rs); }
        if (l % 4 == 3) printf("\n");
      }
    }
  return 0;
}

// Shared pieces of the 256-column GEMM kernels (gemm_256.hip: skewed two-group schedule; gemm_256i.hip: interleaved
// single-stream schedule): tile constants, the conflict-free LDS image permutation, counted vmcnt waits and the
// LDS-transposed full-line epilogue.
#pragma once
#include "common.h"
#include <cstdlib>

#define G2_BM 256
#define G2_BN 256
#define G2_BK 32
#define G2_NST 4
#define G2_A_BYTES (G2_BM * G2_BK * 2)   // 16 KiB
#define G2_B_BYTES (G2_BN * G2_BK * 2)   // 16 KiB
#define G2_STAGE (G2_A_BYTES + G2_B_BYTES)

#define MH_GEMM_OUT_F32 1
#define MH_GEMM_GELU 2

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__device__ __forceinline__ int g2_perm(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // {0,2,3,1}
__device__ __forceinline__ int g2_off(int row, int chunk) { return row * 64 + ((chunk ^ g2_perm(row)) << 4); }


template <int N>
__device__ __forceinline__ void g2_wait_vm() {
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}
__device__ __forceinline__ void g2_wait_younger(int younger_tiles) {   // 4 LDS-DMA instructions per tile share
  if (younger_tiles >= 3) g2_wait_vm<12>();
  else if (younger_tiles == 2) g2_wait_vm<8>();
  else if (younger_tiles == 1) g2_wait_vm<4>();
  else g2_wait_vm<0>();
}

// Epilogue of one workgroup: acc[i][j] is the wave's 128 x 64 fp32 tile (rows grp*128 + i*16 .., columns wc*64 + j*16 ..).
// Every LDS-DMA has been retired and every fragment read is done (last barrier), so the ring is
// free: each wave transposes its 128 x 64 fp32 tile through a private 16-KiB slice, 64 rows at a time, so that the
// global stores are whole rows -- 16 lanes x 8 B (bf16) or x 16 B (fp32) = one or two full 128-B lines per row --
// instead of the MFMA layout's 32-B fragments (the store tail was ~1/3 of a K = 4096 launch).  Bias / GELU /
// residual are applied on the way out, where a lane holds 4 consecutive columns of one row.
__device__ __forceinline__ void g2_epilogue(char* smem, float4_t (&acc)[8][4], void* Cv, const float* __restrict__ bias,
                                            const float* res, int M, int N, int ldc, int ldr, int flags, float alpha,
                                            int m0, int n0, int m_end, int wave, int lane) {
  const int grp = wave >> 2, wc = wave & 3;
  const int lr = lane & 15, lg = lane >> 4;
  const bool out_f32 = flags & MH_GEMM_OUT_F32;
  const bool do_gelu = flags & MH_GEMM_GELU;
  char* ep = smem + wave * 16384;                      // [64 rows][16 chunks of 16 B], chunk ^= row & 15
  const int er = lane >> 4, ec = lane & 15;            // write-out: 4 rows per pass, lane owns columns ec*4 .. +3
  const int ncol = n0 + wc * 64 + ec * 4;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int i = h * 4 + ii;
      const int row = ii * 16 + lr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = j * 4 + lg;
        *reinterpret_cast<float4_t*>(ep + row * 256 + ((c ^ (row & 15)) << 4)) =
            (float4_t){acc[i][j][0] * alpha, acc[i][j][1] * alpha, acc[i][j][2] * alpha, acc[i][j][3] * alpha};
      }
    }
    // the slice is private to the wave: program order + the compiler's lgkmcnt wait are the only ordering needed
    if (!out_f32 && (ldc & 7) == 0) {
      // bf16 output: 16-B stores, 8 lanes cover one 128-B row, 8 rows per pass (half the store instructions of the
      // 8-B form; the store tail is issue-bound, not bandwidth-bound)
      const int r8 = lane >> 3, c8 = lane & 7;
      const int nc = n0 + wc * 64 + c8 * 8;
#pragma unroll 4
      for (int p = 0; p < 8; ++p) {
        const int row = p * 8 + r8;
        const int m = m0 + grp * 128 + h * 64 + row;
        const float4_t va = *reinterpret_cast<const float4_t*>(ep + row * 256 + (((2 * c8) ^ (row & 15)) << 4));
        const float4_t vb = *reinterpret_cast<const float4_t*>(ep + row * 256 + (((2 * c8 + 1) ^ (row & 15)) << 4));
        if (m >= m_end || nc >= N) continue;
        float v[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
        if (nc + 7 < N) {
          if (bias) {
            const float4_t b0 = *reinterpret_cast<const float4_t*>(bias + nc);
            const float4_t b1 = *reinterpret_cast<const float4_t*>(bias + nc + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
          }
          if (do_gelu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
          }
          if (res) {
            const float4_t q0 = *reinterpret_cast<const float4_t*>(res + (size_t)m * ldr + nc);
            const float4_t q1 = *reinterpret_cast<const float4_t*>(res + (size_t)m * ldr + nc + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += q0[e]; v[4 + e] += q1[e]; }
          }
          uint4 pk;
          pk.x = pack_bf2(v[0], v[1]);
          pk.y = pack_bf2(v[2], v[3]);
          pk.z = pack_bf2(v[4], v[5]);
          pk.w = pack_bf2(v[6], v[7]);
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(Cv) + (size_t)m * ldc + nc) = pk;
        } else {
          for (int e = 0; e < 8 && nc + e < N; ++e) {
            float x = v[e];
            if (bias) x += bias[nc + e];
            if (do_gelu) x = gelu_erf(x);
            if (res) x += res[(size_t)m * ldr + nc + e];
            reinterpret_cast<bf16_t*>(Cv)[(size_t)m * ldc + nc + e] = f2bf(x);
          }
        }
      }
      continue;
    }
#pragma unroll 4
    for (int p = 0; p < 16; ++p) {
      const int row = p * 4 + er;
      const int m = m0 + grp * 128 + h * 64 + row;
      const float4_t v4 = *reinterpret_cast<const float4_t*>(ep + row * 256 + ((ec ^ (row & 15)) << 4));
      if (m >= m_end || ncol >= N) continue;
      float v[4] = {v4[0], v4[1], v4[2], v4[3]};
      if (ncol + 3 < N) {
        if (bias) {
          const float4_t b4 = *reinterpret_cast<const float4_t*>(bias + ncol);
          v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3];
        }
        if (do_gelu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
        if (res) {
          const float4_t r4 = *reinterpret_cast<const float4_t*>(res + (size_t)m * ldr + ncol);
          v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3];
        }
        if (out_f32) {
          *reinterpret_cast<float4_t*>(reinterpret_cast<float*>(Cv) + (size_t)m * ldc + ncol) =
              (float4_t){v[0], v[1], v[2], v[3]};
        } else {
          uint2 pk;
          pk.x = pack_bf2(v[0], v[1]);
          pk.y = pack_bf2(v[2], v[3]);
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + (size_t)m * ldc + ncol) = pk;
        }
      } else {
        for (int e = 0; e < 4 && ncol + e < N; ++e) {
          float x = v[e];
          if (bias) x += bias[ncol + e];
          if (do_gelu) x = gelu_erf(x);
          if (res) x += res[(size_t)m * ldr + ncol + e];
          if (out_f32) reinterpret_cast<float*>(Cv)[(size_t)m * ldc + ncol + e] = x;
          else reinterpret_cast<bf16_t*>(Cv)[(size_t)m * ldc + ncol + e] = f2bf(x);
        }
      }
    }
  }
}

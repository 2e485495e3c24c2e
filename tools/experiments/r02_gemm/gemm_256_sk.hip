// EXPERIMENT (round 2, not part of the library): stream-K hybrid of gemm_256_kernel for launches with more tiles than CUs.
// Built into the library behind MYRIAD_STREAMK=1 and measured with tools/experiments/r02_gemm/streamk_check.py, then taken out:
// correct (same bits as the classic launch up to the bf16 rounding of differently ordered fp32 sums), deterministic under a
// concurrent load, and SLOWER -- 243 us against 216 us on the 430-tile gate|up shape (profiles/r02_gemm_experiments.md).
// To rebuild: paste this block into csrc/gemm_256.hip in front of `static long long* g2_trace`, and the dispatcher hook of
// streamk_dispatch.patch into csrc/gemm.hip.
// ---------------------------------------------------------------------------------------------------------------------
// Stream-K hybrid of the same tile kernel for launches with more tiles than CUs (MYRIAD_STREAMK=1, experimental): P = 256
// persistent workgroups; each computes ONE full tile (tiles 0 .. P-1 in the grouped order) and a contiguous span of the
// k-tiles of the remaining T - P tiles, T * nt / P k-tiles of work per workgroup instead of up to two whole tiles.  A span
// covers the tail (or middle) of one tile and possibly the head of the next.  A workgroup writes at most one fp32 partial
// (its first segment, when that segment does not start at k = 0) with write-through stores, then raises flags[w] = epoch;
// the workgroup that holds the HEAD of a tile owns it: after its own segment it waits for the flags of the workgroups that
// hold the rest of the tile, adds their partials in workgroup order (fixed order: deterministic) and runs the epilogue.
// Segments are processed span-first, full tile last, so a partial is produced long before its owner needs it.
// bf16 output without bias / residual / GELU only (the unsplit forward Linears this is for).
// write-through (sc0 sc1) 16-byte store / L1-bypassing (sc1) load: the partial tiles cross workgroups (MI355X_MICROARCH.md,
// inter-workgroup visibility) without an L2 write-back fence
__device__ __forceinline__ void g2_store_wt(float4_t* p, float4_t v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ float4_t g2_load_sc1(const float4_t* p) {
  float4_t v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__global__ __launch_bounds__(512, 2) void gemm_256_sk_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                             bf16_t* __restrict__ C, int M, int N, int K, int lda, int ldb,
                                                             int ldc, int tiles_m, int tiles_n, float* part, int* flags,
                                                             int epoch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int lr = lane & 15, lg = lane >> 4;
  const int P = gridDim.x, bid = blockIdx.x;
  const int w = (bid & 7) * (P >> 3) + (bid >> 3);          // consecutive w on one XCD (P % 8 == 0)
  const int T = tiles_m * tiles_n, nt_all = K / G2_BK;
  const int total = (T - P) * nt_all;                       // k-tiles of the tiles beyond the first P (fits an int by far)
  const int per = total / P, rem = total % P;
  const int beg = w * per + (w < rem ? w : rem), end = beg + per + (w < rem ? 1 : 0);
  const int per_group = 8 * tiles_n;

  int offA[8], offB[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) offA[i] = g2_off(grp * 128 + i * 16 + lr, lg);
#pragma unroll
  for (int j = 0; j < 4; ++j) offB[j] = G2_A_BYTES + g2_off(wc * 64 + j * 16 + lr, lg);

#pragma unroll 1
  for (int item = 0; item < 3; ++item) {
    int lid, kt0, nt;
    bool partial = false, owner = false;
    int seg_end = 0, tile_end = 0;
    if (item == 2) {
      lid = w; kt0 = 0; nt = nt_all;
    } else {
      const int next_tile = (beg / nt_all + 1) * nt_all;
      const int s0 = item == 0 ? beg : next_tile;
      const int e0 = item == 0 ? (end < next_tile ? end : next_tile) : end;
      lid = P + s0 / nt_all; kt0 = s0 % nt_all; nt = s0 < e0 ? e0 - s0 : 0;
      partial = kt0 != 0;
      owner = kt0 == 0 && nt < nt_all;
      seg_end = e0; tile_end = (s0 / nt_all + 1) * nt_all;
    }
    if (nt > 0) {
    const int first_m = (lid / per_group) * 8;
    const int gsz = (tiles_m - first_m) < 8 ? (tiles_m - first_m) : 8;
    const int tm = first_m + (lid % per_group) % gsz, tn = (lid % per_group) / gsz;
    const int m0 = tm * G2_BM, n0 = tn * G2_BN;
    const bf16_t* gA[2];
    const bf16_t* gB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = i * 512 + tid;
      const int row = c >> 2, lc = (c & 3) ^ g2_perm(row);
      int ra = m0 + row, rb = n0 + row;
      ra = ra < M ? ra : M - 1;
      rb = rb < N ? rb : N - 1;
      gA[i] = A + (size_t)ra * lda + lc * 8;
      gB[i] = B + (size_t)rb * ldb + lc * 8;
    }
    float4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    auto issue = [&](int t) {
      char* sA = smem + (t & 3) * G2_STAGE;
      char* sB = sA + G2_A_BYTES;
      const int k0 = (kt0 + t) * G2_BK;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(gA[i] + k0), (lds_void_t*)(sA + (i * 512 + wave * 64) * 16), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(gB[i] + k0), (lds_void_t*)(sB + (i * 512 + wave * 64) * 16), 16, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < G2_NST; ++t)
      if (t < nt) issue(t);
    g2_wait_younger(nt - 1);
    __builtin_amdgcn_s_barrier();
    short8_t af[8], bfr[4];
    auto mfma_block = [&]() {
      if (grp == 1) __builtin_amdgcn_s_setprio(3);
      else __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
    for (int t = 0; t < nt; ++t) {
      if (grp == 1 && t > 0) mfma_block();
      __builtin_amdgcn_sched_barrier(0);
      const char* st = smem + (t & 3) * G2_STAGE;
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const short8_t*>(st + offB[j]);
#pragma unroll
      for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const short8_t*>(st + offA[i]);
      if (t >= 1 && t + 3 < nt) issue(t + 3);
      __builtin_amdgcn_sched_barrier(0);
      if (grp == 0) mfma_block();
      g2_wait_younger(nt - t - 2 < 2 ? nt - t - 2 : 2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
    if (grp == 1 && nt > 0) mfma_block();

    const unsigned voff = (unsigned)tid * 16u;
    if (partial) {
      const char* myp = reinterpret_cast<const char*>(part) + (long)w * 32 * 8192;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          g2_store_wt(reinterpret_cast<float4_t*>(const_cast<char*>(myp) + (i * 4 + j) * 8192 + voff), acc[i][j]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + w, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
    if (owner) {
      int covered = seg_end;
      for (int x = w + 1; covered < tile_end && x < P; ++x) {
        if (tid == 0) {
          int spins = 0;
          while (__hip_atomic_load(flags + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch && ++spins < (1 << 22))
            __builtin_amdgcn_s_sleep(2);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        const char* xp = reinterpret_cast<const char*>(part) + (long)x * 32 * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i) {                   // four loads in flight at a time (acc must stay in registers: static indices)
          float4_t v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = g2_load_sc1(reinterpret_cast<const float4_t*>(xp + (i * 4 + j) * 8192 + voff));
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[i][j][0] += v[j][0]; acc[i][j][1] += v[j][1]; acc[i][j][2] += v[j][2]; acc[i][j][3] += v[j][3];
          }
        }
        const int xe = (x + 1) * per + ((x + 1) < rem ? (x + 1) : rem);
        covered = xe < tile_end ? xe : tile_end;
      }
    }
    // epilogue: the bf16 full-line path of the tile kernel
    char* ep = smem + wave * 16384;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int i = h * 4 + ii;
        const int row = ii * 16 + lr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = j * 4 + lg;
          *reinterpret_cast<float4_t*>(ep + row * 256 + ((c ^ (row & 15)) << 4)) = acc[i][j];
        }
      }
      const int r8 = lane >> 3, c8 = lane & 7;
      const int nc = n0 + wc * 64 + c8 * 8;
#pragma unroll 4
      for (int pp = 0; pp < 8; ++pp) {
        const int row = pp * 8 + r8;
        const int m = m0 + grp * 128 + h * 64 + row;
        const float4_t va = *reinterpret_cast<const float4_t*>(ep + row * 256 + (((2 * c8) ^ (row & 15)) << 4));
        const float4_t vb = *reinterpret_cast<const float4_t*>(ep + row * 256 + (((2 * c8 + 1) ^ (row & 15)) << 4));
        if (m >= M || nc >= N) continue;
        if (nc + 7 < N) {
          uint4 pk;
          pk.x = pack_bf2(va[0], va[1]); pk.y = pack_bf2(va[2], va[3]);
          pk.z = pack_bf2(vb[0], vb[1]); pk.w = pack_bf2(vb[2], vb[3]);
          *reinterpret_cast<uint4*>(C + (size_t)m * ldc + nc) = pk;
        } else {
          const float v[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
          for (int e = 0; e < 8 && nc + e < N; ++e) C[(size_t)m * ldc + nc + e] = f2bf(v[e]);
        }
      }
    }
    }
    __syncthreads();                                   // the ring is the next segment's staging area
    }
  }
}

// part: P x 256 KiB fp32 partial tiles, flags: P ints (both caller-provided scratch), epoch: unique per launch
int mh_launch_gemm_256_sk(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, float* part,
                          int* flags, int epoch, hipStream_t stream) {
  const int tiles_m = (M + G2_BM - 1) / G2_BM, tiles_n = (N + G2_BN - 1) / G2_BN;
  const int P = 256;
  if (tiles_m * tiles_n <= P || (ldc & 7) || (K % G2_BK) || !part || !flags) return MH_ERR_ARG;
  const size_t shmem = G2_NST * G2_STAGE;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_256_sk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_256_sk_kernel, dim3(P), dim3(512), shmem, stream, (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, M, N, K,
                     lda, ldb, ldc, tiles_m, tiles_n, part, flags, epoch);
  MH_CHECK_LAUNCH();
  return MH_OK;
}


// K5b (decode): ONE persistent launch for all decoder layers of a token step (modeling_llama.py:184-299 with the KV cache,
// :564-596 the layer loop, :66-74 the final norm, :604 lm_head) at 1-2 rows.
//
// The step of round 2 was ~290 launches, the fused step of round 3 four per layer; what they all pay is the ramp of every
// weight-streaming launch: the HBM pipe drains at each kernel boundary and refills behind the next kernel's prologue
// (5.2 of 6.3 TB/s over the GEMVs, ~0.3 ms of small kernels per token).  A launch boundary forbids the one thing that
// hides it -- asking for the NEXT matrix while the current phase finishes -- because a kernel cannot start before its
// predecessor ends.  Inside one launch it is legal: the weights do not depend on the activations.
//
//   grid  = one 8-wave workgroup per CU (all resident: the phases are separated by a grid barrier)
//   phase = P1 RMSNorm + q|k|v product, P2 rotary + KV append + attention, P3 o_proj + residual,
//           P4 RMSNorm + gate|up product, P5 SiLU gate + down projection + residual; after the last layer the final
//           RMSNorm + lm_head.  Every workgroup rebuilds the <= 2 operand rows of its phase in LDS (as gemv_pro_kernel does).
//   before every barrier each wave requests the first 16 KiB of ITS weights of the next phase into registers; the barrier
//   (release fence, one agent-scope counter, relaxed polling, acquire fence: 5-7 us, MI355X_MICROARCH.md price list) is then
//   covered by ~32 MiB of weight loads in flight chip-wide.
//
// Same arithmetic as the separate kernels: the GEMV walks the packed weight copy exactly as gemv_kernel<2> (same k per
// lane, waves reduced through LDS in wave order), the norm / gate expressions are those of rmsnorm_fwd_kernel /
// silu_mul_fwd_kernel, the attention is attn_decode_kernel's with its 16 waves emulated by 8 (each runs two of the
// original waves' key sets and partial outputs, reduced in the original order) -- logits are bit-identical.  (The rotated q is
// kept in LDS, so the q columns of the `qkv` scratch stay un-rotated here; nothing else reads them.)
#include "common.h"
#include <cstdlib>

#define MG_NW 8
#define MG_NT (MG_NW * 64)
#define MG_U 8                 // 64-deep K steps per batch of weight loads (16 KiB per wave)
#define MG_MAXM 2

struct MegaLayer {             // device table, one entry per decoder layer
  const bf16_t* wqkv;          // packed (mh_gemv_pack): [3D, D], 4 waves per column block
  const bf16_t* wo;            //                        [D, D],  8 waves per column block
  const bf16_t* wgu;           //                        [2I, D], 4
  const bf16_t* wd;            //                        [D, I],  8
  const float* ln1;
  const float* ln2;
  bf16_t* cache;               // [B][T_cap][2D]
};

struct MegaParams {
  const MegaLayer* layers;
  int n_layers, M, D, H, hd, I, V, T_cap;
  float eps, scale;
  float* h;                    // [M, D] f32: the embedded token on entry; residual stream (ping)
  float* h2;                   // [M, D] f32 (pong)
  bf16_t* qkv;                 // [M, 3D]
  bf16_t* o;                   // [M, D]
  bf16_t* gu;                  // [M, 2I]
  const float* norm;           // final RMSNorm weight
  const bf16_t* lm_head;       // packed [V, D], 4 waves per block
  float* logits;               // [M, V] f32
  const int* pos;              // [M] rotary position of the new token
  const int* pos_dev;          // [1] cache row
  const int* kvlen;            // [M] valid keys after the append
  const float* cs;
  const float* sn;
  long cache_bs;
  int ld_cache;
  unsigned* bar;               // MG_BAR_WORDS uint32: barrier counters (mg_arrive), [1] = sticky abort flag
  int dbg;                     // debug (MYRIAD_MEGA_DBG): 1 = barriers do not wait, 2 = no attention phase -- WRONG RESULTS, timing only
  long long* trace;            // debug: workgroup 0 stamps the 100 MHz counter at every phase boundary of layer 1 (NULL: off)
};
#define MG_STAMP(i) if (p.trace && li == 1 && blockIdx.x == 0 && threadIdx.x == 0) p.trace[i] = (long long)__builtin_amdgcn_s_memrealtime();

struct MgBatch { short8_t w0[MG_U], w1[MG_U]; };

// ---- grid barrier, XCD-hierarchical, in two halves ---------------------------------------------------------------------
// Workgroups are grouped by blockIdx & 7 (the dispatcher deals workgroups round-robin over the 8 XCDs, so a group shares an
// L2).  arrive: every wave drains its stores (all cross-workgroup outputs are agent-scope write-through stores: no L2
// write-back fence is needed), lane 0 counts into ITS group's counter.  The last arriver of a group counts into the top
// counter, polls it until all groups are in, then publishes the group's generation; everybody else polls that generation word
// -- a line in its own XCD's L2, so 248 of the 256 pollers never touch the fabric.  One acquire (L1 invalidate) per workgroup.
// Layout of `bar` (uint32, zeroed by mg_reset_kernel before the launch): [0] top counter, [1] abort flag (sticky across
// launches: never zeroed), [32 (1 + x)] counter of group x, [32 (9 + x)] generation of group x.
#define MG_BAR_WORDS (32 * 17)
struct MgBar { unsigned gen; int leader; };
__device__ __forceinline__ void mg_arrive(unsigned* bar, MgBar& st, int nwg, int dbg = 0) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (dbg & 1) return;
  st.gen += 1;
  st.leader = 0;
  if (threadIdx.x == 0) {
    const int x = blockIdx.x & 7;
    const unsigned members = (unsigned)((nwg - x + 7) / 8);             // workgroups with blockIdx & 7 == x
    const unsigned old = __hip_atomic_fetch_add(bar + 32 * (1 + x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == st.gen * members) {                                  // last of the group
      st.leader = 1;
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
__device__ __forceinline__ void mg_wait(unsigned* bar, const MgBar& st, int nwg, int dbg = 0) {
  if (threadIdx.x == 0 && !(dbg & 1)) {
    const int x = blockIdx.x & 7;
    const unsigned groups = nwg < 8 ? (unsigned)nwg : 8u;
    unsigned spins = 0;
    if (st.leader) {
      while ((int)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - st.gen * groups) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 23)) { __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        if ((spins & 1023u) == 0 && __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
      __hip_atomic_store(bar + 32 * (9 + x), st.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while ((int)(__hip_atomic_load(bar + 32 * (9 + x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - st.gen) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 23)) { __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        if ((spins & 1023u) == 0 && __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}
__device__ __forceinline__ void mg_store_f32(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mg_store_u16(bf16_t* p, bf16_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mg_store_u32(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- GEMV over the packed copy: NWP = waves per 16-column block in the layout (4: two blocks per workgroup pass, 8: one) ----
template <int NWP>
__device__ __forceinline__ const bf16_t* mg_wave_ptr(const bf16_t* P, int K, long unit, int wave, int lane, int* block, int* nb_per) {
  const int nsteps = K / 64, per = (nsteps + NWP - 1) / NWP;
  const int sw = wave % NWP;
  *block = (int)(unit * (MG_NW / NWP) + wave / NWP);
  *nb_per = per;
  return P + ((size_t)(*block) * NWP + sw) * per * 1024 + lane * 8;
}

template <int NWP>
__device__ __forceinline__ void mg_prefetch(const bf16_t* P, int N, int K, long unit, MgBatch& b) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int block, per;
  const bf16_t* wp = mg_wave_ptr<NWP>(P, K, unit, wave, lane, &block, &per);
  const int nsteps = K / 64, sw = wave % NWP;
  const int s0 = sw * per, s_end = (s0 + per) < nsteps ? (s0 + per) : nsteps;
  if (block * 16 < N && s0 + MG_U <= s_end) {
#pragma unroll
    for (int u = 0; u < MG_U; ++u) {
      b.w0[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + (size_t)u * 1024));
      b.w1[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + (size_t)u * 1024 + 512));
    }
  }
}

// C[M, N] = xs[M, K] . W^T (+ res); `b` holds the first batch of this workgroup's first unit (mg_prefetch); on return it
// holds nothing.  red: [MG_NW][256] floats of LDS.
template <int NWP>
__device__ __forceinline__ void mg_gemv(const bf16_t* P, int N, int K, const bf16_t* xs, int M, float* red, void* Cv, int ldc,
                                        int out_f32, const float* res, int ldr, int g, int G, MgBatch& b) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int nb = (N + 15) / 16, nunits = (nb + (MG_NW / NWP) - 1) / (MG_NW / NWP);
  const int nsteps = K / 64;
  const int mrow = lr < M ? lr : M - 1;
  const bf16_t* xp = xs + (size_t)mrow * K + lg * 16;
  for (long unit = g; unit < nunits; unit += G) {
    int block, per;
    const bf16_t* wp = mg_wave_ptr<NWP>(P, K, unit, wave, lane, &block, &per);
    const int sw = wave % NWP;
    int s = sw * per;
    const int s0 = s;
    const int s_end = (s + per) < nsteps ? (s + per) : nsteps;
    const bool live = block < nb;
    float4_t acc = (float4_t){0.f, 0.f, 0.f, 0.f};
    if (live) {
      bool have = s + MG_U <= s_end;                          // the batch in `b` was requested by mg_prefetch under the same test
      while (have) {
#pragma unroll
        for (int u = 0; u < MG_U; ++u) {
          const int k = (s + u) * 64;
          const short8_t x0 = *reinterpret_cast<const short8_t*>(xp + k), x1 = *reinterpret_cast<const short8_t*>(xp + k + 8);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0, b.w0[u], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, b.w1[u], acc, 0, 0, 0);
          if (u & 1) __builtin_amdgcn_sched_barrier(0);       // keep the LDS operand reads next to their MFMAs: 16 registers, not 64
        }
        s += MG_U;
        have = s + MG_U <= s_end;
        if (have) {
#pragma unroll
          for (int u = 0; u < MG_U; ++u) {
            b.w0[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + (size_t)(s - s0 + u) * 1024));
            b.w1[u] = __builtin_nontemporal_load(reinterpret_cast<const short8_t*>(wp + (size_t)(s - s0 + u) * 1024 + 512));
          }
        }
      }
      for (; s < s_end; ++s) {
        const int k = s * 64;
        const short8_t w0 = *reinterpret_cast<const short8_t*>(wp + (size_t)(s - s0) * 1024);
        const short8_t w1 = *reinterpret_cast<const short8_t*>(wp + (size_t)(s - s0) * 1024 + 512);
        const short8_t x0 = *reinterpret_cast<const short8_t*>(xp + k), x1 = *reinterpret_cast<const short8_t*>(xp + k + 8);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0, w0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, w1, acc, 0, 0, 0);
      }
    }
    if (unit + G < nunits) mg_prefetch<NWP>(P, N, K, unit + G, b);   // the next unit's first batch flies through the reduction
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + (4 * lg + r) * 16 + lr] = acc[r];
    __syncthreads();
    if ((wave % NWP) == 0 && live) {
      const int n0 = block * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * lg + r, n = n0 + lr;
        if (m < M && n < N) {
          float v = 0.f;
#pragma unroll
          for (int w = 0; w < NWP; ++w) v += red[(wave + w) * 256 + m * 16 + lr];
          v *= 1.0f;
          if (res) v += res[(size_t)m * ldr + n];
          if (out_f32) mg_store_f32(reinterpret_cast<float*>(Cv) + (size_t)m * ldc + n, v);
          else mg_store_u16(reinterpret_cast<bf16_t*>(Cv) + (size_t)m * ldc + n, f2bf(v));
        }
      }
    }
    __syncthreads();
  }
}

// operand rows: bf16(w * (h * rsqrt(mean(h^2) + eps))) -- rmsnorm_fwd_kernel's summation order and expression (K <= 4096)
__device__ __forceinline__ void mg_rows_rmsnorm(const float* h, const float* w, int M, int K, float eps, bf16_t* xs, float* bred) {
  const int tid = threadIdx.x;
  for (int m = 0; m < M; ++m) {
    const float* xr = h + (size_t)m * K;
    float4_t hv[4];
    float ss = 0.f;
    int c = 0;
    if (tid < 256)
      for (int i = tid * 4; i < K; i += 1024, ++c) {
        hv[c] = *reinterpret_cast<const float4_t*>(xr + i);
        ss += hv[c][0] * hv[c][0] + hv[c][1] * hv[c][1] + hv[c][2] * hv[c][2] + hv[c][3] * hv[c][3];
      }
    ss = block_sum<MG_NW>(ss, bred);
    const float r = rsqrtf(ss / K + eps);
    c = 0;
    if (tid < 256)
      for (int i = tid * 4; i < K; i += 1024, ++c) {
        const float4_t g = *reinterpret_cast<const float4_t*>(w + i);
        uint2 pk;
        pk.x = pack_bf2(g[0] * (hv[c][0] * r), g[1] * (hv[c][1] * r));
        pk.y = pack_bf2(g[2] * (hv[c][2] * r), g[3] * (hv[c][3] * r));
        *reinterpret_cast<uint2*>(xs + (size_t)m * K + i) = pk;
      }
  }
  __syncthreads();
}

__device__ __forceinline__ void mg_rows_silu(const bf16_t* gu, int M, int I, bf16_t* xs) {
  const int per_row = I >> 3;
  for (int it = threadIdx.x; it < M * per_row; it += MG_NT) {
    const int m = it / per_row, c = (it - m * per_row) * 8;
    const long gc = (long)(c >> 7) * 256 + (c & 127);
    const short8_t g = *reinterpret_cast<const short8_t*>(gu + (size_t)m * 2 * I + gc);
    const short8_t u = *reinterpret_cast<const short8_t*>(gu + (size_t)m * 2 * I + gc + 128);
    short8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gv = bf2f((bf16_t)g[e]), uv = bf2f((bf16_t)u[e]);
      o[e] = (short)f2bf(gv / (1.f + __expf(-gv)) * uv);
    }
    *reinterpret_cast<short8_t*>(xs + (size_t)m * I + c) = o;
  }
  __syncthreads();
}

__device__ __forceinline__ void mg_rows_copy(const bf16_t* x, int M, int K, bf16_t* xs) {
  for (int it = threadIdx.x; it < M * (K >> 3); it += MG_NT)
    *reinterpret_cast<short8_t*>(xs + (size_t)it * 8) = *reinterpret_cast<const short8_t*>(x + (size_t)it * 8);
  __syncthreads();
}

// attn_decode_kernel for one (b, head) with its rotary + KV append prologue, 16 waves emulated by 8 (vw = wave, wave + 8)
__device__ __forceinline__ void mg_attention(const MegaParams& p, const MegaLayer& L, int unit, float* sc, float* part) {
  __shared__ __attribute__((aligned(16))) bf16_t qs[128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = p.hd, H = p.H, W = p.H * p.hd;
  const int b = unit / H, h = unit % H;
  int len = p.kvlen[b];
  len = len < p.T_cap ? len : p.T_cap;
  bf16_t* src = p.qkv + (size_t)b * 3 * W;
  bf16_t* cache_b = L.cache + (size_t)b * p.cache_bs;
  {
    const int half = D >> 1, items = half >> 2;
    bf16_t* crow = cache_b + (size_t)p.pos_dev[0] * p.ld_cache;
    if (tid < 2 * items) {
      const int which = tid / items, i = (tid % items) * 4;
      bf16_t* e = src + which * W + h * D + i;
      const int ps = p.pos[b];
      const float4_t c4 = *reinterpret_cast<const float4_t*>(p.cs + (size_t)ps * half + i);
      const float4_t s4 = *reinterpret_cast<const float4_t*>(p.sn + (size_t)ps * half + i);
      const short4_t a = *reinterpret_cast<const short4_t*>(e);
      const short4_t bb = *reinterpret_cast<const short4_t*>(e + half);
      short4_t oa, ob;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float x1 = bf2f((bf16_t)a[t]), x2 = bf2f((bf16_t)bb[t]);
        oa[t] = (short)f2bf(x1 * c4[t] - x2 * s4[t]);
        ob[t] = (short)f2bf(x2 * c4[t] + x1 * s4[t]);
      }
      // the rotated q stays in LDS.  (In place in `qkv`, as the per-op kernel does, a plain store would leave a dirty line in
      // this XCD's L2 that the NEXT layer's write-through store of the same element -- from a workgroup on another XCD -- does
      // not see: the stale line then wins or loses by eviction order.  Found as a 1-in-600-tokens mismatch.)
      bf16_t* dst = which == 0 ? qs + i : crow + h * D + i;
      *reinterpret_cast<short4_t*>(dst) = oa;
      *reinterpret_cast<short4_t*>(dst + half) = ob;
    } else if (tid < 2 * items + (D >> 3)) {
      const int c = (tid - 2 * items) * 8;
      *reinterpret_cast<short8_t*>(crow + W + h * D + c) = *reinterpret_cast<const short8_t*>(src + 2 * W + h * D + c);
    }
    __syncthreads();
  }
  const bf16_t* qp = qs;
  const bf16_t* kp = cache_b + h * D;
  const bf16_t* vp = cache_b + W + h * D;
  const int ldk = p.ld_cache;
  const int sub = lane & 15, kq = lane >> 4;
  float qf[8];
  const bool dim_ok = sub * 8 < D;
  {
    short8_t qv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (dim_ok) qv = *reinterpret_cast<const short8_t*>(qp + sub * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[e] = bf2f((bf16_t)qv[e]) * p.scale;
  }
  for (int vw = wave; vw < 16; vw += MG_NW) {
    for (int j0 = 0; j0 < len; j0 += 2 * 16 * 4) {
      short8_t kv[2];
      int jj[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        jj[u] = j0 + u * 16 * 4 + vw * 4 + kq;
        kv[u] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
        if (jj[u] < len && dim_ok) kv[u] = *reinterpret_cast<const short8_t*>(kp + (size_t)jj[u] * ldk + sub * 8);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += qf[e] * bf2f((bf16_t)kv[u][e]);
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 8, 64);
        if (sub == 0 && jj[u] < len) sc[jj[u]] = s;
      }
    }
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int j = lane; j < len; j += 64) mx = fmaxf(mx, sc[j]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < len; j += 64) sum += __expf(sc[j] - mx);
  sum = wave_sum(sum);
  const bool own = 2 * lane < D;
  for (int vw = wave; vw < 16; vw += MG_NW) {
    float o0 = 0.f, o1 = 0.f;
    int j = vw;
    for (; j + 7 * 16 < len; j += 8 * 16) {
      unsigned vv[8];
      float pj[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        vv[u] = own ? *reinterpret_cast<const unsigned*>(vp + (size_t)(j + 16 * u) * ldk + 2 * lane) : 0u;
        pj[u] = __expf(sc[j + 16 * u] - mx);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        o0 += pj[u] * bf2f((bf16_t)(vv[u] & 0xffffu));
        o1 += pj[u] * bf2f((bf16_t)(vv[u] >> 16));
      }
    }
    {
      unsigned vv[7];
      float pj[7];
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        const int jr = j + 16 * u;
        const bool ok = jr < len;
        vv[u] = (own && ok) ? *reinterpret_cast<const unsigned*>(vp + (size_t)jr * ldk + 2 * lane) : 0u;
        pj[u] = ok ? __expf(sc[jr] - mx) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        o0 += pj[u] * bf2f((bf16_t)(vv[u] & 0xffffu));
        o1 += pj[u] * bf2f((bf16_t)(vv[u] >> 16));
      }
    }
    part[vw * 128 + 2 * lane] = o0;
    part[vw * 128 + 2 * lane + 1] = o1;
  }
  __syncthreads();
  if (wave == 0 && own) {
    const float inv = len > 0 ? 1.f / sum : 0.f;
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      r0 += part[w * 128 + 2 * lane];
      r1 += part[w * 128 + 2 * lane + 1];
    }
    mg_store_u32(reinterpret_cast<unsigned*>(p.o + (size_t)b * W + h * D + 2 * lane), pack_bf2(r0 * inv, r1 * inv));
  }
  __syncthreads();
}

__global__ __launch_bounds__(MG_NT, 2) void decode_mega_kernel(MegaParams p) {
  extern __shared__ __attribute__((aligned(16))) char msm[];
  // LDS: operand rows [M][max(D, I)] bf16 | reduction scratch [8][256] f32 | attention scores [T_cap] + partials [16][128] f32
  const int KX = p.I > p.D ? p.I : p.D;
  bf16_t* xs = reinterpret_cast<bf16_t*>(msm);
  float* red = reinterpret_cast<float*>(msm + (((size_t)p.M * KX * 2 + 255) & ~(size_t)255));
  float* sc = red + MG_NW * 256;
  float* part = sc + ((p.T_cap + 63) & ~63);
  __shared__ float bred[MG_NW];
  const int g = blockIdx.x, G = gridDim.x;
  const int M = p.M, D = p.D, I = p.I;
  MgBar target = {0u, 0};                                  // mg_reset_kernel zeroed the counters in front of this launch
  MgBatch b;
  float* h = p.h;
  float* h2 = p.h2;
  mg_prefetch<4>(p.layers[0].wqkv, 3 * D, D, g, b);
  for (int li = 0; li < p.n_layers; ++li) {
    const MegaLayer L = p.layers[li];
    // P1: q | k | v
    MG_STAMP(0)
    mg_rows_rmsnorm(h, L.ln1, M, D, p.eps, xs, bred);
    MG_STAMP(1)
    mg_gemv<4>(L.wqkv, 3 * D, D, xs, M, red, p.qkv, 3 * D, 0, nullptr, 0, g, G, b);
    MG_STAMP(2)
    mg_arrive(p.bar, target, G, p.dbg);
    mg_prefetch<8>(L.wo, D, D, g, b);                      // all of o_proj: 16 KiB per wave
    mg_wait(p.bar, target, G, p.dbg);
    MG_STAMP(3)
    // P2: rotary + append + attention, one (row, head) per workgroup
    if (!(p.dbg & 2))
      for (int unit = g; unit < M * p.H; unit += G) mg_attention(p, L, unit, sc, part);
    MG_STAMP(4)
    mg_arrive(p.bar, target, G, p.dbg);
    mg_wait(p.bar, target, G, p.dbg);
    MG_STAMP(5)
    // P3: o_proj + residual
    mg_rows_copy(p.o, M, D, xs);
    mg_gemv<8>(L.wo, D, D, xs, M, red, h2, D, 1, h, D, g, G, b);
    MG_STAMP(6)
    mg_arrive(p.bar, target, G, p.dbg);
    mg_prefetch<4>(L.wgu, 2 * I, D, g, b);
    mg_wait(p.bar, target, G, p.dbg);
    MG_STAMP(7)
    // P4: gate | up
    mg_rows_rmsnorm(h2, L.ln2, M, D, p.eps, xs, bred);
    MG_STAMP(8)
    mg_gemv<4>(L.wgu, 2 * I, D, xs, M, red, p.gu, 2 * I, 0, nullptr, 0, g, G, b);
    MG_STAMP(9)
    mg_arrive(p.bar, target, G, p.dbg);
    mg_prefetch<8>(L.wd, D, I, g, b);
    mg_wait(p.bar, target, G, p.dbg);
    MG_STAMP(10)
    // P5: SiLU gate + down projection + residual
    mg_rows_silu(p.gu, M, I, xs);
    MG_STAMP(11)
    mg_gemv<8>(L.wd, D, I, xs, M, red, h, D, 1, h2, D, g, G, b);
    MG_STAMP(12)
    mg_arrive(p.bar, target, G, p.dbg);
    if (li + 1 < p.n_layers) mg_prefetch<4>(p.layers[li + 1].wqkv, 3 * D, D, g, b);
    else mg_prefetch<4>(p.lm_head, p.V, D, g, b);
    mg_wait(p.bar, target, G, p.dbg);
    MG_STAMP(13)
  }
  mg_rows_rmsnorm(h, p.norm, M, D, p.eps, xs, bred);
  mg_gemv<4>(p.lm_head, p.V, D, xs, M, red, p.logits, p.V, 1, nullptr, 0, g, G, b);
}

__global__ void mg_reset_kernel(unsigned* bar) {
  for (int i = threadIdx.x; i < MG_BAR_WORDS; i += blockDim.x)
    if (i != 1) bar[i] = 0u;
}

static long long* g_mega_trace = nullptr;
extern "C" void mhdbg_set_mega_trace(void* ptr) { g_mega_trace = (long long*)ptr; }   // debug hook, not part of the ABI

extern "C" long mh_decode_mega_lds_bytes(int M, int D, int I, int T_cap) {
  const int KX = I > D ? I : D;
  return (long)((((size_t)M * KX * 2 + 255) & ~(size_t)255) + MG_NW * 256 * 4 + (((size_t)T_cap + 63) & ~(size_t)63) * 4 + 16 * 128 * 4);
}

// layers: device array of n_layers MegaLayer records (7 pointers each: packed wqkv [3D, D], wo [D, D], wgu [2I, D] 128-blocked
// gate|up, wd [D, I]; ln1, ln2 f32 [D]; cache [B][T_cap][2D] bf16).  bar: 1024 zero-initialised unsigneds owned by the caller (MG_BAR_WORDS used):
// [0] the barrier's arrival counter (a one-thread launch in front of the kernel zeroes it), [1] a sticky abort flag a
// workgroup raises if a barrier ever times out (the step's results are then garbage: check it).  n_wg: workgroups = CUs that
// are certainly free (all must be resident at once).  Returns MH_ERR_UNSUPPORTED for shapes the kernel does not cover (M > 2,
// head_dim != 128, D > 4096 or not a multiple of 1024, I not a multiple of 128, matrices whose packed layout is not the
// 4-wave form for q|k|v, gate|up, lm_head and the 8-wave form for o_proj, down).
extern "C" int mh_decode_mega(const void* layers, int n_layers, int M, int D, int H, int hd, int I, int V, int T_cap, float eps,
                              float scale, float* h, float* h2, void* qkv, void* o, void* gu, const float* norm,
                              const void* lm_head, float* logits, const int* pos, const int* pos_dev, const int* kvlen,
                              const float* cos_tab, const float* sin_tab, long cache_bstride, long ld_cache, void* bar, int n_wg,
                              hipStream_t stream) {
  if (n_layers <= 0 || M <= 0) return MH_OK;
  if (M > MG_MAXM || hd != 128 || H * hd != D || D > 4096 || (D % 1024) || (I % 128) || (V % 4) || T_cap <= 0 || T_cap > 8192 ||
      n_wg < 1 || n_wg > 1024)
    return MH_ERR_UNSUPPORTED;
  if ((3 * D + 15) / 16 < 512 || (D + 15) / 16 >= 512 || (2 * I + 15) / 16 < 512 || (V + 15) / 16 < 512) return MH_ERR_UNSUPPORTED;
  if (!layers || !h || !h2 || !qkv || !o || !gu || !norm || !lm_head || !logits || !pos || !pos_dev || !kvlen || !cos_tab || !sin_tab || !bar)
    return MH_ERR_ARG;
  MegaParams p;
  p.layers = (const MegaLayer*)layers; p.n_layers = n_layers; p.M = M; p.D = D; p.H = H; p.hd = hd; p.I = I; p.V = V; p.T_cap = T_cap;
  p.eps = eps; p.scale = scale; p.h = h; p.h2 = h2; p.qkv = (bf16_t*)qkv; p.o = (bf16_t*)o; p.gu = (bf16_t*)gu; p.norm = norm;
  p.lm_head = (const bf16_t*)lm_head; p.logits = logits; p.pos = pos; p.pos_dev = pos_dev; p.kvlen = kvlen; p.cs = cos_tab; p.sn = sin_tab;
  p.cache_bs = cache_bstride; p.ld_cache = (int)ld_cache; p.bar = (unsigned*)bar; p.trace = g_mega_trace;
  static const int dbg = getenv("MYRIAD_MEGA_DBG") ? atoi(getenv("MYRIAD_MEGA_DBG")) : 0;
  p.dbg = dbg;
  const size_t sh = (size_t)mh_decode_mega_lds_bytes(M, D, I, T_cap);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)decode_mega_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attr = true; }
  if (sh > 96 * 1024) return MH_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(mg_reset_kernel, dim3(1), dim3(256), 0, stream, (unsigned*)bar);
  hipLaunchKernelGGL(decode_mega_kernel, dim3(n_wg), dim3(MG_NT), sh, stream, p);
  MH_CHECK_LAUNCH();
  return MH_OK;
}

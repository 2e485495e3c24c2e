// The opaque context of SURVEY 8(b): the library state a caller owns explicitly -- the split-K scratch record and the
// gradient-exchange resources (an RCCL communicator + a side HIP stream + two events) behind mh_allreduce_start / _wait
// (reference minigpt4/runners/runner_base.py:94-98: DistributedDataParallel's gradient all-reduce; here ONE all-reduce of the
// flat gradient buffer per step, issued on the context's side stream so that it overlaps whatever the producer stream does
// next, e.g. the next step's frozen ViT forward).
//
// RCCL is loaded lazily (dlopen) at the first mh_ctx_comm_id / mh_ctx_comm_init: the library has no link-time dependency on
// it, a process that never exchanges gradients (one GPU, tests) never loads it, and a host framework that ships its own RCCL
// is not disturbed.  The handle is process-wide, opened ONCE and never closed (RTLD_NODELETE): ncclGetUniqueId starts RCCL's
// bootstrap listener thread inside that library image, so unloading it (round 3 dlclose'd after fetching the id) could unmap
// code a live thread runs.
//
// Round 4: the three verbs data parallelism needs are behind the ABI -- all-reduce, reduce-scatter and all-gather, fp32 or
// bf16 on the wire -- so runner.DataParallel's 'rs_ag' mode and its bf16 wire run through the context as well; several verbs
// may be started back to back (they queue on the context's side stream), one mh_allreduce_wait covers them all.
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>

typedef struct { char internal[128]; } mh_rccl_id;                 // == ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*fn_get_id)(mh_rccl_id*);
typedef int (*fn_init_rank)(void**, int, mh_rccl_id, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_reduce_scatter)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);

struct mh_ctx {
  MhScratch scratch;
  hipStream_t side;
  hipEvent_t ev_in, ev_out;
  void* comm;                  // ncclComm_t
  int rank, world;
  fn_all_reduce all_reduce;
  fn_reduce_scatter reduce_scatter;
  fn_all_gather all_gather;
  fn_destroy destroy;
  bool pending;
};

// process-wide, opened once, never closed
static void* open_rccl() {
  static void* handle = nullptr;
  if (handle) return handle;
  // MYRIAD_RCCL_LIB=<path>: that image and no other (a site-specific RCCL build; the two-ranks-on-one-GPU test's stand-in,
  // tests/fake_rccl) -- no fallback to the search path when it cannot be loaded: the caller then sees MH_ERR_UNSUPPORTED
  if (const char* forced = getenv("MYRIAD_RCCL_LIB")) {
    if (forced[0]) {
      handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL | RTLD_NODELETE);
      return handle;
    }
  }
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
  // a host framework that already brought an RCCL into the process (PyTorch-ROCm ships its own): use THAT image -- one
  // bootstrap, one set of IPC handles per process -- before loading another copy from the search path
  for (const char* n : names) {
    void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD | RTLD_NODELETE);
    if (h) { handle = h; return h; }
  }
  for (const char* n : names) {
    void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NODELETE);
    if (h) { handle = h; return h; }
  }
  return nullptr;
}

// wire element types of the exchange verbs: MH_DT_F32 / MH_DT_BF16 (include/myriad_hip.h) -> ncclFloat32 (7) / ncclBfloat16 (9)
static inline int nccl_dtype(int dt) { return dt == 1 ? 9 : 7; }
static inline size_t dt_bytes(int dt) { return dt == 1 ? 2 : 4; }

extern "C" int mh_ctx_create(mh_ctx** out) {
  if (!out) return MH_ERR_ARG;
  mh_ctx* c = (mh_ctx*)calloc(1, sizeof(mh_ctx));
  if (!c) return MH_ERR_ARG;
  c->world = 1;
  if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming) != hipSuccess) {
    free(c);
    return MH_ERR_LAUNCH;
  }
  *out = c;
  return MH_OK;
}

extern "C" int mh_ctx_destroy(mh_ctx* c) {
  if (!c) return MH_OK;
  if (g_scratch == &c->scratch) g_scratch = &g_default_scratch;
  if (c->comm && c->destroy) (void)c->destroy(c->comm);
  (void)hipEventDestroy(c->ev_in);
  (void)hipEventDestroy(c->ev_out);
  (void)hipStreamDestroy(c->side);
  free(c);
  return MH_OK;
}

// The split-K scratch of THIS context: stream == NULL registers the main scratch, otherwise the scratch of one more stream
// that may split K concurrently (up to 4); ptr == NULL unregisters.  Takes effect for the calls made while the context is
// current (mh_ctx_make_current).
extern "C" int mh_ctx_set_workspace(mh_ctx* c, hipStream_t stream, void* ptr, long bytes) {
  if (!c) return MH_ERR_ARG;
  MhScratch& s = c->scratch;
  if (!stream) {
    s.ws = (float*)ptr;
    s.bytes = ptr ? (size_t)bytes : 0;
    if (!ptr)
      for (int i = 0; i < MH_MAX_ALT_WS; ++i) { s.alt[i] = nullptr; s.alt_stream[i] = nullptr; }
    return MH_OK;
  }
  if (ptr && (!s.ws || (size_t)bytes < s.bytes)) return MH_ERR_ARG;
  int slot = -1;
  for (int i = 0; i < MH_MAX_ALT_WS; ++i)
    if (s.alt_stream[i] == stream) slot = i;
  if (slot < 0)
    for (int i = MH_MAX_ALT_WS - 1; i >= 0; --i)
      if (!s.alt_stream[i]) slot = i;
  if (slot < 0) return ptr ? MH_ERR_UNSUPPORTED : MH_OK;
  s.alt[slot] = (float*)ptr;
  s.alt_stream[slot] = ptr ? stream : nullptr;
  return MH_OK;
}

// Library calls use the scratch of the current context; NULL selects the process default (mh_set_workspace).
extern "C" int mh_ctx_make_current(mh_ctx* c) {
  g_scratch = c ? &c->scratch : &g_default_scratch;
  return MH_OK;
}

// 128 bytes identifying a new communicator: rank 0 generates it, the caller hands it to every rank (any channel), every rank
// then calls mh_ctx_comm_init with it.
extern "C" int mh_ctx_comm_id(void* id128) {
  if (!id128) return MH_ERR_ARG;
  void* h = open_rccl();
  if (!h) return MH_ERR_UNSUPPORTED;
  fn_get_id get = (fn_get_id)dlsym(h, "ncclGetUniqueId");
  mh_rccl_id id;
  const int rc = get ? get(&id) : 1;
  if (rc == 0) memcpy(id128, &id, sizeof(id));
  return rc == 0 ? MH_OK : MH_ERR_LAUNCH;
}

extern "C" int mh_ctx_comm_init(mh_ctx* c, const void* id128, int rank, int world) {
  if (!c || !id128 || world < 1 || rank < 0 || rank >= world || c->comm) return MH_ERR_ARG;
  void* h = open_rccl();
  if (!h) return MH_ERR_UNSUPPORTED;
  fn_init_rank init = (fn_init_rank)dlsym(h, "ncclCommInitRank");
  c->all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
  c->reduce_scatter = (fn_reduce_scatter)dlsym(h, "ncclReduceScatter");
  c->all_gather = (fn_all_gather)dlsym(h, "ncclAllGather");
  c->destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
  if (!init || !c->all_reduce || !c->reduce_scatter || !c->all_gather || !c->destroy) return MH_ERR_UNSUPPORTED;
  mh_rccl_id id;
  memcpy(&id, id128, sizeof(id));
  if (init(&c->comm, world, id, rank) != 0) { c->comm = nullptr; return MH_ERR_LAUNCH; }
  c->rank = rank;
  c->world = world;
  return MH_OK;
}

extern "C" int mh_ctx_world(const mh_ctx* c) { return c ? c->world : 0; }

// Every verb: ordered after everything `producer` has queued so far, runs on the context's side stream, returns at once;
// mh_allreduce_wait makes a stream depend on everything started so far.  With a communicator the verb is RCCL's, also at one
// rank (the GPU tests run exactly that: the dlsym'd entry points with their real argument lists); a context without a
// communicator is a one-rank world by definition -- the all-reduce is the identity, reduce-scatter / all-gather copy the one
// shard (device to device) when the buffers differ.
static int exchange_begin(mh_ctx* c, hipStream_t producer) {
  if (hipEventRecord(c->ev_in, producer) != hipSuccess) return MH_ERR_LAUNCH;
  if (hipStreamWaitEvent(c->side, c->ev_in, 0) != hipSuccess) return MH_ERR_LAUNCH;
  return MH_OK;
}
static int exchange_end(mh_ctx* c) {
  if (hipEventRecord(c->ev_out, c->side) != hipSuccess) return MH_ERR_LAUNCH;
  c->pending = true;
  return MH_OK;
}

// In-place sum over the ranks of buf[0 .. n), elements of wire type dt (MH_DT_F32 / MH_DT_BF16).
extern "C" int mh_allreduce_start_dt(mh_ctx* c, void* buf, long n, int dt, hipStream_t producer) {
  if (!c || (!buf && n > 0) || n < 0 || (dt != 0 && dt != 1)) return MH_ERR_ARG;
  if (c->world > 1 && n > 0 && !c->comm) return MH_ERR_ARG;
  int rc = exchange_begin(c, producer);
  if (rc) return rc;
  if (c->comm && n > 0 && c->all_reduce(buf, buf, (size_t)n, nccl_dtype(dt), /*ncclSum*/ 0, c->comm, c->side) != 0) return MH_ERR_LAUNCH;
  return exchange_end(c);
}
extern "C" int mh_allreduce_start(mh_ctx* c, float* buf, long n, hipStream_t producer) {
  return mh_allreduce_start_dt(c, buf, n, 0, producer);
}

// recv[0 .. n_per_rank) = sum over ranks of their send[rank * n_per_rank .. + n_per_rank)  (send holds world * n_per_rank
// elements; recv may point into send at this rank's own piece: RCCL reduces in place then).
extern "C" int mh_reduce_scatter_start(mh_ctx* c, const void* send, void* recv, long n_per_rank, int dt, hipStream_t producer) {
  if (!c || n_per_rank < 0 || (n_per_rank > 0 && (!send || !recv)) || (dt != 0 && dt != 1)) return MH_ERR_ARG;
  if (c->world > 1 && n_per_rank > 0 && !c->comm) return MH_ERR_ARG;
  int rc = exchange_begin(c, producer);
  if (rc) return rc;
  if (n_per_rank > 0) {
    if (c->comm) {
      if (c->reduce_scatter(send, recv, (size_t)n_per_rank, nccl_dtype(dt), /*ncclSum*/ 0, c->comm, c->side) != 0) return MH_ERR_LAUNCH;
    } else if (send != recv) {
      if (hipMemcpyAsync(recv, send, (size_t)n_per_rank * dt_bytes(dt), hipMemcpyDeviceToDevice, c->side) != hipSuccess) return MH_ERR_LAUNCH;
    }
  }
  return exchange_end(c);
}

// recv[r * n_per_rank .. + n_per_rank) = rank r's send[0 .. n_per_rank) for every r  (recv holds world * n_per_rank elements;
// send may be this rank's own piece of recv).
extern "C" int mh_allgather_start(mh_ctx* c, const void* send, void* recv, long n_per_rank, int dt, hipStream_t producer) {
  if (!c || n_per_rank < 0 || (n_per_rank > 0 && (!send || !recv)) || (dt != 0 && dt != 1)) return MH_ERR_ARG;
  if (c->world > 1 && n_per_rank > 0 && !c->comm) return MH_ERR_ARG;
  int rc = exchange_begin(c, producer);
  if (rc) return rc;
  if (n_per_rank > 0) {
    if (c->comm) {
      if (c->all_gather(send, recv, (size_t)n_per_rank, nccl_dtype(dt), c->comm, c->side) != 0) return MH_ERR_LAUNCH;
    } else if (send != recv) {
      if (hipMemcpyAsync(recv, send, (size_t)n_per_rank * dt_bytes(dt), hipMemcpyDeviceToDevice, c->side) != hipSuccess) return MH_ERR_LAUNCH;
    }
  }
  return exchange_end(c);
}

// `consumer` waits (on the device) for every exchange started so far; no host synchronisation.
extern "C" int mh_allreduce_wait(mh_ctx* c, hipStream_t consumer) {
  if (!c) return MH_ERR_ARG;
  if (!c->pending) return MH_OK;
  if (hipStreamWaitEvent(consumer, c->ev_out, 0) != hipSuccess) return MH_ERR_LAUNCH;
  c->pending = false;
  return MH_OK;
}

// The opaque context of SURVEY 8(b): the library state a caller owns explicitly -- the split-K scratch record and the
// gradient-exchange resources (an RCCL communicator + a side HIP stream + two events) behind mh_allreduce_start / _wait
// (reference minigpt4/runners/runner_base.py:94-98: DistributedDataParallel's gradient all-reduce; here ONE all-reduce of the
// flat gradient buffer per step, issued on the context's side stream so that it overlaps whatever the producer stream does
// next, e.g. the next step's frozen ViT forward).
//
// RCCL is loaded lazily (dlopen) by mh_ctx_comm_init only: the library has no link-time dependency on it, a process that
// never exchanges gradients (one GPU, tests) never loads it, and a host framework that ships its own RCCL is not disturbed.
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>

typedef struct { char internal[128]; } mh_rccl_id;                 // == ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*fn_get_id)(mh_rccl_id*);
typedef int (*fn_init_rank)(void**, int, mh_rccl_id, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);

struct mh_ctx {
  MhScratch scratch;
  hipStream_t side;
  hipEvent_t ev_in, ev_out;
  void* rccl;                  // dlopen handle
  void* comm;                  // ncclComm_t
  int rank, world;
  fn_all_reduce all_reduce;
  fn_destroy destroy;
  bool pending;
};

static void* open_rccl() {
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) return h;
  }
  return nullptr;
}

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
  if (c->rccl) dlclose(c->rccl);
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
  dlclose(h);
  return rc == 0 ? MH_OK : MH_ERR_LAUNCH;
}

extern "C" int mh_ctx_comm_init(mh_ctx* c, const void* id128, int rank, int world) {
  if (!c || !id128 || world < 1 || rank < 0 || rank >= world || c->comm) return MH_ERR_ARG;
  c->rccl = open_rccl();
  if (!c->rccl) return MH_ERR_UNSUPPORTED;
  fn_init_rank init = (fn_init_rank)dlsym(c->rccl, "ncclCommInitRank");
  c->all_reduce = (fn_all_reduce)dlsym(c->rccl, "ncclAllReduce");
  c->destroy = (fn_destroy)dlsym(c->rccl, "ncclCommDestroy");
  if (!init || !c->all_reduce || !c->destroy) return MH_ERR_UNSUPPORTED;
  mh_rccl_id id;
  memcpy(&id, id128, sizeof(id));
  if (init(&c->comm, world, id, rank) != 0) { c->comm = nullptr; return MH_ERR_LAUNCH; }
  c->rank = rank;
  c->world = world;
  return MH_OK;
}

extern "C" int mh_ctx_world(const mh_ctx* c) { return c ? c->world : 0; }

// In-place sum over the ranks of buf[0 .. n) (f32) on the context's side stream, ordered after everything `producer` has
// queued so far.  Returns at once; mh_allreduce_wait makes a stream depend on the result.  world == 1: no communicator is
// needed, the call only records the dependency.
extern "C" int mh_allreduce_start(mh_ctx* c, float* buf, long n, hipStream_t producer) {
  if (!c || (!buf && n > 0) || n < 0 || c->pending) return MH_ERR_ARG;
  if (hipEventRecord(c->ev_in, producer) != hipSuccess) return MH_ERR_LAUNCH;
  if (hipStreamWaitEvent(c->side, c->ev_in, 0) != hipSuccess) return MH_ERR_LAUNCH;
  if (c->world > 1 && n > 0) {
    if (!c->comm) return MH_ERR_ARG;
    if (c->all_reduce(buf, buf, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm, c->side) != 0) return MH_ERR_LAUNCH;
  }
  if (hipEventRecord(c->ev_out, c->side) != hipSuccess) return MH_ERR_LAUNCH;
  c->pending = true;
  return MH_OK;
}

// `consumer` waits (on the device) for the exchange started last; no host synchronisation.
extern "C" int mh_allreduce_wait(mh_ctx* c, hipStream_t consumer) {
  if (!c) return MH_ERR_ARG;
  if (!c->pending) return MH_OK;
  if (hipStreamWaitEvent(consumer, c->ev_out, 0) != hipSuccess) return MH_ERR_LAUNCH;
  c->pending = false;
  return MH_OK;
}

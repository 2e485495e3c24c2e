// Test stand-in for librccl: the six entry points csrc/ctx.hip binds (ncclGetUniqueId / CommInitRank / AllReduce / ReduceScatter /
// AllGather / CommDestroy) over a file-backed shared mapping, so that the mh_ctx gradient exchange can run with TWO ranks on ONE
// GPU (RCCL itself refuses two ranks per device).  Loaded through MYRIAD_RCCL_LIB by tests/dp_worker.py only; never by the
// product.  Semantics of the real verbs -- sum over ranks in rank order, fp32 or bf16 elements (bf16: fp32 accumulate, one
// round-to-nearest-even), stream-ordered after the work already queued on `stream` -- implemented synchronously: wait for the
// stream, stage chunks through the mapping with a process-shared counter barrier, copy the result back.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
static const size_t CHUNK = 16u << 20;           // bytes staged per rank per round
static const int MAXR = 8;

struct Header { std::atomic<uint64_t> arrived; };
struct Comm {
  int rank, world;
  char path[160];
  char* base;          // Header | slot[0] | slot[1] | ...
  size_t map_bytes;
  uint64_t barriers;   // barriers this rank has passed
  char* tmp;           // private result chunk
};
static char* slot(Comm* c, int r) { return c->base + 4096 + (size_t)r * CHUNK; }

static void barrier(Comm* c) {
  Header* h = (Header*)c->base;
  h->arrived.fetch_add(1, std::memory_order_acq_rel);
  c->barriers++;
  const uint64_t want = c->barriers * (uint64_t)c->world;
  struct timespec ts = {0, 20000};
  while (h->arrived.load(std::memory_order_acquire) < want) nanosleep(&ts, nullptr);
}

static inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline uint16_t f2bf(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static size_t esize(int dt) { return dt == 9 ? 2 : 4; }     // ncclBfloat16 = 9, ncclFloat32 = 7

// out[0 .. n) = sum over ranks (rank order) of their slot contents
static void reduce_slots(Comm* c, void* out, size_t n, int dt) {
  if (dt == 9) {
    uint16_t* o = (uint16_t*)out;
    for (size_t i = 0; i < n; ++i) {
      float s = bf2f(((uint16_t*)slot(c, 0))[i]);
      for (int r = 1; r < c->world; ++r) s += bf2f(((uint16_t*)slot(c, r))[i]);
      o[i] = f2bf(s);
    }
  } else {
    float* o = (float*)out;
    for (size_t i = 0; i < n; ++i) {
      float s = ((float*)slot(c, 0))[i];
      for (int r = 1; r < c->world; ++r) s += ((float*)slot(c, r))[i];
      o[i] = s;
    }
  }
}

#define HIPOK(x) do { if ((x) != hipSuccess) return 1; } while (0)

extern "C" int ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
  snprintf(id->internal, sizeof(id->internal), "/tmp/fake_rccl_%d_%ld_%ld", (int)getpid(), (long)ts.tv_sec, (long)ts.tv_nsec);
  return 0;
}

extern "C" int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return 1;
  Comm* c = (Comm*)calloc(1, sizeof(Comm));
  c->rank = rank; c->world = nranks;
  id.internal[127] = 0;
  snprintf(c->path, sizeof(c->path), "%s", id.internal);
  c->map_bytes = 4096 + (size_t)nranks * CHUNK;
  const int fd = open(c->path, O_RDWR | O_CREAT, 0600);
  if (fd < 0) { free(c); return 1; }
  if (ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); free(c); return 1; }     // a fresh file reads as zeros: arrived = 0
  c->base = (char*)mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->base == MAP_FAILED) { free(c); return 1; }
  c->tmp = (char*)malloc(CHUNK);
  barrier(c);                                     // every rank has mapped the file (the real call is a rendezvous too)
  *comm = c;
  return 0;
}

extern "C" int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return 0;
  munmap(c->base, c->map_bytes);
  if (c->rank == 0) unlink(c->path);
  free(c->tmp);
  free(c);
  return 0;
}

extern "C" int ncclAllReduce(const void* send, void* recv, size_t count, int dt, int op, void* comm, hipStream_t stream) {
  Comm* c = (Comm*)comm;
  if (!c || op != 0 || (dt != 7 && dt != 9)) return 1;
  HIPOK(hipStreamSynchronize(stream));
  const size_t es = esize(dt), per = CHUNK / es;
  for (size_t off = 0; off < count; off += per) {
    const size_t n = count - off < per ? count - off : per;
    HIPOK(hipMemcpy(slot(c, c->rank), (const char*)send + off * es, n * es, hipMemcpyDeviceToHost));
    barrier(c);
    reduce_slots(c, c->tmp, n, dt);
    barrier(c);
    HIPOK(hipMemcpy((char*)recv + off * es, c->tmp, n * es, hipMemcpyHostToDevice));
  }
  return 0;
}

extern "C" int ncclReduceScatter(const void* send, void* recv, size_t recvcount, int dt, int op, void* comm, hipStream_t stream) {
  Comm* c = (Comm*)comm;
  if (!c || op != 0 || (dt != 7 && dt != 9)) return 1;
  HIPOK(hipStreamSynchronize(stream));
  const size_t es = esize(dt), per = CHUNK / es;
  for (int d = 0; d < c->world; ++d)
    for (size_t off = 0; off < recvcount; off += per) {
      const size_t n = recvcount - off < per ? recvcount - off : per;
      HIPOK(hipMemcpy(slot(c, c->rank), (const char*)send + ((size_t)d * recvcount + off) * es, n * es, hipMemcpyDeviceToHost));
      barrier(c);
      if (d == c->rank) reduce_slots(c, c->tmp, n, dt);
      barrier(c);
      if (d == c->rank) HIPOK(hipMemcpy((char*)recv + off * es, c->tmp, n * es, hipMemcpyHostToDevice));
    }
  return 0;
}

extern "C" int ncclAllGather(const void* send, void* recv, size_t sendcount, int dt, void* comm, hipStream_t stream) {
  Comm* c = (Comm*)comm;
  if (!c || (dt != 7 && dt != 9)) return 1;
  HIPOK(hipStreamSynchronize(stream));
  const size_t es = esize(dt), per = CHUNK / es;
  for (int src = 0; src < c->world; ++src)
    for (size_t off = 0; off < sendcount; off += per) {
      const size_t n = sendcount - off < per ? sendcount - off : per;
      if (src == c->rank) HIPOK(hipMemcpy(slot(c, 0), (const char*)send + off * es, n * es, hipMemcpyDeviceToHost));
      barrier(c);
      HIPOK(hipMemcpy((char*)recv + ((size_t)src * sendcount + off) * es, slot(c, 0), n * es, hipMemcpyHostToDevice));
      barrier(c);
    }
  return 0;
}

// svo_comm.hip -- the exchange steps of the multi-GPU path behind the C-ABI: a communicator object the
// host C++ hands to svo_hip_sia_run_sharded / svo_hip_seed_gather_converged_dev, so that the drop-in can shard without
// Python or torch in the process.
//
// Two transports behind one interface (sum-all-reduce of doubles in place, all-gather of equal blocks in place):
//   * RCCL over xGMI (the product path): ncclAllReduce / ncclAllGather enqueued on the context's stream.  librccl is
//     resolved at run time (dlopen of the copy the process already holds -- e.g. the one torch loaded -- else
//     librccl.so.1 from the loader path), so libsvo_hip.so has no link-time dependency on it and a single-GPU
//     deployment needs no RCCL at all.  The application either lets the library create the communicator from a
//     ncclUniqueId it distributes itself (svo_hip_comm_unique_id / svo_hip_comm_create_rccl) or hands over an existing
//     ncclComm_t (svo_hip_comm_from_nccl).
//   * host-staged exchange through a POSIX shared-memory segment (svo_hip_comm_create_shm): for ranks that share one
//     device or have no peer path -- bring-up and the world-size-2 tests on the one-GPU box.  Device -> host copy, a
//     barrier in the segment, every rank sums the slots in rank order (bitwise identical result on every rank, as a
//     ring all-reduce gives), host -> device copy.  It blocks the calling thread; it is not a performance path.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>

#include "svo_internal.h"

namespace {

// ---- RCCL, resolved at run time (prototypes as in rccl/rccl.h of ROCm 7.2) ----------------------------------
typedef void* NcclComm;
struct NcclUniqueId { char internal[128]; };
typedef int (*fn_GetUniqueId)(NcclUniqueId*);
typedef int (*fn_CommInitRank)(NcclComm*, int, NcclUniqueId, int);
typedef int (*fn_CommDestroy)(NcclComm);
typedef int (*fn_AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*fn_AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t);
typedef const char* (*fn_GetErrorString)(int);
typedef int (*fn_CommCount)(NcclComm, int*);
constexpr int kNcclSum = 0, kNcclInt8 = 0, kNcclFloat64 = 8;      // ncclRedOp_t / ncclDataType_t values

struct Rccl {
  void* lib = nullptr;
  fn_GetUniqueId GetUniqueId = nullptr;
  fn_CommInitRank CommInitRank = nullptr;
  fn_CommDestroy CommDestroy = nullptr;
  fn_AllReduce AllReduce = nullptr;
  fn_AllGather AllGather = nullptr;
  fn_GetErrorString GetErrorString = nullptr;
  fn_CommCount CommCount = nullptr;
  std::string err;
};

Rccl& rccl() {
  static Rccl r;
  if (r.lib || !r.err.empty()) return r;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (r.lib) break; }       // a copy already in the process
  if (!r.lib) for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.lib) break; }
  if (!r.lib) {
    const char* e = dlerror();                                 // (one call: dlerror() clears the message it returns)
    r.err = std::string("librccl not found: ") + (e ? e : "");
    return r;
  }
  r.GetUniqueId = (fn_GetUniqueId)dlsym(r.lib, "ncclGetUniqueId");
  r.CommInitRank = (fn_CommInitRank)dlsym(r.lib, "ncclCommInitRank");
  r.CommDestroy = (fn_CommDestroy)dlsym(r.lib, "ncclCommDestroy");
  r.AllReduce = (fn_AllReduce)dlsym(r.lib, "ncclAllReduce");
  r.AllGather = (fn_AllGather)dlsym(r.lib, "ncclAllGather");
  r.GetErrorString = (fn_GetErrorString)dlsym(r.lib, "ncclGetErrorString");
  r.CommCount = (fn_CommCount)dlsym(r.lib, "ncclCommCount");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.AllGather) {
    r.err = "librccl lacks an expected symbol";
    r.lib = nullptr;
  }
  return r;
}

int nccl_fail(svo_hip_ctx* ctx, const char* what, int rc) {
  const Rccl& r = rccl();
  return svo_fail(ctx, SVO_HIP_ERR_DEVICE, what, r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
}

// ---- shared-memory segment of the host-staged transport -----------------------------------------------------
struct ShmHeader {
  unsigned magic;              // set by rank 0 once the segment is sized and zeroed
  unsigned world;
  unsigned long long slot_bytes;
  unsigned arrived;            // barrier: ranks that reached the current generation
  unsigned generation;
  unsigned go;                 // set by rank 0 once IT has passed the first barrier: the segment belongs to a live group
};
constexpr unsigned kShmMagic = 0x53564f43u;      // "SVOC"
constexpr size_t kShmHeaderBytes = 256;

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

}  // namespace

struct svo_hip_comm {
  svo_hip_ctx* ctx = nullptr;
  int rank = 0, world = 1;
  int kind = 0;                // 0 = RCCL, 1 = shared memory
  NcclComm nccl = nullptr;
  bool own_nccl = false;
  // shared memory
  std::string shm_name;
  int shm_fd = -1;
  unsigned char* shm = nullptr;
  size_t shm_bytes = 0, slot_bytes = 0;
  unsigned char* host = nullptr;            // page-locked staging of one slot
  double timeout_s = 60.0;
  bool failed = false;                      // a barrier timed out: the segment's counters are no longer consistent
  unsigned long long id = 0;                // unique per communicator object of this process (graph-replay cache key)
};

namespace {

ShmHeader* hdr(svo_hip_comm* c) { return reinterpret_cast<ShmHeader*>(c->shm); }
unsigned char* slot(svo_hip_comm* c, int r) { return c->shm + kShmHeaderBytes + (size_t)r * c->slot_bytes; }

// sense-reversing barrier over the segment; returns false on timeout (a dead peer must not hang the caller for ever)
bool shm_barrier(svo_hip_comm* c) {
  ShmHeader* h = hdr(c);
  const unsigned gen = __atomic_load_n(&h->generation, __ATOMIC_ACQUIRE);
  if (__atomic_add_fetch(&h->arrived, 1u, __ATOMIC_ACQ_REL) == (unsigned)c->world) {
    __atomic_store_n(&h->arrived, 0u, __ATOMIC_RELAXED);
    __atomic_add_fetch(&h->generation, 1u, __ATOMIC_RELEASE);
    return true;
  }
  const double t0 = now_s();
  while (__atomic_load_n(&h->generation, __ATOMIC_ACQUIRE) == gen) {
    if (now_s() - t0 > c->timeout_s) return false;
    usleep(20);
  }
  return true;
}

// a timed-out barrier leaves `arrived` incremented: the communicator is unusable from then on and says so
int shm_timeout(svo_hip_comm* c) {
  c->failed = true;
  return svo_fail(c->ctx, SVO_HIP_ERR_DEVICE, "svo_hip_comm (shm)", "timed out waiting for the other ranks");
}

int shm_exchange(svo_hip_comm* c, void* dev, size_t bytes, bool reduce_f64, void* gather_dst_dev) {
  svo_hip_ctx* ctx = c->ctx;
  if (c->failed) return svo_fail(ctx, SVO_HIP_ERR_STATE, "svo_hip_comm (shm)", "an earlier exchange timed out: destroy the communicator");
  if (bytes > c->slot_bytes) return svo_fail(ctx, SVO_HIP_ERR_INVALID, "svo_hip_comm (shm)", "message larger than the segment's slot");
  SVO_CHECK_HIP(ctx, hipMemcpyAsync(c->host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(slot(c, c->rank), c->host, bytes);
  if (!shm_barrier(c)) return shm_timeout(c);
  if (reduce_f64) {
    const size_t n = bytes / sizeof(double);
    double* acc = reinterpret_cast<double*>(c->host);
    for (size_t i = 0; i < n; ++i) acc[i] = 0.0;
    for (int r = 0; r < c->world; ++r) {                       // fixed rank order: the same bits on every rank
      const double* s = reinterpret_cast<const double*>(slot(c, r));
      for (size_t i = 0; i < n; ++i) acc[i] += s[i];
    }
    SVO_CHECK_HIP(ctx, hipMemcpyAsync(dev, c->host, bytes, hipMemcpyHostToDevice, ctx->stream));
    SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  } else {
    for (int r = 0; r < c->world; ++r) {
      SVO_CHECK_HIP(ctx, hipMemcpyAsync(static_cast<unsigned char*>(gather_dst_dev) + (size_t)r * bytes, slot(c, r), bytes,
                                        hipMemcpyHostToDevice, ctx->stream));
      SVO_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the slot is pageable shared memory: finish before it is reused
    }
  }
  if (!shm_barrier(c)) return shm_timeout(c);
  return SVO_HIP_OK;
}

unsigned long long next_comm_id() {
  static unsigned long long n = 0;
  return __atomic_add_fetch(&n, 1ull, __ATOMIC_RELAXED);
}

}  // namespace

unsigned long long svo_comm_id(const svo_hip_comm* c) { return c ? c->id : 0ull; }
svo_hip_ctx* svo_comm_ctx(const svo_hip_comm* c) { return c ? c->ctx : nullptr; }

// used by svo_sia.hip / svo_depth.hip
int svo_comm_all_reduce_sum_f64(svo_hip_comm* c, double* dev, size_t count) {
  if (!c) return SVO_HIP_ERR_INVALID;
  if (c->kind == 0) {                                          // also with one rank: the call sequence is the product's
    const int rc = rccl().AllReduce(dev, dev, count, kNcclFloat64, kNcclSum, c->nccl, c->ctx->stream);
    return rc == 0 ? SVO_HIP_OK : nccl_fail(c->ctx, "ncclAllReduce", rc);
  }
  if (c->world == 1) return SVO_HIP_OK;
  return shm_exchange(c, dev, count * sizeof(double), true, nullptr);
}

// every rank contributes `bytes` at recv_dev + rank * bytes (in place) and receives all blocks
int svo_comm_all_gather(svo_hip_comm* c, void* recv_dev, size_t bytes) {
  if (!c) return SVO_HIP_ERR_INVALID;
  unsigned char* mine = static_cast<unsigned char*>(recv_dev) + (size_t)c->rank * bytes;
  if (c->kind == 0) {
    const int rc = rccl().AllGather(mine, recv_dev, bytes, kNcclInt8, c->nccl, c->ctx->stream);
    return rc == 0 ? SVO_HIP_OK : nccl_fail(c->ctx, "ncclAllGather", rc);
  }
  if (c->world == 1) return SVO_HIP_OK;
  return shm_exchange(c, mine, bytes, false, recv_dev);
}

extern "C" {

int svo_hip_comm_unique_id(void* id128) {
  if (!id128) return SVO_HIP_ERR_INVALID;
  Rccl& r = rccl();
  if (!r.lib) return SVO_HIP_ERR_DEVICE;
  NcclUniqueId id;
  if (r.GetUniqueId(&id) != 0) return SVO_HIP_ERR_DEVICE;
  memcpy(id128, id.internal, 128);
  return SVO_HIP_OK;
}

int svo_hip_comm_create_rccl(svo_hip_ctx* ctx, const void* id128, int rank, int world, svo_hip_comm** out) {
  if (!ctx || !out) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, id128 && world >= 1 && rank >= 0 && rank < world);
  Rccl& r = rccl();
  if (!r.lib) return svo_fail(ctx, SVO_HIP_ERR_DEVICE, "svo_hip_comm_create_rccl", r.err.c_str());
  SVO_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  NcclUniqueId id;
  memcpy(id.internal, id128, 128);
  NcclComm comm = nullptr;
  const int rc = r.CommInitRank(&comm, world, id, rank);
  if (rc != 0) return nccl_fail(ctx, "ncclCommInitRank", rc);
  svo_hip_comm* c = new svo_hip_comm;
  c->ctx = ctx; c->rank = rank; c->world = world; c->kind = 0; c->nccl = comm; c->own_nccl = true;
  c->id = next_comm_id();
  *out = c;
  return SVO_HIP_OK;
}

int svo_hip_comm_from_nccl(svo_hip_ctx* ctx, void* nccl_comm, int rank, int world, svo_hip_comm** out) {
  if (!ctx || !out) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, nccl_comm && world >= 1 && rank >= 0 && rank < world);
  Rccl& r = rccl();
  if (!r.lib) return svo_fail(ctx, SVO_HIP_ERR_DEVICE, "svo_hip_comm_from_nccl", r.err.c_str());
  svo_hip_comm* c = new svo_hip_comm;
  c->ctx = ctx; c->rank = rank; c->world = world; c->kind = 0; c->nccl = nccl_comm; c->own_nccl = false;
  c->id = next_comm_id();
  *out = c;
  return SVO_HIP_OK;
}

int svo_hip_comm_create_shm(svo_hip_ctx* ctx, const char* name, int rank, int world, size_t slot_bytes, svo_hip_comm** out) {
  if (!ctx || !out) return SVO_HIP_ERR_INVALID;
  SVO_REQUIRE(ctx, name && name[0] == '/' && world >= 1 && world <= 64 && rank >= 0 && rank < world && slot_bytes >= 8);
  slot_bytes = (slot_bytes + 63) & ~(size_t)63;
  const size_t total = kShmHeaderBytes + (size_t)world * slot_bytes;
  svo_hip_comm* c = new svo_hip_comm;
  c->ctx = ctx; c->rank = rank; c->world = world; c->kind = 1; c->shm_name = name; c->slot_bytes = slot_bytes; c->shm_bytes = total;
  c->id = next_comm_id();
  bool created = false;                                        // rank 0: the name exists and is ours to remove
  auto unmap = [&]() {
    if (c->shm) { munmap(c->shm, total); c->shm = nullptr; }
    if (c->shm_fd >= 0) { close(c->shm_fd); c->shm_fd = -1; }
  };
  auto fail = [&](const char* what) {
    const int rc = svo_fail(ctx, SVO_HIP_ERR_DEVICE, "svo_hip_comm_create_shm", what);
    unmap();
    if (created) shm_unlink(name);                             // never leave a segment behind that a later run could attach to
    if (c->host) (void)hipHostFree(c->host);
    delete c;
    return rc;
  };
  const double t0 = now_s();
  void* hp = nullptr;
  if (hipHostMalloc(&hp, slot_bytes, hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc failed");
  c->host = static_cast<unsigned char*>(hp);
  if (rank == 0) {
    shm_unlink(name);                                          // a stale segment of an earlier run
    c->shm_fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (c->shm_fd < 0) return fail("shm_open (create) failed");
    created = true;
    if (ftruncate(c->shm_fd, (off_t)total) != 0) return fail("ftruncate failed");
    void* p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, c->shm_fd, 0);
    if (p == MAP_FAILED) return fail("mmap failed");
    c->shm = static_cast<unsigned char*>(p);
    ShmHeader* h = hdr(c);
    h->world = (unsigned)world; h->slot_bytes = slot_bytes; h->arrived = 0; h->generation = 0; h->go = 0;
    __atomic_store_n(&h->magic, kShmMagic, __ATOMIC_RELEASE);
    if (!shm_barrier(c)) return fail("timed out at the first barrier");
    __atomic_store_n(&h->go, 1u, __ATOMIC_RELEASE);            // the attaching ranks wait for this before they trust the segment
    shm_unlink(name);                                          // everybody has it mapped: the name can go
    *out = c;
    return SVO_HIP_OK;
  }
  // rank > 0: attach to the segment the name refers to.  The name may still refer to the segment of a crashed earlier
  // run when this rank gets here first (rank 0 unlinks and recreates it): while waiting, the name is looked up again, and
  // when it has moved to another inode this rank lets go of the dead one and attaches to the new one.
  for (;;) {
    if (now_s() - t0 > c->timeout_s) return fail("timed out waiting for rank 0's segment");
    c->shm_fd = shm_open(name, O_RDWR, 0600);
    struct stat st;
    if (c->shm_fd < 0 || fstat(c->shm_fd, &st) != 0 || (size_t)st.st_size < total) {
      if (c->shm_fd >= 0) { close(c->shm_fd); c->shm_fd = -1; }
      usleep(1000);
      continue;
    }
    void* p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, c->shm_fd, 0);
    if (p == MAP_FAILED) return fail("mmap failed");
    c->shm = static_cast<unsigned char*>(p);
    ShmHeader* h = hdr(c);
    auto name_moved = [&]() -> bool {                          // does the name point at another inode by now?
      const int fd2 = shm_open(name, O_RDWR, 0600);
      if (fd2 < 0) return false;                               // (unlinked: either rank 0 is between unlink and create, or all is well)
      struct stat s2;
      const bool moved = fstat(fd2, &s2) == 0 && (s2.st_ino != st.st_ino || s2.st_dev != st.st_dev);
      close(fd2);
      return moved;
    };
    bool restart = false;
    while (__atomic_load_n(&h->magic, __ATOMIC_ACQUIRE) != kShmMagic) {
      if (now_s() - t0 > c->timeout_s) return fail("timed out waiting for rank 0 to initialise the segment");
      if (name_moved()) { restart = true; break; }
      usleep(1000);
    }
    if (!restart) {
      if (h->world != (unsigned)world || h->slot_bytes != slot_bytes) {
        if (!name_moved()) return fail("segment geometry differs from this rank's arguments");
        restart = true;
      }
    }
    if (!restart) {
      // first barrier, with the same look-out (a stale segment has its magic set: this is where a rank would sit)
      const unsigned gen = __atomic_load_n(&h->generation, __ATOMIC_ACQUIRE);
      if (__atomic_add_fetch(&h->arrived, 1u, __ATOMIC_ACQ_REL) == (unsigned)world) {
        __atomic_store_n(&h->arrived, 0u, __ATOMIC_RELAXED);
        __atomic_add_fetch(&h->generation, 1u, __ATOMIC_RELEASE);
      } else {
        double t_check = now_s();
        while (__atomic_load_n(&h->generation, __ATOMIC_ACQUIRE) == gen) {
          const double t = now_s();
          if (t - t0 > c->timeout_s) return fail("timed out at the first barrier");
          if (t - t_check > 0.01) { t_check = t; if (name_moved()) { restart = true; break; } }
          usleep(20);
        }
      }
    }
    if (!restart) {
      // The barrier of a DEAD segment can open too: a rank 0 that was killed while it sat in its first barrier leaves
      // magic set and arrived == world - 1, and this rank's own increment completes the count.  Only a live rank 0 sets
      // `go` (after it has passed the barrier itself); until then the name is watched as above.
      double t_check = now_s();
      while (__atomic_load_n(&h->go, __ATOMIC_ACQUIRE) != 1u) {
        const double t = now_s();
        if (t - t0 > c->timeout_s) return fail("timed out waiting for rank 0 behind the first barrier");
        if (t - t_check > 0.01) { t_check = t; if (name_moved()) { restart = true; break; } }
        usleep(20);
      }
    }
    if (!restart) break;
    unmap();
  }
  *out = c;
  return SVO_HIP_OK;
}

int svo_hip_comm_destroy(svo_hip_comm* c) {
  if (!c) return SVO_HIP_ERR_INVALID;
  if (c->kind == 0 && c->own_nccl && c->nccl) (void)rccl().CommDestroy(c->nccl);
  if (c->host) (void)hipHostFree(c->host);
  if (c->shm) munmap(c->shm, c->shm_bytes);
  if (c->shm_fd >= 0) close(c->shm_fd);
  delete c;
  return SVO_HIP_OK;
}

int svo_hip_comm_info(const svo_hip_comm* c, int* rank, int* world, int* kind) {
  if (!c) return SVO_HIP_ERR_INVALID;
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (kind) *kind = c->kind;
  return SVO_HIP_OK;
}

// The number of ranks the TRANSPORT itself reports: ncclCommCount of the RCCL communicator, the world size stored in the
// shared-memory segment's header.  (svo_hip_comm_info returns what the caller passed at creation.)
int svo_hip_comm_count(const svo_hip_comm* c, int* count) {
  if (!c || !count) return SVO_HIP_ERR_INVALID;
  if (c->kind == 0) {
    const Rccl& r = rccl();
    if (!r.CommCount || !c->nccl) return SVO_HIP_ERR_STATE;
    int n = 0;
    const int rc = r.CommCount(c->nccl, &n);
    if (rc != 0) return nccl_fail(c->ctx, "ncclCommCount", rc);
    *count = n;
  } else {
    if (!c->shm) return SVO_HIP_ERR_STATE;
    *count = (int)reinterpret_cast<const ShmHeader*>(c->shm)->world;
  }
  return SVO_HIP_OK;
}

}  // extern "C"

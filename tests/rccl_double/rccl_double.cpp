// TEST DOUBLE of librccl for ranks that SHARE one GPU (RCCL itself refuses two ranks on one device, and the boxes
// the tests run on have one).  It exports the seven symbols csrc/comm.cpp resolves with dlsym (ncclGetUniqueId,
// ncclCommInitRank, ncclCommDestroy, ncclAllReduce, ncclAllGather, ncclBroadcast, ncclGetErrorString) with RCCL's
// signatures and stream semantics as far as a caller can observe them: a collective is ordered after the work already
// enqueued on `stream` and before the work enqueued after it; every rank receives the sum in RANK ORDER (so the ranks
// hold bit-identical results, as RCCL's fixed reduction trees give them).  Transport: a file in /tmp mapped by every
// rank (the "unique id" is its name), chunks staged device -> host -> file -> host -> device, two process barriers per
// chunk.  It blocks the calling host thread (RCCL does not); a barrier gives up after SLV_DBL_TIMEOUT_S seconds
// (default 120) with ncclSystemError instead of hanging the box.
//
// Test infrastructure only: loaded through slv_comm_load(path) by tests/test_native_comm_gpu.py; never shipped, never
// the thing measured.  Build: hipcc -shared -fPIC tests/rccl_double/rccl_double.cpp -o tests/rccl_double/librccl_double.so
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace {

constexpr size_t CHUNK = 8u << 20;      // bytes per rank and round trip
constexpr int MAX_RANKS = 8;

struct Header {
  std::atomic<int> arrived;             // barrier: arrivals of the current generation
  std::atomic<int> generation;
  std::atomic<int> attached;            // ranks that mapped the file (the last to leave unlinks it)
  std::atomic<int> error;               // a rank gave up: everybody gives up
  char pad[4096 - 4 * sizeof(std::atomic<int>)];
};

struct Comm {
  int rank, world, fd;
  char name[128];
  Header* hdr;
  unsigned char* slots;                 // [world][CHUNK]
  std::vector<unsigned char> tmp;
  size_t map_bytes;
};

std::mutex g_mutex;
double timeout_s() {
  const char* e = getenv("SLV_DBL_TIMEOUT_S");
  return e ? atof(e) : 120.0;
}

bool barrier(Comm* c) {
  Header* h = c->hdr;
  const int gen = h->generation.load(std::memory_order_acquire);
  if (h->arrived.fetch_add(1, std::memory_order_acq_rel) == c->world - 1) {
    h->arrived.store(0, std::memory_order_relaxed);
    h->generation.store(gen + 1, std::memory_order_release);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (h->generation.load(std::memory_order_acquire) == gen) {
    if (h->error.load(std::memory_order_relaxed)) return false;
    if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if ((spins & 1023) == 0 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) {
      h->error.store(1, std::memory_order_relaxed);
      fprintf(stderr, "rccl_double: rank %d gave up at a barrier (peer missing or collectives out of order)\n", c->rank);
      return false;
    }
  }
  return true;
}

size_t dtype_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

template <typename T>
void reduce_ranks(const Comm* c, T* out, size_t n, ncclRedOp_t op) {
  for (size_t i = 0; i < n; ++i) {
    T acc = ((const T*)(c->slots))[i];
    for (int r = 1; r < c->world; ++r) acc = acc + ((const T*)(c->slots + (size_t)r * CHUNK))[i];      // rank order
    out[i] = acc;
  }
  if (op == ncclAvg) {
    const T inv = (T)1 / (T)c->world;                       // RCCL's ncclAvg: sum, then one multiply by 1 / nranks
    for (size_t i = 0; i < n; ++i) out[i] = out[i] * inv;
  }
}

#define DBL_HIP(call)                                                                          \
  do {                                                                                         \
    hipError_t e__ = (call);                                                                   \
    if (e__ != hipSuccess) {                                                                   \
      fprintf(stderr, "rccl_double: %s -> %s\n", #call, hipGetErrorString(e__));               \
      return ncclUnhandledCudaError;                                                           \
    }                                                                                          \
  } while (0)

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
  const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
  snprintf(id->internal, NCCL_UNIQUE_ID_BYTES, "/tmp/slv_rccl_double_%d_%llx", (int)getpid(), (unsigned long long)now);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  std::lock_guard<std::mutex> lk(g_mutex);
  Comm* c = new Comm();
  c->rank = rank;
  c->world = nranks;
  snprintf(c->name, sizeof(c->name), "%s", id.internal);
  c->map_bytes = sizeof(Header) + (size_t)nranks * CHUNK;
  c->fd = open(c->name, O_RDWR | O_CREAT, 0600);
  if (c->fd < 0) {
    delete c;
    return ncclSystemError;
  }
  if (ftruncate(c->fd, (off_t)c->map_bytes) != 0) {        // (every rank sets the same size; new pages read as zero)
    close(c->fd);
    delete c;
    return ncclSystemError;
  }
  void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
  if (p == MAP_FAILED) {
    close(c->fd);
    delete c;
    return ncclSystemError;
  }
  c->hdr = (Header*)p;
  c->slots = (unsigned char*)p + sizeof(Header);
  c->tmp.resize(CHUNK);
  c->hdr->attached.fetch_add(1);
  if (!barrier(c)) return ncclSystemError;                  // like ncclCommInitRank: returns once every rank has joined
  *comm = (ncclComm_t)c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = (Comm*)comm;
  if (!c) return ncclSuccess;
  std::lock_guard<std::mutex> lk(g_mutex);
  const bool last = c->hdr->attached.fetch_sub(1) == 1;
  munmap((void*)c->hdr, c->map_bytes);
  close(c->fd);
  if (last) unlink(c->name);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {                    // (optional symbol: slv_comm_count's diagnostics)
  if (!comm || !count) return ncclInvalidArgument;
  *count = ((const Comm*)comm)->world;
  return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) { return ncclCommDestroy(comm); }     // (optional symbol of the product's watchdog path)

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
  Comm* c = (Comm*)comm;
  const size_t es = dtype_size(dt);
  if (!c || !es || (op != ncclSum && op != ncclAvg)) return ncclInvalidArgument;
  if (dt != ncclFloat32 && dt != ncclFloat64 && dt != ncclInt64 && dt != ncclInt32) return ncclInvalidArgument;
  std::lock_guard<std::mutex> lk(g_mutex);
  DBL_HIP(hipStreamSynchronize(stream));
  const size_t per = CHUNK / es;
  for (size_t off = 0; off < count; off += per) {
    const size_t n = count - off < per ? count - off : per;
    DBL_HIP(hipMemcpy(c->slots + (size_t)c->rank * CHUNK, (const char*)send + off * es, n * es, hipMemcpyDeviceToHost));
    if (!barrier(c)) return ncclSystemError;
    if (dt == ncclFloat32) reduce_ranks(c, (float*)c->tmp.data(), n, op);
    else if (dt == ncclFloat64) reduce_ranks(c, (double*)c->tmp.data(), n, op);
    else if (dt == ncclInt64) reduce_ranks(c, (long long*)c->tmp.data(), n, ncclSum);
    else reduce_ranks(c, (int*)c->tmp.data(), n, ncclSum);
    if (!barrier(c)) return ncclSystemError;                // every rank has read the slots: they may be overwritten
    DBL_HIP(hipMemcpy((char*)recv + off * es, c->tmp.data(), n * es, hipMemcpyHostToDevice));
  }
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t comm,
                           hipStream_t stream) {
  Comm* c = (Comm*)comm;
  const size_t es = dtype_size(dt);
  if (!c || !es) return ncclInvalidArgument;
  std::lock_guard<std::mutex> lk(g_mutex);
  DBL_HIP(hipStreamSynchronize(stream));
  const size_t bytes = count * es;
  for (size_t off = 0; off < bytes; off += CHUNK) {
    const size_t n = bytes - off < CHUNK ? bytes - off : CHUNK;
    DBL_HIP(hipMemcpy(c->slots + (size_t)c->rank * CHUNK, (const char*)send + off, n, hipMemcpyDeviceToHost));
    if (!barrier(c)) return ncclSystemError;
    for (int r = 0; r < c->world; ++r)
      DBL_HIP(hipMemcpy((char*)recv + (size_t)r * bytes + off, c->slots + (size_t)r * CHUNK, n, hipMemcpyHostToDevice));
    if (!barrier(c)) return ncclSystemError;
  }
  return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t dt, int root, ncclComm_t comm,
                           hipStream_t stream) {
  Comm* c = (Comm*)comm;
  const size_t es = dtype_size(dt);
  if (!c || !es || root < 0 || root >= c->world) return ncclInvalidArgument;
  std::lock_guard<std::mutex> lk(g_mutex);
  DBL_HIP(hipStreamSynchronize(stream));
  const size_t bytes = count * es;
  for (size_t off = 0; off < bytes; off += CHUNK) {
    const size_t n = bytes - off < CHUNK ? bytes - off : CHUNK;
    if (c->rank == root) DBL_HIP(hipMemcpy(c->slots, (const char*)send + off, n, hipMemcpyDeviceToHost));
    if (!barrier(c)) return ncclSystemError;
    if (c->rank != root || recv != send) DBL_HIP(hipMemcpy((char*)recv + off, c->slots, n, hipMemcpyHostToDevice));
    if (!barrier(c)) return ncclSystemError;
  }
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "success (rccl_double)";
    case ncclUnhandledCudaError: return "HIP error (rccl_double)";
    case ncclSystemError: return "system error / barrier timeout (rccl_double)";
    case ncclInvalidArgument: return "invalid argument (rccl_double)";
    default: return "error (rccl_double)";
  }
}

}  // extern "C"

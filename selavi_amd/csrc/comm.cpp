// RCCL behind the C ABI (comm {init from unique id, allreduce_f32/f64, allgather, destroy}).
// Replaces the NCCL process group of /root/reference/utils.py:133-146 for the hot path's exchanges: the SyncBN sums
// (main.py:117-118), the Sinkhorn-Knopp column sums (sk_utils.py:401 re-designed as a K+1 fp64 all-reduce) and, if the
// host side wants it, the gradient buckets (main.py:156-160).  One communicator per process (= per GPU); every
// collective is enqueued on the CALLER's stream, in order with the kernels around it -- no process-group stream, no
// event hops, no host round trip between "partials -> sums", the all-reduce and "sums -> coefficients".
//
// librccl is resolved at run time (dlopen): the copy already loaded into the process (PyTorch-ROCm brings its own) is
// preferred so that one RCCL runtime serves both; nothing links against it at build time.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.hpp"
#include "../../include/selavi_hip.h"

namespace slv {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                       // optional (watchdog path)
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr; // optional
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;            // optional (diagnostics)
  char path[512] = {0};
};
static RcclApi g_api;
static std::mutex g_api_mutex;

static int load_api(const char* path_hint) {
  std::lock_guard<std::mutex> lk(g_api_mutex);
  if (g_api.handle) return 0;
  void* h = nullptr;
  if (path_hint && path_hint[0]) {
    // an explicit path is loaded as named (never replaced by a librccl that happens to be in the process already:
    // PyTorch-ROCm brings one) and must load
    h = dlopen(path_hint, RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(-5, "slv_comm: cannot load the named librccl: %s", dlerror());
    snprintf(g_api.path, sizeof(g_api.path), "%s", path_hint);
  }
  const char* tried[3] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  // 1. a copy that is already in the process (RTLD_NOLOAD), 2. load by name / path
  for (int pass = 0; pass < 2 && !h; ++pass)
    for (int i = 0; i < 3 && !h; ++i) {
      h = dlopen(tried[i], RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (h) snprintf(g_api.path, sizeof(g_api.path), "%s", tried[i]);
    }
  if (!h) return fail(-5, "slv_comm: librccl.so not found (dlopen): %s", dlerror());
#define SLV_SYM(field, name)                                                   \
  *(void**)(&g_api.field) = dlsym(h, name);                                    \
  if (!g_api.field) {                                                          \
    dlclose(h);                                                                \
    return fail(-5, "slv_comm: symbol %s missing in librccl", name);           \
  }
  SLV_SYM(GetUniqueId, "ncclGetUniqueId");
  SLV_SYM(CommInitRank, "ncclCommInitRank");
  SLV_SYM(CommDestroy, "ncclCommDestroy");
  SLV_SYM(AllReduce, "ncclAllReduce");
  SLV_SYM(AllGather, "ncclAllGather");
  SLV_SYM(Broadcast, "ncclBroadcast");
  SLV_SYM(GetErrorString, "ncclGetErrorString");
#undef SLV_SYM
  *(void**)(&g_api.CommAbort) = dlsym(h, "ncclCommAbort");
  *(void**)(&g_api.CommGetAsyncError) = dlsym(h, "ncclCommGetAsyncError");
  *(void**)(&g_api.CommCount) = dlsym(h, "ncclCommCount");
  g_api.handle = h;
  return 0;
}

struct Comm {
  ncclComm_t nccl;
  int rank, world;
};

#define SLV_NCCL(call)                                                                                    \
  do {                                                                                                    \
    ncclResult_t r__ = (call);                                                                            \
    if (r__ != ncclSuccess) {                                                                             \
      snprintf(::slv::g_err, sizeof(::slv::g_err), "%s: %s -> %s", __func__, #call, g_api.GetErrorString(r__)); \
      return -6;                                                                                          \
    }                                                                                                     \
  } while (0)

// internal entry used by the fused SyncBN / Sinkhorn-Knopp calls of the other translation units
int comm_allreduce_sum_f64(void* comm, double* buf, size_t n, hipStream_t st) {
  Comm* c = (Comm*)comm;
  if (!c) return fail(-2, "%s: null communicator", "comm_allreduce_sum_f64");
  SLV_NCCL(g_api.AllReduce(buf, buf, n, ncclFloat64, ncclSum, c->nccl, st));
  return 0;
}
int comm_world(void* comm) { return comm ? ((Comm*)comm)->world : 1; }

}  // namespace slv

extern "C" {

int slv_comm_load(const char* librccl_path) { return slv::load_api(librccl_path); }

const char* slv_comm_library(void) { return slv::g_api.handle ? slv::g_api.path : ""; }

int slv_comm_unique_id(void* id_out_128) {
  using namespace slv;
  SLV_CHECK_ARG(id_out_128, "null pointer");
  if (int rc = load_api(nullptr)) return rc;
  ncclUniqueId id;
  SLV_NCCL(g_api.GetUniqueId(&id));
  memcpy(id_out_128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

int slv_comm_init(slv_comm_t* comm_out, const void* unique_id_128, int rank, int world) {
  using namespace slv;
  SLV_CHECK_ARG(comm_out && unique_id_128 && world > 0 && rank >= 0 && rank < world, "bad argument");
  if (int rc = load_api(nullptr)) return rc;
  ncclUniqueId id;
  memcpy(id.internal, unique_id_128, NCCL_UNIQUE_ID_BYTES);
  Comm* c = new Comm{nullptr, rank, world};
  ncclResult_t r = g_api.CommInitRank(&c->nccl, world, id, rank);      // collective: every rank of the id calls it
  if (r != ncclSuccess) {
    snprintf(g_err, sizeof(g_err), "slv_comm_init: ncclCommInitRank -> %s", g_api.GetErrorString(r));
    delete c;
    return -6;
  }
  *comm_out = (slv_comm_t)c;
  return 0;
}

int slv_comm_destroy(slv_comm_t comm) {
  using namespace slv;
  if (!comm) return 0;
  Comm* c = (Comm*)comm;
  ncclResult_t r = g_api.CommDestroy(c->nccl);
  delete c;
  if (r != ncclSuccess) return fail(-6, "slv_comm_destroy: ncclCommDestroy failed: %s", g_api.GetErrorString(r));
  return 0;
}

int slv_comm_abort(slv_comm_t comm) {
  using namespace slv;
  if (!comm) return 0;
  Comm* c = (Comm*)comm;
  ncclResult_t r = g_api.CommAbort ? g_api.CommAbort(c->nccl) : g_api.CommDestroy(c->nccl);
  delete c;
  if (r != ncclSuccess) return fail(-6, "slv_comm_abort: %s", g_api.GetErrorString(r));
  return 0;
}

int slv_comm_async_error(slv_comm_t comm) {
  using namespace slv;
  SLV_CHECK_ARG(comm, "null communicator");
  Comm* c = (Comm*)comm;
  if (!g_api.CommGetAsyncError) return 0;
  ncclResult_t st = ncclSuccess;
  SLV_NCCL(g_api.CommGetAsyncError(c->nccl, &st));
  if (st != ncclSuccess && st != ncclInProgress) return fail(-6, "slv_comm_async_error: %s", g_api.GetErrorString(st));
  return 0;
}

int32_t slv_comm_count(slv_comm_t comm) {
  using namespace slv;
  if (!comm) return 0;
  Comm* c = (Comm*)comm;
  if (!g_api.CommCount) return -1;
  int n = 0;
  if (g_api.CommCount(c->nccl, &n) != ncclSuccess) return -1;
  return n;
}

int32_t slv_comm_rank(slv_comm_t comm) { return comm ? ((slv::Comm*)comm)->rank : 0; }
int32_t slv_comm_world(slv_comm_t comm) { return comm ? ((slv::Comm*)comm)->world : 1; }

int slv_comm_allreduce_f64(slv_comm_t comm, double* buf, int64_t n, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(comm && buf && n > 0, "bad argument");
  return comm_allreduce_sum_f64(comm, buf, (size_t)n, (hipStream_t)stream);
}

int slv_comm_allreduce_f32(slv_comm_t comm, float* buf, int64_t n, int average, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(comm && buf && n > 0, "bad argument");
  Comm* c = (Comm*)comm;
  SLV_NCCL(g_api.AllReduce(buf, buf, (size_t)n, ncclFloat32, average ? ncclAvg : ncclSum, c->nccl, (hipStream_t)stream));
  return 0;
}

int slv_comm_allreduce_i64(slv_comm_t comm, int64_t* buf, int64_t n, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(comm && buf && n > 0, "bad argument");
  Comm* c = (Comm*)comm;
  SLV_NCCL(g_api.AllReduce(buf, buf, (size_t)n, ncclInt64, ncclSum, c->nccl, (hipStream_t)stream));
  return 0;
}

int slv_comm_allgather(slv_comm_t comm, const void* send, void* recv, int64_t bytes_per_rank, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(comm && send && recv && bytes_per_rank > 0, "bad argument");
  Comm* c = (Comm*)comm;
  SLV_NCCL(g_api.AllGather(send, recv, (size_t)bytes_per_rank, ncclInt8, c->nccl, (hipStream_t)stream));
  return 0;
}

int slv_comm_broadcast(slv_comm_t comm, void* buf, int64_t bytes, int root, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(comm && buf && bytes > 0 && root >= 0, "bad argument");
  Comm* c = (Comm*)comm;
  SLV_CHECK_ARG(root < c->world, "root outside the communicator");
  SLV_NCCL(g_api.Broadcast(buf, buf, (size_t)bytes, ncclInt8, root, c->nccl, (hipStream_t)stream));
  return 0;
}

}  // extern "C"

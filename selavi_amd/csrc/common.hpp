// Shared helpers for the selavi_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace slv {

// ---- error plumbing: C-ABI entry points return 0 / negative code, message is thread local
extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, const char* a = "") {
  snprintf(g_err, sizeof(g_err), fmt, a);
  return code;
}

// An error another library (or an earlier asynchronous launch) left in this thread's HIP state must not be blamed on
// THIS entry point's launches (SLV_LAUNCH_CHECK reads hipGetLastError), so the entry clears it -- but not silently: the
// first few are reported on stderr with the entry point that found them, and counted (slv_stale_hip_errors()).
extern int g_stale_hip_errors;
inline void note_pending_hip_error(const char* fn) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return;
  if (g_stale_hip_errors++ < 8)
    fprintf(stderr, "libselavi_hip: %s found a pending HIP error left by earlier work: %s\n", fn, hipGetErrorString(e));
}
#define SLV_CHECK_ARG(cond, msg)                                              \
  do {                                                                        \
    ::slv::note_pending_hip_error(__func__);                                  \
    if (!(cond)) return ::slv::fail(-2, "%s: bad argument: " msg, __func__);  \
  } while (0)

inline int launch_check(const char* fn) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  snprintf(g_err, sizeof(g_err), "%s: kernel launch failed: %s", fn, hipGetErrorString(e));
  return -3;
}
#define SLV_LAUNCH_CHECK()                          \
  do {                                              \
    int rc__ = ::slv::launch_check(__func__);       \
    if (rc__) return rc__;                          \
  } while (0)

#define SLV_HIP(call)                                                                     \
  do {                                                                                    \
    hipError_t e__ = (call);                                                              \
    if (e__ != hipSuccess) {                                                              \
      snprintf(::slv::g_err, sizeof(::slv::g_err), "%s: %s -> %s", __func__, #call,       \
               hipGetErrorString(e__));                                                   \
      return -4;                                                                          \
    }                                                                                     \
  } while (0)

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---- wave-level butterflies (all 64 lanes end up with the result; fixed order => deterministic)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// RCCL hooks for the fused SyncBN / sharded Sinkhorn-Knopp entry points (csrc/comm.cpp)
int comm_allreduce_sum_f64(void* comm, double* buf, size_t n, hipStream_t st);
int comm_world(void* comm);

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// exact unsigned division by a runtime constant (Granlund-Montgomery round-up form)
struct FastDiv {
  unsigned m, s1, s2, d;
};
inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.s1 = l > 1 ? 1 : l;
  f.s2 = l > 0 ? l - 1 : 0;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv f) {
  const unsigned t = __umulhi(f.m, n);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}


}  // namespace slv

// Sinkhorn-Knopp pseudo-label solver kernels for gfx950 (MI355X).
//
// Replaces the torch fp64 ops of /root/reference/src/sk_utils.py:359-422 (optimize_L_sk_gpu) and
// the per-head softmax64 product of :309-315.  The matrix P (N x K fp64, row-major) is the only
// large object; every kernel here is HBM-bound and is written as a coalesced streaming pass:
// one 64-lane wavefront owns a row at a time (lane l holds columns l, l+64, ...), row sums are
// wave butterflies, column sums are kept per lane in registers and reduced block -> grid in a
// FIXED order (no atomics) so results are bit-reproducible run to run.
//
// The reference reads P twice per iteration (matmul(beta.t(), PS) then matmul(PS, alpha)); here the
// row pass of iteration t also accumulates the column sums iteration t+1 needs, so P is read once
// per iteration: algorithmic traffic = N*K*8 bytes/iteration.
#include "common.hpp"
#include "../../include/selavi_hip.h"

namespace slv {
thread_local char g_err[512] = {0};
int g_stale_hip_errors = 0;

constexpr int SK_NGRP = 8;   // workgroup groups of the fused pass's two-level grid reduction
struct SkCtrl {       // lives at the head of the workspace (64 bytes)
  int counter;        // iterations executed so far (the reference's _counter)
  int done;           // 1 -> loop has terminated, later launches are no-ops
  double err;         // last tested err (init 1e6, sk_utils.py:396)
  // arrival counters of the fused pass (sk_pass_kernel<.., FUSED>): zero between launches (the last arriver resets them)
  unsigned grp_arrive[SK_NGRP];
  unsigned fin_arrive;
  unsigned pad[3];
};
static_assert(sizeof(SkCtrl) == 64, "SkCtrl is the 64-byte head of the workspace");

struct SkWs {
  SkCtrl* ctrl;
  double* alpha;    // Kp
  double* s;        // Kp + 64 ; s[K] carries the err partial so one all-reduce moves both
  double* partial;  // grid * Kp
  double* errp;     // grid
  double* gpart;    // SK_NGRP * Kp + SK_NGRP: per-group sums (+ the groups' err sums) of the fused pass
  int Kp;
};

// Row width of the per-block partial sums: 64 x the KJ the kernels are DISPATCHED with (SK_DISPATCH_KJ rounds
// ceil(K / 64) up to an instantiated value: 3 -> 4, 6 -> 7, 9..11 -> 12), not 64 x ceil(K / 64): block_reduce_cols
// writes KJ * 64 entries per block, and with the narrower rows a block overwrote the first columns of its neighbour's
// partials for 128 < K <= 192, 320 < K <= 384, 512 < K <= 704 (found by the K = 700 edge case going through 2 000
// iterations; the reference's K = 309 and 400 have KJ = ceil(K / 64) and were never affected).
static inline int kj_dispatched(int K) {
  const int kj = (K + 63) / 64;
  return kj <= 2 ? kj : kj <= 4 ? 4 : kj <= 5 ? 5 : kj <= 7 ? 7 : kj <= 8 ? 8 : 12;
}
static inline int kpad(int K) { return kj_dispatched(K) * 64; }

static inline SkWs carve(void* ws, int K, int grid) {
  SkWs w;
  char* p = (char*)ws;
  w.Kp = kpad(K);
  w.ctrl = (SkCtrl*)p;
  p += sizeof(SkCtrl);
  w.alpha = (double*)p;
  p += sizeof(double) * w.Kp;
  w.s = (double*)p;
  p += sizeof(double) * (w.Kp + 64);
  w.partial = (double*)p;
  p += sizeof(double) * (size_t)grid * w.Kp;
  w.errp = (double*)p;
  p += sizeof(double) * (size_t)grid;
  w.gpart = (double*)p;
  return w;
}

constexpr int SK_THREADS = 512;           // 8 waves per workgroup
constexpr int SK_WAVES = SK_THREADS / 64;
#ifndef SK_PASS_ROWS
#define SK_PASS_ROWS 4                    // rows a wave keeps in flight in the fused pass
#endif

// torch.argmax semantics: NaN counts as the maximum, first index wins ties.
__device__ __forceinline__ bool beats(double a, int ia, double b, int ib) {
  const bool an = a != a, bn = b != b;
  if (an != bn) return an;
  if (!an && a != b) return a > b;
  return ia < ib;
}

// ------------------------------------------------------------------------------------------
// softmax64(lv) * softmax64(la) [^power]   (sk_utils.py:309-315 + :391), one wave per row
// ------------------------------------------------------------------------------------------
template <int KJ, bool TWO>
__global__ __launch_bounds__(SK_THREADS) void sk_prepare_kernel(const float* __restrict__ lv,
                                                               const float* __restrict__ la,
                                                               double* __restrict__ P, int64_t N,
                                                               int K, double power, int do_pow,
                                                               int64_t* __restrict__ labels = nullptr) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * SK_WAVES + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * SK_WAVES;
  for (int64_t i = wave0; i < N; i += nwaves) {
    double xv[KJ], xa[KJ];
    double mv = -INFINITY, ma = -INFINITY;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      const int k = lane + 64 * j;
      xv[j] = (k < K) ? (double)lv[i * K + k] : -INFINITY;
      mv = fmax(mv, xv[j]);
      if (TWO) {
        xa[j] = (k < K) ? (double)la[i * K + k] : -INFINITY;
        ma = fmax(ma, xa[j]);
      }
    }
    mv = wave_max(mv);
    if (TWO) ma = wave_max(ma);
    double sv = 0.0, sa = 0.0;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      const int k = lane + 64 * j;
      xv[j] = (k < K) ? exp(xv[j] - mv) : 0.0;
      sv += xv[j];
      if (TWO) {
        xa[j] = (k < K) ? exp(xa[j] - ma) : 0.0;
        sa += xa[j];
      }
    }
    sv = wave_sum(sv);
    if (TWO) sa = wave_sum(sa);
    double best = 0.0;
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      const int k = lane + 64 * j;
      if (k < K) {
        double p = xv[j] / sv;
        if (TWO) p = p * (xa[j] / sa);
        if (do_pow) p = pow(p, power);
        if (P) P[i * K + k] = p;
        if (bi == 0x7fffffff || beats(p, k, best, bi)) {
          best = p;
          bi = k;
        }
      }
    }
    if (labels) {                            // clustering_metrics.py:121-126: PS_av.argmax(1)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || beats(ov, oi, best, bi))) {
          best = ov;
          bi = oi;
        }
      }
      if (lane == 0) labels[i] = bi;
    }
  }
}

// counts[pred[i]][target[i]] += 1 (clustering_metrics.py:41-56 builds this with K*K masked sums over N).
// Integer atomics commute, so the result does not depend on the schedule.
__global__ __launch_bounds__(256) void contingency_kernel(const int64_t* __restrict__ pred,
                                                          const int64_t* __restrict__ target, int64_t N, int K1,
                                                          int K2, unsigned long long* __restrict__ counts,
                                                          int* __restrict__ bad) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < N; i += stride) {
    const int64_t a = pred[i], b = target[i];
    if (a < 0 || a >= K1 || b < 0 || b >= K2) {
      *bad = 1;
      continue;
    }
    atomicAdd(&counts[a * K2 + b], 1ull);
  }
}

__global__ __launch_bounds__(256) void sk_pow_kernel(double* __restrict__ P, int64_t n, double power) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) P[i] = pow(P[i], power);
}

// ------------------------------------------------------------------------------------------
// block-level fixed-order reduction of the per-lane column accumulators -> partial[block][k]
// ------------------------------------------------------------------------------------------
// Agent-coherent accesses without a fence: relaxed atomics at agent scope compile to plain global_load / global_store with
// sc1 (write-through to / read from the device's coherence point, past the per-XCD L2).  A release / acquire FENCE at agent
// scope is buffer_wbl2 + buffer_inv over the whole L2 of the XCD: measured at ~0.34 us per workgroup and serialised
// (512 workgroups: 81 -> 255 us per iteration, profiles/r06_sk_fused_ab.txt) -- the fused tail below orders its few
// coherent stores with s_waitcnt vmcnt(0) + the arrival atomic instead.
__device__ __forceinline__ void coh_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double coh_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int KJ, bool COH = false>
__device__ __forceinline__ void block_reduce_cols(const double (&acc)[KJ], double e, SkWs w,
                                                  double* sh /* SK_WAVES*KJ*64 + SK_WAVES */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < KJ; ++j) sh[(wave * KJ + j) * 64 + lane] = acc[j];
  double* she = sh + SK_WAVES * KJ * 64;
  if (lane == 0) she[wave] = e;
  __syncthreads();
  for (int idx = threadIdx.x; idx < KJ * 64; idx += SK_THREADS) {
    double v = 0.0;
#pragma unroll
    for (int wv = 0; wv < SK_WAVES; ++wv) v += sh[wv * KJ * 64 + idx];
    // idx = j*64 + lane  <->  column k = lane + 64*j
    if constexpr (COH) coh_store(&w.partial[(size_t)blockIdx.x * w.Kp + idx], v);
    else w.partial[(size_t)blockIdx.x * w.Kp + idx] = v;
  }
  if (threadIdx.x == 0) {
    double v = 0.0;
#pragma unroll
    for (int wv = 0; wv < SK_WAVES; ++wv) v += she[wv];
    if constexpr (COH) coh_store(&w.errp[blockIdx.x], v);
    else w.errp[blockIdx.x] = v;
  }
}

// weighted column sums: partial[b][k] = sum_{rows of block b} weight_i * P[i][k]
template <int KJ>
__global__ __launch_bounds__(SK_THREADS) void sk_colsum_kernel(const double* __restrict__ P,
                                                              const double* __restrict__ wgt,
                                                              double wconst, int64_t N, int K,
                                                              SkWs w, int64_t rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc[KJ];
#pragma unroll
  for (int j = 0; j < KJ; ++j) acc[j] = 0.0;
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t row1 = (row0 + rows_per_block < N) ? row0 + rows_per_block : N;
  for (int64_t i = row0 + wave; i < row1; i += SK_WAVES) {
    const double b = wgt ? wgt[i] : wconst;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      const int k = lane + 64 * j;
      if (k < K) acc[j] += b * P[i * K + k];
    }
  }
  block_reduce_cols<KJ>(acc, 0.0, w, sh);
}

// ------------------------------------------------------------------------------------------
// The fused iteration pass  (sk_utils.py:401-405, one read of P)
//   t_i = sum_k P_ik alpha_k ; beta'_i = c / t_i ; err += |beta_i / beta'_i - 1| (tested iters)
//   s'_k += beta'_i P_ik  (column sums the NEXT iteration's alpha needs)
// ------------------------------------------------------------------------------------------
// FUSED: the grid-level reduction (and, with r != nullptr, the alpha update + loop control of sk_update_kernel) runs in the
// TAIL of this launch instead of in two more launches: the workgroups are cut into SK_NGRP groups of consecutive indices; the
// last workgroup of a group to arrive (a device-scope atomic behind a release fence) sums that group's partials in INDEX
// order, the last group to finish sums the group sums in group order -- a fixed summation tree whatever the arrival order,
// so the result is bit-reproducible run to run like the separate sk_local_reduce_kernel's (its tree differs: results agree to
// fp64 rounding, labels / iteration counts / cost are pinned by the same goldens).  The idea: seven of the eight group sums are
// formed while other workgroups still stream, only one group sum + the final stage stay exposed.  MEASURED SLOWER than the
// two extra launches (see sk_fused() below): opt-in, SELAVI_SK_FUSED=1.
// NT: the loads of P carry the non-temporal hint -- P is read exactly once per iteration; beyond the 256 MB Infinity Cache
// (one GPU at the VGG-Sound / Kinetics sizes) nothing of it survives to the next iteration anyway, and streaming past the
// caches' allocation leaves them to beta / the partials.  A row shard that FITS the cache (8-way sharding: 52.8 MB) must stay
// resident from iteration to iteration: the hint is off there (slv_sk_pass picks by the shard's bytes).  Measured, whole
// iterations (pass + grid reduce + update), 200 iterations that never terminate, same box, same call
// (profiles/r06_sk_ab_honest.txt): N = 170 752, K = 309, grid 512: 88.9-89.6 -> 77.6-78.0 us; N = 230 976, K = 400, grid 256:
// 146.6 -> 129.0 us.  Tried on top and dropped: eight rows per wave in flight instead of four (slower); walking a workgroup's
// rows back and forth on alternate iterations so that an iteration starts on the rows its predecessor finished with (the
// Infinity Cache holds 60 % of P): slower, the reversed address stream costs more than the hits return
// (profiles/r06_sk_alt_ab.txt -- relative figures only: that call averaged over launches behind the solver's fixed point).
template <int KJ, int ROWS, bool FUSED = false, bool NT = false>
__global__ __launch_bounds__(SK_THREADS) void sk_pass_kernel(const double* __restrict__ P,
                                                            int64_t N, int K, double c,
                                                            double* __restrict__ beta, SkWs w,
                                                            int64_t rows_per_block, const double* __restrict__ r = nullptr,
                                                            double tol = 0.0, int max_iter = 0) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  if (w.ctrl->done) return;
  const bool check = (w.ctrl->counter % 10) == 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double al[KJ], acc[KJ];
#pragma unroll
  for (int j = 0; j < KJ; ++j) {
    const int k = lane + 64 * j;
    al[j] = (k < K) ? w.alpha[k] : 0.0;
    acc[j] = 0.0;
  }
  double e = 0.0;
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t row1 = (row0 + rows_per_block < N) ? row0 + rows_per_block : N;
  for (int64_t i = row0 + (int64_t)wave * ROWS; i < row1; i += (int64_t)SK_WAVES * ROWS) {
    double p[ROWS][KJ];
    double t[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int64_t ii = i + r;
      const bool rv = ii < row1;
      t[r] = 0.0;
#pragma unroll
      for (int j = 0; j < KJ; ++j) {
        const int k = lane + 64 * j;
        p[r][j] = (rv && k < K) ? (NT ? __builtin_nontemporal_load(&P[ii * K + k]) : P[ii * K + k]) : 0.0;
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
#pragma unroll
      for (int j = 0; j < KJ; ++j) t[r] += p[r][j] * al[j];
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) t[r] = wave_sum(t[r]);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int64_t ii = i + r;
      if (ii < row1) {
        const double bn = c / t[r];
        if (check) e += fabs(beta[ii] / bn - 1.0);
        if (lane == 0) beta[ii] = bn;
#pragma unroll
        for (int j = 0; j < KJ; ++j) acc[j] += bn * p[r][j];
      }
    }
  }
  block_reduce_cols<KJ, FUSED>(acc, e, w, sh);
  if constexpr (FUSED) {
    __shared__ int s_role;
    const int grid = (int)gridDim.x, gs = (grid + SK_NGRP - 1) / SK_NGRP, ngrp = (grid + gs - 1) / gs;
    const int grp = (int)blockIdx.x / gs, b0 = grp * gs, b1 = (b0 + gs < grid) ? b0 + gs : grid;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's coherent stores have reached the coherence point ...
    __syncthreads();                                      // ... every thread's, before thread 0 announces the row
    if (threadIdx.x == 0) s_role = (atomicAdd(&w.ctrl->grp_arrive[grp], 1u) == (unsigned)(b1 - b0 - 1)) ? 1 : 0;
    __syncthreads();
    if (!s_role) return;
    for (int k = threadIdx.x; k < w.Kp; k += SK_THREADS) {
      double v = 0.0;
      for (int b = b0; b < b1; ++b) v += coh_load(&w.partial[(size_t)b * w.Kp + k]);       // index order
      coh_store(&w.gpart[(size_t)grp * w.Kp + k], v);
    }
    if (threadIdx.x == 0) {
      double v = 0.0;
      for (int b = b0; b < b1; ++b) v += coh_load(&w.errp[b]);
      coh_store(&w.gpart[(size_t)SK_NGRP * w.Kp + grp], v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_role = (atomicAdd(&w.ctrl->fin_arrive, 1u) == (unsigned)(ngrp - 1)) ? 2 : 0;
    __syncthreads();
    if (s_role != 2) return;
    // final stage: s[k] = sum over groups in group order, s[K] = the err sum; then (single-GPU loop) sk_update_kernel's work
    const int cnt = w.ctrl->counter;
    double err = w.ctrl->err;
    double ev = 0.0;
    for (int g = 0; g < ngrp; ++g) ev += coh_load(&w.gpart[(size_t)SK_NGRP * w.Kp + g]);
    int done = 0, newc = cnt;
    if (r) {
      if ((cnt % 10) == 0) err = ev;
      newc = cnt + 1;
      done = (!(err > tol)) || (newc >= max_iter);
    }
    for (int k = threadIdx.x; k < K; k += SK_THREADS) {
      double v = 0.0;
      for (int g = 0; g < ngrp; ++g) v += coh_load(&w.gpart[(size_t)g * w.Kp + k]);
      w.s[k] = v;
      if (r && !done) w.alpha[k] = r[k] / v;
    }
    __syncthreads();                              // everybody has read ctrl before thread 0 rewrites it
    if (threadIdx.x == 0) {
      w.s[K] = ev;
      if (r) {
        w.ctrl->counter = newc;
        w.ctrl->done = done;
        w.ctrl->err = err;
      }
      for (int g = 0; g < SK_NGRP; ++g) w.ctrl->grp_arrive[g] = 0u;      // clean for the next launch
      w.ctrl->fin_arrive = 0u;
    }
  }
}

// grid-level fixed-order reduce: s[k] = sum_b partial[b][k];  s[K] = sum_b errp[b].
// 16 columns x 64 block-groups per workgroup: short per-thread chains (latency bound otherwise:
// the 5-workgroup version took 12 us of a 94 us iteration), fixed-order LDS tree -> deterministic.
__global__ __launch_bounds__(1024) void sk_local_reduce_kernel(SkWs w, int K, int grid, int respect_done) {
  if (respect_done && w.ctrl->done) return;
  __shared__ double sh[64][17];
  __shared__ double she[1024];
  const int kx = threadIdx.x & 15, by = threadIdx.x >> 4;
  const int k = blockIdx.x * 16 + kx;
  double v = 0.0;
  for (int b = by; b < grid; b += 64) v += w.partial[(size_t)b * w.Kp + k];
  sh[by][kx] = v;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) {
    if (by < o) sh[by][kx] += sh[by + o][kx];
    __syncthreads();
  }
  if (by == 0 && k < K) w.s[k] = sh[0][kx];
  if (blockIdx.x == 0) {
    double ev = 0.0;
    for (int b = threadIdx.x; b < grid; b += 1024) ev += w.errp[b];
    she[threadIdx.x] = ev;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) she[threadIdx.x] += she[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) w.s[K] = she[0];
  }
}

// alpha = r / s and loop control  (sk_utils.py:400-401,403-406)
__global__ __launch_bounds__(1024) void sk_update_kernel(const double* __restrict__ r, SkWs w, int K,
                                                        double tol, int max_iter, int first) {
  const int cnt = w.ctrl->counter;
  const int was_done = w.ctrl->done;
  double err = w.ctrl->err;
  int done = was_done, newc = cnt;
  if (!was_done && !first) {
    const bool tested = (cnt % 10) == 0;
    if (tested) err = w.s[K];
    newc = cnt + 1;
    done = (!(err > tol)) || (newc >= max_iter);
  }
  __syncthreads();  // everybody has read ctrl before thread 0 rewrites it
  if (!was_done && !done) {
    for (int k = threadIdx.x; k < K; k += blockDim.x) w.alpha[k] = r[k] / w.s[k];
  }
  if (threadIdx.x == 0 && !was_done) {
    w.ctrl->counter = newc;
    w.ctrl->done = done;
    w.ctrl->err = err;
  }
}

__global__ void sk_begin_kernel(SkWs w, double* __restrict__ beta, int64_t N_local, double b0) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    w.ctrl->counter = 0;
    w.ctrl->done = 0;
    w.ctrl->err = 1e6;
    for (int g = 0; g < SK_NGRP; ++g) w.ctrl->grp_arrive[g] = 0u;
    w.ctrl->fin_arrive = 0u;
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < N_local; i += stride) beta[i] = b0;
}

__global__ void sk_status_kernel(SkWs w, double* __restrict__ out) {
  out[0] = (double)w.ctrl->counter;
  out[1] = (double)w.ctrl->done;
  out[2] = w.ctrl->err;
  out[3] = 0.0;
}

// labels + cost  (sk_utils.py:411-419): argmax_k (P_ik beta_i) alpha_k ; log of the undone value
template <int KJ>
__global__ __launch_bounds__(SK_THREADS) void sk_labels_kernel(const double* __restrict__ P, int64_t N,
                                                              int K, const double* __restrict__ beta,
                                                              SkWs w, int64_t* __restrict__ labels,
                                                              int64_t rows_per_block) {
  __shared__ double she[SK_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double al[KJ];
#pragma unroll
  for (int j = 0; j < KJ; ++j) {
    const int k = lane + 64 * j;
    al[j] = (k < K) ? w.alpha[k] : 0.0;
  }
  double lsum = 0.0;
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t row1 = (row0 + rows_per_block < N) ? row0 + rows_per_block : N;
  for (int64_t i = row0 + wave; i < row1; i += SK_WAVES) {
    const double b = beta[i];
    double best = 0.0, bestx = 0.0;
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      const int k = lane + 64 * j;
      if (k < K) {
        const double v = (P[i * K + k] * b) * al[j];                    // :411-412
        if (bi == 0x7fffffff || beats(v, k, best, bi)) {
          best = v;
          bi = k;
          bestx = (v * (1.0 / al[j])) * (1.0 / b);                      // :416-417
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(best, o, 64);
      const double ox = __shfl_xor(bestx, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      const bool take = (bi == 0x7fffffff) ? (oi != 0x7fffffff)
                                           : (oi != 0x7fffffff && beats(ov, oi, best, bi));
      if (take) {
        best = ov;
        bestx = ox;
        bi = oi;
      }
    }
    if (lane == 0) {
      labels[i] = bi;
      const double lg = log(bestx);                                      // :418
      if (lg == lg) lsum += lg;                                          // nansum
    }
  }
  if (lane == 0) she[wave] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < SK_WAVES; ++q) v += she[q];
    w.errp[blockIdx.x] = v;
  }
}

__global__ __launch_bounds__(512) void sk_sum_errp_kernel(SkWs w, int grid, double* __restrict__ out) {
  __shared__ double she[512];
  double ev = 0.0;
  for (int b = threadIdx.x; b < grid; b += 512) ev += w.errp[b];
  she[threadIdx.x] = ev;
  __syncthreads();
  for (int o = 256; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) she[threadIdx.x] += she[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = she[0];
}

// 64x64 tile of the L1 column-distance matrix over one slice of the rows; thread (ty,tx) owns a 4x4 patch
__global__ __launch_bounds__(256) void sk_l1_cost_kernel(const double* __restrict__ e1, const double* __restrict__ e2,
                                                        int64_t N, int K, double* __restrict__ partial,
                                                        int nsplit) {
  __shared__ double s1[16][64], s2[16][64];
  const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
  const int64_t per = (N + nsplit - 1) / nsplit;
  const int64_t r0 = blockIdx.z * per, r1 = (r0 + per < N) ? r0 + per : N;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (int64_t r = r0; r < r1; r += 16) {
    for (int t = threadIdx.x; t < 16 * 64; t += 256) {
      const int rr = t >> 6, cc = t & 63;
      const bool ok = r + rr < r1;
      s1[rr][cc] = (ok && i0 + cc < K) ? e1[(r + rr) * K + i0 + cc] : 0.0;
      s2[rr][cc] = (ok && j0 + cc < K) ? e2[(r + rr) * K + j0 + cc] : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
      double a_[4], b_[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) a_[a] = s1[rr][ty * 4 + a];
#pragma unroll
      for (int b = 0; b < 4; ++b) b_[b] = s2[rr][tx * 4 + b];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] += fabs(a_[a] - b_[b]);
    }
    __syncthreads();
  }
  // rows beyond r1 contributed |0 - 0| = 0
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = i0 + ty * 4 + a, j = j0 + tx * 4 + b;
      if (i < K && j < K) partial[((size_t)blockIdx.z * K + i) * K + j] = acc[a][b];
    }
}
__global__ void sk_sum_splits_kernel(const double* __restrict__ partial, double* __restrict__ out, size_t n,
                                     int nsplit) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = 0.0;
  for (int s = 0; s < nsplit; ++s) v += partial[(size_t)s * n + i];
  out[i] = v;
}

// ---- KJ dispatch -----------------------------------------------------------------------------
#define SK_DISPATCH_KJ(K, ...)                                             \
  do {                                                                     \
    const int kj__ = ((K) + 63) / 64;                                      \
    if (kj__ <= 1) { constexpr int KJ = 1; __VA_ARGS__; }                         \
    else if (kj__ <= 2) { constexpr int KJ = 2; __VA_ARGS__; }                    \
    else if (kj__ <= 4) { constexpr int KJ = 4; __VA_ARGS__; }                    \
    else if (kj__ <= 5) { constexpr int KJ = 5; __VA_ARGS__; }                    \
    else if (kj__ <= 7) { constexpr int KJ = 7; __VA_ARGS__; }                    \
    else if (kj__ <= 8) { constexpr int KJ = 8; __VA_ARGS__; }                    \
    else if (kj__ <= 12) { constexpr int KJ = 12; __VA_ARGS__; }                  \
    else return ::slv::fail(-2, "%s: K > 768 is not supported (per-lane column registers; the reference's datasets use 309 / 400)", __func__); \
  } while (0)

static inline size_t sh_bytes(int KJ) { return sizeof(double) * (SK_WAVES * KJ * 64 + SK_WAVES); }

// SELAVI_SK_NT: "1" / "0" force the non-temporal loads of the pass on / off; default: on when the shard does not fit the
// Infinity Cache (> 200 MB).
static inline bool sk_pass_nt(int64_t N_local, int K) {
  static const int mode = []() {
    const char* e = getenv("SELAVI_SK_NT");
    return !e ? -1 : (e[0] == '0' ? 0 : 1);
  }();
  if (mode >= 0) return mode == 1;
  return (double)N_local * K * 8.0 > 200e6;
}

}  // namespace slv

using namespace slv;

extern "C" {

int32_t slv_version(void) { return 1; }
const char* slv_last_error(void) { return slv::g_err; }
int32_t slv_stale_hip_errors(void) { return slv::g_stale_hip_errors; }

int slv_device_info(int* cu_count, int* wave_size, char* arch_name, int arch_name_len) {
  int dev = 0;
  SLV_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  SLV_HIP(hipGetDeviceProperties(&p, dev));
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (wave_size) *wave_size = p.warpSize;
  if (arch_name && arch_name_len > 0) {
    strncpy(arch_name, p.gcnArchName, arch_name_len - 1);
    arch_name[arch_name_len - 1] = 0;
  }
  return 0;
}

size_t slv_sk_workspace_bytes(int K, int grid) {
  const int Kp = kpad(K);
  return sizeof(SkCtrl) + sizeof(double) * ((size_t)Kp + (Kp + 64) + (size_t)grid * Kp + grid + (size_t)SK_NGRP * Kp + SK_NGRP);
}

int32_t slv_sk_default_grid(int64_t N, int K) {
  // 2 workgroups of 8 waves per CU on a 256-CU part; never more blocks than 8-row chunks.  Measured per iteration (pass with
  // non-temporal loads + grid reduce + update, 200 iterations, profiles/r06_sk_ab_honest.txt) at N = 170 752, K = 309:
  // 256: 78.4-79.0, 512: 77.6-78.0, 768: 79.1-79.8, 1024: 80.8, 1536: 83, 2048: 88.6 us (the reduce over [grid][K] partials grows
  // with the grid); N = 230 976, K = 400: 256: 129, 512: 136, 768: 132, 1024: 142.  A row shard of an 8-GPU run (21 344 rows,
  // resident in the Infinity Cache): 256: 17.4, 512: 18.6, 768: 20.9 us -- there the reduce is a third of the iteration.
  // SELAVI_SK_GRID overrides (A/B).
  static const int64_t forced = []() {
    const char* e = getenv("SELAVI_SK_GRID");
    return e ? (int64_t)atoi(e) : (int64_t)0;
  }();
  int64_t g = forced > 0 ? forced : ((double)N * K * 8.0 > 200e6 ? 512 : 256);
  const int64_t maxg = (N + 7) / 8;
  if (g > maxg) g = maxg;
  if (g < 1) g = 1;
  return (int)g;
}

double* slv_sk_s_ptr(void* ws, int K, int grid) { return carve(ws, K, grid).s; }
double* slv_sk_alpha_ptr(void* ws, int K, int grid) { return carve(ws, K, grid).alpha; }

int slv_sk_prepare(const float* lv, const float* la, double* P, int64_t N, int K, double power,
                   slv_stream_t stream) {
  SLV_CHECK_ARG(lv && la && P && N >= 0 && K > 0, "null pointer or empty shape");
  if (N == 0) return 0;
  const int grid = (int)((N + SK_WAVES - 1) / SK_WAVES < 4096 ? (N + SK_WAVES - 1) / SK_WAVES : 4096);
  SK_DISPATCH_KJ(K, hipLaunchKernelGGL((sk_prepare_kernel<KJ, true>), dim3(grid), dim3(SK_THREADS), 0,
                                        (hipStream_t)stream, lv, la, P, N, K, power, power != 1.0));
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_av_argmax(const float* lv, const float* la, int64_t N, int K, int64_t* labels, slv_stream_t stream) {
  SLV_CHECK_ARG(lv && la && labels && N >= 0 && K > 0, "null pointer or empty shape");
  if (N == 0) return 0;
  const int grid = (int)((N + SK_WAVES - 1) / SK_WAVES < 4096 ? (N + SK_WAVES - 1) / SK_WAVES : 4096);
  SK_DISPATCH_KJ(K, hipLaunchKernelGGL((sk_prepare_kernel<KJ, true>), dim3(grid), dim3(SK_THREADS), 0,
                                        (hipStream_t)stream, lv, la, (double*)nullptr, N, K, 1.0, 0, labels));
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_contingency(const int64_t* pred, const int64_t* target, int64_t N, int K1, int K2, int64_t* counts,
                    int32_t* bad, slv_stream_t stream) {
  SLV_CHECK_ARG(pred && target && counts && bad && N >= 0 && K1 > 0 && K2 > 0, "null pointer or empty shape");
  SLV_HIP(hipMemsetAsync(counts, 0, sizeof(int64_t) * (size_t)K1 * K2, (hipStream_t)stream));
  SLV_HIP(hipMemsetAsync(bad, 0, sizeof(int32_t), (hipStream_t)stream));
  if (N == 0) return 0;
  const int grid = (int)((N + 255) / 256 < 2048 ? (N + 255) / 256 : 2048);
  hipLaunchKernelGGL(contingency_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pred, target, N, K1, K2,
                     (unsigned long long*)counts, (int*)bad);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_sk_softmax64(const float* logits, double* P, int64_t N, int K, slv_stream_t stream) {
  SLV_CHECK_ARG(logits && P && N >= 0 && K > 0, "null pointer or empty shape");
  if (N == 0) return 0;
  const int grid = (int)((N + SK_WAVES - 1) / SK_WAVES < 4096 ? (N + SK_WAVES - 1) / SK_WAVES : 4096);
  SK_DISPATCH_KJ(K, hipLaunchKernelGGL((sk_prepare_kernel<KJ, false>), dim3(grid), dim3(SK_THREADS), 0,
                                        (hipStream_t)stream, logits, (const float*)nullptr, P, N, K, 1.0,
                                        0));
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_sk_pow(double* P, int64_t count, double power, slv_stream_t stream) {
  SLV_CHECK_ARG(P && count >= 0, "null pointer");
  if (count == 0) return 0;
  int64_t blocks = (count + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(sk_pow_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, P, count, power);
  SLV_LAUNCH_CHECK();
  return 0;
}

static int launch_colsum(const double* P, const double* wgt, double wconst, int64_t N, int K, SkWs w,
                         int grid, hipStream_t st) {
  const int64_t rpb = (N + grid - 1) / grid;
  SK_DISPATCH_KJ(K, hipLaunchKernelGGL((sk_colsum_kernel<KJ>), dim3(grid), dim3(SK_THREADS), sh_bytes(KJ),
                                        st, P, wgt, wconst, N, K, w, rpb));
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_sk_colsum(const double* P, const double* row_weight, int64_t N, int K, double* out, void* ws,
                  int grid, slv_stream_t stream) {
  SLV_CHECK_ARG(P && out && ws && N > 0 && K > 0 && grid > 0, "null pointer or empty shape");
  SkWs w = carve(ws, K, grid);
  int rc = launch_colsum(P, row_weight, 1.0, N, K, w, grid, (hipStream_t)stream);
  if (rc) return rc;
  hipLaunchKernelGGL(sk_local_reduce_kernel, dim3(w.Kp / 16), dim3(1024), 0, (hipStream_t)stream, w, K, grid,
                     0);
  SLV_LAUNCH_CHECK();
  SLV_HIP(hipMemcpyAsync(out, w.s, sizeof(double) * K, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

int slv_sk_begin(const double* P, int64_t N_local, int64_t N_global, int K, double* beta, void* ws,
                 int grid, slv_stream_t stream) {
  SLV_CHECK_ARG(P && beta && ws && N_local > 0 && N_global >= N_local && K > 0 && grid > 0,
                "null pointer or empty shape");
  SkWs w = carve(ws, K, grid);
  const double b0 = 1.0 / (double)N_global;  // sk_utils.py:390
  hipLaunchKernelGGL(sk_begin_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, w, beta, N_local, b0);
  SLV_LAUNCH_CHECK();
  // s_0 = beta_0^T P   (the first matmul(beta.t(), PS) of sk_utils.py:401)
  return launch_colsum(P, nullptr, b0, N_local, K, w, grid, (hipStream_t)stream);
}

int slv_sk_pass(const double* P, int64_t N_local, int64_t N_global, int K, double* beta, void* ws,
                int grid, slv_stream_t stream) {
  SLV_CHECK_ARG(P && beta && ws && N_local > 0 && K > 0 && grid > 0, "null pointer or empty shape");
  SkWs w = carve(ws, K, grid);
  const int64_t rpb = (N_local + grid - 1) / grid;
  const double c = 1.0 / (double)N_global;  // sk_utils.py:395
  if (sk_pass_nt(N_local, K)) {
    SK_DISPATCH_KJ(K, hipLaunchKernelGGL((sk_pass_kernel<KJ, SK_PASS_ROWS, false, true>), dim3(grid), dim3(SK_THREADS), sh_bytes(KJ),
                                          (hipStream_t)stream, P, N_local, K, c, beta, w, rpb));
  } else {
    SK_DISPATCH_KJ(K, hipLaunchKernelGGL((sk_pass_kernel<KJ, SK_PASS_ROWS>), dim3(grid), dim3(SK_THREADS), sh_bytes(KJ),
                                          (hipStream_t)stream, P, N_local, K, c, beta, w, rpb));
  }
  SLV_LAUNCH_CHECK();
  return 0;
}

/* pass + local reduce in one host call: the sharded loop is host-enqueue bound (a pass over 1/8 of the rows takes ~11 us) */
int slv_sk_pass_reduce(const double* P, int64_t N_local, int64_t N_global, int K, double* beta, void* ws, int grid,
                       slv_stream_t stream) {
  const int rc = slv_sk_pass(P, N_local, N_global, K, beta, ws, grid, stream);
  return rc ? rc : slv_sk_local_reduce(K, ws, grid, stream);
}

// SELAVI_SK_FUSED=1 (an experiment, OFF by default): one launch per iteration -- measured SLOWER than the three launches it
// replaces on this chip (profiles/r06_sk_fused_ab*.txt, N = 170 752, K = 309, grid 512): 84-87 us per iteration as pass +
// grid reduce + update; 255-268 us with the tail behind agent-scope fences (buffer_wbl2 / buffer_inv of the XCD's L2 per
// workgroup, serialised: + 0.34 us per workgroup); 99-103 us with coherent (sc1) stores / loads and no fence -- the 512 arrival
// atomics on 8 counters at the device's coherence point cost more than the two dispatch gaps they save, and the cost grows
// with the grid (2 048 workgroups: 203 us).  Kept as the A/B of that measurement; results are identical to the default path's
// up to fp64 rounding of a different (equally fixed) summation tree (tests/test_sk_gpu.py runs both).
static bool sk_fused() {
  static const bool on = []() {
    const char* e = getenv("SELAVI_SK_FUSED");
    return e && e[0] == '1';
  }();
  return on;
}

static int launch_pass_fused(const double* P, int64_t N_local, int64_t N_global, int K, double* beta, const double* r, double tol,
                             int max_iter, void* ws, int grid, hipStream_t stream) {
  SkWs w = carve(ws, K, grid);
  const int64_t rpb = (N_local + grid - 1) / grid;
  const double c = 1.0 / (double)N_global;  // sk_utils.py:395
  SK_DISPATCH_KJ(K, hipLaunchKernelGGL((sk_pass_kernel<KJ, SK_PASS_ROWS, true>), dim3(grid), dim3(SK_THREADS), sh_bytes(KJ), stream,
                                        P, N_local, K, c, beta, w, rpb, r, tol, max_iter));
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_sk_iterate(const double* P, int64_t N, int K, double* beta, const double* r, double tol,
                   int max_iter, int n_iters, void* ws, int grid, slv_stream_t stream) {
  SLV_CHECK_ARG(P && beta && r && ws && N > 0 && K > 0 && grid > 0, "null pointer or empty shape");
  if (sk_fused()) {
    // one launch per iteration: pass + grid reduce + alpha update / loop control in its tail (sk_pass_kernel<.., FUSED>)
    for (int it = 0; it < n_iters; ++it) {
      const int rc = launch_pass_fused(P, N, N, K, beta, r, tol, max_iter, ws, grid, (hipStream_t)stream);
      if (rc) return rc;
    }
    return 0;
  }
  // n_iters x (pass, local_reduce, update) enqueued from C, no host sync
  for (int it = 0; it < n_iters; ++it) {
    int rc = slv_sk_pass(P, N, N, K, beta, ws, grid, stream);
    if (rc) return rc;
    rc = slv_sk_local_reduce(K, ws, grid, stream);
    if (rc) return rc;
    rc = slv_sk_update(r, K, tol, max_iter, 0, ws, grid, stream);
    if (rc) return rc;
  }
  return 0;
}

// Row-sharded multi-GPU loop (sk_utils.py:287-329 re-designed: every rank keeps its rows): n_iters x (pass, local
// reduce, all-reduce of the K column sums + err over RCCL, update) enqueued from ONE host call on ONE stream -- the
// K-vector exchange sits between two kernels of the same stream, no process-group stream, no event hops, no Python.
int slv_sk_iterate_sharded(slv_comm_t comm, const double* P, int64_t N_local, int64_t N_global, int K, double* beta,
                           const double* r, double tol, int max_iter, int n_iters, void* ws, int grid,
                           slv_stream_t stream) {
  SLV_CHECK_ARG(comm, "null communicator");
  SLV_CHECK_ARG(P && beta && r && ws && N_local > 0 && K > 0 && grid > 0, "null pointer or empty shape");
  for (int it = 0; it < n_iters; ++it) {
    int rc;
    if (sk_fused()) {     // the local column sums come out of the pass's tail (no update there: the all-reduce sits in between)
      rc = launch_pass_fused(P, N_local, N_global, K, beta, nullptr, 0.0, 0, ws, grid, (hipStream_t)stream);
      if (rc) return rc;
    } else {
      rc = slv_sk_pass(P, N_local, N_global, K, beta, ws, grid, stream);
      if (rc) return rc;
      rc = slv_sk_local_reduce(K, ws, grid, stream);
      if (rc) return rc;
    }
    rc = slv::comm_allreduce_sum_f64(comm, slv_sk_s_ptr(ws, K, grid), (size_t)K + 1, (hipStream_t)stream);
    if (rc) return rc;
    rc = slv_sk_update(r, K, tol, max_iter, 0, ws, grid, stream);
    if (rc) return rc;
  }
  return 0;
}

int slv_sk_local_reduce(int K, void* ws, int grid, slv_stream_t stream) {
  SLV_CHECK_ARG(ws && K > 0 && grid > 0, "null pointer or empty shape");
  SkWs w = carve(ws, K, grid);
  hipLaunchKernelGGL(sk_local_reduce_kernel, dim3(w.Kp / 16), dim3(1024), 0, (hipStream_t)stream, w, K, grid,
                     1);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_sk_update(const double* r, int K, double tol, int max_iter, int first, void* ws, int grid,
                  slv_stream_t stream) {
  SLV_CHECK_ARG(r && ws && K > 0 && grid > 0, "null pointer or empty shape");
  SkWs w = carve(ws, K, grid);
  hipLaunchKernelGGL(sk_update_kernel, dim3(1), dim3(K < 1024 ? ((K + 63) / 64) * 64 : 1024), 0,
                     (hipStream_t)stream, r, w, K, tol, max_iter, first);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_sk_status(void* ws, int K, int grid, double* host_out, slv_stream_t stream) {
  SLV_CHECK_ARG(ws && host_out, "null pointer");
  SkWs w = carve(ws, K, grid);
  // the status words are staged through s[K+1..K+4] so the copy is one small D2H
  double* stage = w.s + K + 8;
  hipLaunchKernelGGL(sk_status_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, w, stage);
  SLV_LAUNCH_CHECK();
  SLV_HIP(hipMemcpyAsync(host_out, stage, 4 * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream));
  return 0;
}

// match_order (sk_utils.py:424-467): C[i][j] = sum_n |e1[n][i] - e2[n][j]| -- the whole hill-climb then
// runs on this K x K table instead of 100k synced column reductions.  partial [S][K][K], fixed order.
int slv_sk_l1_cost_matrix(const double* e1, const double* e2, int64_t N, int K, double* partial, int nsplit,
                          double* out, slv_stream_t stream) {
  SLV_CHECK_ARG(e1 && e2 && partial && out && N > 0 && K > 0 && nsplit > 0, "bad argument");
  const int T = (K + 63) / 64;
  hipLaunchKernelGGL(sk_l1_cost_kernel, dim3(T, T, nsplit), dim3(256), 0, (hipStream_t)stream, e1, e2, N, K,
                     partial, nsplit);
  SLV_LAUNCH_CHECK();
  const size_t kk = (size_t)K * K;
  hipLaunchKernelGGL(sk_sum_splits_kernel, dim3((unsigned)((kk + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     partial, out, kk, nsplit);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_sk_labels(const double* P, int64_t N_local, int K, const double* beta, void* ws, int grid,
                  int64_t* labels, double* logsum_out, slv_stream_t stream) {
  SLV_CHECK_ARG(P && beta && ws && labels && logsum_out && N_local > 0 && K > 0 && grid > 0,
                "null pointer or empty shape");
  SkWs w = carve(ws, K, grid);
  const int64_t rpb = (N_local + grid - 1) / grid;
  SK_DISPATCH_KJ(K, hipLaunchKernelGGL((sk_labels_kernel<KJ>), dim3(grid), dim3(SK_THREADS), 0,
                                        (hipStream_t)stream, P, N_local, K, beta, w, labels, rpb));
  SLV_LAUNCH_CHECK();
  hipLaunchKernelGGL(sk_sum_errp_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, w, grid, logsum_out);
  SLV_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

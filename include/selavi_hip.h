/* selavi_hip.h -- C ABI of libselavi_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the SeLaVi data-parallel hot path.  The reference
 * (facebookresearch/selavi) is 100 % Python with no FFI of its own; its boundary for this path
 * is the Python API of model.py / utils.py / src/sk_utils.py.  The host side of
 * this build (the selavi_amd Python package) mirrors that Python API and reaches the device ONLY through the
 * entry points declared here (ctypes stubs: INTEGRATION.md).  Each group below cites the
 * reference lines whose ATen/cuDNN/cuBLAS work it replaces.
 *
 * Conventions
 *   - plain pointers and sizes; no torch types.  All pointers are DEVICE pointers owned by the
 *     caller unless marked "host".  The library never allocates outputs and owns no streams.
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) of the CURRENT device and returns without synchronising unless stated.
 *   - functions returning `int` return a STATUS: 0 on success, negative on error, and
 *     slv_last_error() gives a thread-local text.  Functions that return a VALUE are declared
 *     with int32_t / size_t / pointer return types (the ctypes binding keys on this).
 *   - one process per GPU; calls are re-entrant and thread-safe (no global mutable state).
 *   - tensors are dense row-major ("contiguous" in torch terms); activations are N,C,T,H,W
 *     (2-D audio tensors are the T==1 case).
 */
#ifndef SELAVI_HIP_H_
#define SELAVI_HIP_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* slv_stream_t; /* hipStream_t */

/* ---------------------------------------------------------------- library ------------------ */
int32_t slv_version(void);                /* ABI version, bumps on any signature change          */
const char* slv_last_error(void);      /* thread-local, valid until the next failing call      */
int slv_device_info(int* cu_count, int* wave_size, char* arch_name, int arch_name_len);

/* ---------------------------------------------------------------- Sinkhorn-Knopp ------------
 * Replaces the torch fp64 ops of src/sk_utils.py:
 *   slv_sk_prepare      <- softmax(dtype=float64) x2, torch.mul(out=), PS.pow_()  (:309-315,:391)
 *   slv_sk_pow          <- PS.pow_(0.5*lamb)                                      (:391)
 *   slv_sk_colsum       <- PS.sum(0) / matmul(beta.t(), PS)                       (:368,:401)
 *   slv_sk_begin/_pass/_local_reduce/_update <- the while-loop body               (:400-406)
 *   slv_sk_labels       <- the in-place rescale, argmax, gather+log+nansum        (:411-419)
 * P is N x K fp64 row-major.  `ws` is a device workspace of slv_sk_workspace_bytes(K, grid).
 * Iteration protocol (all on `stream`, no host sync inside):
 *   slv_sk_begin(...)                       beta = 1/N, s = beta^T P, alpha = r/s, counter = 0
 *   repeat: slv_sk_pass(...)                one fused row pass: t = P alpha, beta' = c/t,
 *                                           err partial (on counter%10==0), s' partial = beta'^T P
 *           slv_sk_local_reduce(...)        ws.s[0..K) = sum of partials, ws.s[K] = err partial sum
 *           (multi-GPU: all-reduce(sum) the K+1 doubles at slv_sk_s_ptr(ws) across ranks)
 *           slv_sk_update(...)              counter += 1; if tested err <= tol or counter == max
 *                                           -> done = 1 (later passes become no-ops);
 *                                           else alpha = r / s
 *   slv_sk_status(...)                      async copy of {counter, done, err} to host memory
 * After done: ws holds the alpha used by the last executed pass and `beta` the last beta'.
 */
size_t slv_sk_workspace_bytes(int K, int grid);
int32_t slv_sk_default_grid(int64_t N, int K);
double* slv_sk_s_ptr(void* ws, int K, int grid);      /* K+1 doubles: column sums + err     */
double* slv_sk_alpha_ptr(void* ws, int K, int grid);  /* K doubles                          */

int slv_sk_prepare(const float* logits_v, const float* logits_a, double* P, int64_t N, int K,
                   double power, slv_stream_t stream);
int slv_sk_softmax64(const float* logits, double* P, int64_t N, int K, slv_stream_t stream);
int slv_sk_pow(double* P, int64_t count, double power, slv_stream_t stream);
int slv_sk_colsum(const double* P, const double* row_weight /* nullable -> 1 */, int64_t N, int K,
                  double* out /* K */, void* ws, int grid, slv_stream_t stream);

int slv_sk_begin(const double* P, int64_t N_local, int64_t N_global, int K, double* beta,
                 void* ws, int grid, slv_stream_t stream);
int slv_sk_pass(const double* P, int64_t N_local, int64_t N_global, int K, double* beta,
                void* ws, int grid, slv_stream_t stream);
int slv_sk_local_reduce(int K, void* ws, int grid, slv_stream_t stream);
int slv_sk_update(const double* r /* K, normalised */, int K, double tol, int max_iter,
                  int first /* 1: right after slv_sk_begin (no counter++) */, void* ws, int grid,
                  slv_stream_t stream);
/* single-GPU convenience: n_iters x (pass, local_reduce, update) enqueued back to back        */
int slv_sk_iterate(const double* P, int64_t N, int K, double* beta, const double* r, double tol,
                   int max_iter, int n_iters, void* ws, int grid, slv_stream_t stream);
/* host_out: 4 doubles {counter, done, err, reserved} in (pinned) host memory                 */
int slv_sk_status(void* ws, int K, int grid, double* host_out, slv_stream_t stream);
int slv_sk_labels(const double* P, int64_t N_local, int K, const double* beta, void* ws, int grid,
                  int64_t* labels /* N_local */, double* logsum_out /* 1 double, device */,
                  slv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SELAVI_HIP_H_ */

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
int32_t slv_stale_hip_errors(void);    /* HIP errors found pending at entry (left by other work; reported on stderr, cleared) */
int slv_device_info(int* cu_count, int* wave_size, char* arch_name, int arch_name_len);

/* ---------------------------------------------------------------- communicator (RCCL over xGMI) -
 * Replaces the NCCL process group of utils.py:133-146 for the hot path's exchanges.  librccl is resolved at run time
 * (dlopen; slv_comm_load(path) names a copy, NULL/"" = the one already in the process, then the system's).  One
 * communicator per process; slv_comm_init is COLLECTIVE over the ranks sharing the 128-byte id of
 * slv_comm_unique_id (rank 0 makes it, the host side distributes it).  Every collective is enqueued on `stream`
 * in order with the kernels around it; buffers are device pointers, in place. */
typedef void* slv_comm_t;
int slv_comm_load(const char* librccl_path /* nullable */);
const char* slv_comm_library(void);               /* what was loaded ("" before the first use)           */
int slv_comm_unique_id(void* id_out_128 /* host, 128 bytes */);
int slv_comm_init(slv_comm_t* comm_out /* host */, const void* unique_id_128 /* host */, int rank, int world);
int slv_comm_destroy(slv_comm_t comm);
int slv_comm_abort(slv_comm_t comm);               /* ncclCommAbort: tear down a communicator whose collective hangs (watchdog) */
int slv_comm_async_error(slv_comm_t comm);         /* ncclCommGetAsyncError: 0 = healthy / in progress, < 0 = failed            */
int32_t slv_comm_count(slv_comm_t comm);          /* ncclCommCount: the ranks RCCL itself reports for this communicator (-1: not available) */
int32_t slv_comm_rank(slv_comm_t comm);
int32_t slv_comm_world(slv_comm_t comm);
int slv_comm_allreduce_f64(slv_comm_t comm, double* buf, int64_t n, slv_stream_t stream);              /* sum */
int slv_comm_allreduce_f32(slv_comm_t comm, float* buf, int64_t n, int average, slv_stream_t stream);  /* sum / mean */
int slv_comm_allreduce_i64(slv_comm_t comm, int64_t* buf, int64_t n, slv_stream_t stream);             /* sum */
int slv_comm_allgather(slv_comm_t comm, const void* send, void* recv /* world * bytes_per_rank */,
                       int64_t bytes_per_rank, slv_stream_t stream);
int slv_comm_broadcast(slv_comm_t comm, void* buf, int64_t bytes, int root, slv_stream_t stream);

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
/* slv_sk_pass followed by slv_sk_local_reduce in one call (the sharded multi-GPU loop is host-enqueue bound) */
int slv_sk_pass_reduce(const double* P, int64_t N_local, int64_t N_global, int K, double* beta, void* ws,
                       int grid, slv_stream_t stream);
int slv_sk_update(const double* r /* K, normalised */, int K, double tol, int max_iter,
                  int first /* 1: right after slv_sk_begin (no counter++) */, void* ws, int grid,
                  slv_stream_t stream);
/* single-GPU convenience: n_iters x (pass, local_reduce, update) enqueued back to back        */
int slv_sk_iterate(const double* P, int64_t N, int K, double* beta, const double* r, double tol,
                   int max_iter, int n_iters, void* ws, int grid, slv_stream_t stream);
/* multi-GPU: n_iters x (pass, local_reduce, all-reduce of the K+1 doubles over `comm`, update), one host call */
int slv_sk_iterate_sharded(slv_comm_t comm, const double* P, int64_t N_local, int64_t N_global, int K, double* beta,
                           const double* r, double tol, int max_iter, int n_iters, void* ws, int grid,
                           slv_stream_t stream);
/* host_out: 4 doubles {counter, done, err, reserved} in (pinned) host memory                 */
int slv_sk_status(void* ws, int K, int grid, double* host_out, slv_stream_t stream);
/* match_order (sk_utils.py:424-467): out[i][j] = sum_n |e1[n][i] - e2[n][j]|; partial = nsplit*K*K doubles */
int slv_sk_l1_cost_matrix(const double* e1, const double* e2, int64_t N, int K, double* partial,
                          int nsplit, double* out /* K*K */, slv_stream_t stream);
int slv_sk_labels(const double* P, int64_t N_local, int K, const double* beta, void* ws, int grid,
                  int64_t* labels /* N_local */, double* logsum_out /* 1 double, device */,
                  slv_stream_t stream);

/* ---------------------------------------------------------------- convolutions ---------------
 * Replace cuDNN Conv3d/Conv2d forward / backward-data / backward-weight of the torchvision nets
 * built by model.py:95 (r2plus1d_18) and model.py:114 (ResNet-9) and driven by main.py:284,301.
 * fp32 in, fp32 accumulate on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32).
 *
 * geom: HOST pointer to 18 int32 = {Bn, Cin, Ti, Hi, Wi, Cout, To, Ho, Wo, kt, kh, kw, st, sh, sw,
 * pt, ph, pw}; strides 1 or 2; 2-D convs are Ti = kt = 1.  Activations N,C,T,H,W; weights
 * [Cout][Cin][kt][kh][kw]; all convs bias-free.  `tab` is a DEVICE copy of slv_conv_table().
 *
 * Fused BatchNorm (every conv of the model is followed by one):
 *   in_scale_shift [2][Cin]  : the input is read as relu?(x*scale[c] + shift[c]) (zero padding applied
 *                              AFTER the affine), i.e. the producer's BN+ReLU is applied on load;
 *   stat_sum/stat_sq [Cout][slv_conv_fwd_nblk()] : per-channel partial sum / sum of squares of y;
 * The backward GEMMs take the gradient w.r.t. the RAW conv output (dXout), which slv_bn_bwd_apply
 * materialises once per layer from the folded BN-backward coefficients of slv_bn_bwd_finalize.
 */
/* gather tables (host side, upload once per layer): dgrad == 0 -> forward / weight-gradient table,
 * dgrad == 1 -> one table block per stride-parity class of the backward-data conv.  _len = int32 words. */
/* Arithmetic of the fp32 convolutions (all but the stems' channel-major launches): 1 (default; SELAVI_CONV_X3) = every fp32
 * operand is cut exactly into three bf16 pieces and a product is six partial products on v_mfma_f32_16x16x32_bf16 with fp32
 * accumulation (csrc/igemm3.hpp: the dropped terms are <= 3 * 2^-24 |a b|, one fp32 rounding; 2.67x the native rate);
 * 0 = the native fp32-input MFMA (csrc/igemm.hpp).  Weight images (slv_conv_w_transform), their sizes and the launch
 * configurations depend on the setting: make them again after switching. */
int slv_conv_set_arithmetic(int split_bf16x3);
int32_t slv_conv_get_arithmetic(void);
int32_t slv_conv_table_len(const int32_t* geom, int dgrad);
int slv_conv_table(const int32_t* geom, int dgrad, int32_t* tab_host_out /* host, table_len words */);
/* Launch configuration `cfg`: 0 = built-in heuristic, otherwise one of the values enumerated by
 * slv_conv_configs (tile rows/16 | tile cols/64 << 8 | MFMA shape << 12 (0: 16x16x4, 1: 32x32x2) | K-slices << 16).  The host may time the
 * candidates once per layer shape -- what the reference gets from cudnn.benchmark = True (main.py:187).
 * op: 0 forward, 1 backward-data, 2 backward-weight.  Returns the number of candidates written. */
int32_t slv_conv_configs(const int32_t* geom, int op, int32_t* cfg_out, int32_t max_out);
int32_t slv_conv_fwd_nblk(const int32_t* geom, int32_t cfg);
/* split-K scratch (late layers: few columns, deep K): 0 when the layer runs unsplit */
size_t slv_conv_fwd_ws_bytes(const int32_t* geom, int32_t cfg);
/* Weight layouts.  Layers whose channel count pads to a multiple of 16 with <= 10 % waste run with a
 * TAP-MAJOR K order (k = ((c/16)*ntaps + tap)*16 + c%16: every 16-deep K chunk has one tap, so padding
 * validity and address math are per chunk, not per element; a 16-channel group visits its taps back to
 * back so the shifted re-reads stay in L1/L2); they read re-laid-out copies of the weights made
 * once per step by slv_conv_w_transform (one read of w):
 *   wf  forward weights   [Cout][K order above over (ci, tap)]   slv_conv_wf_elems() floats, 0 = the forward
 *                                                               conv reads w itself (channel-major)
 *   wt  backward-data weights, per stride-parity class c:       slv_conv_wt_elems() floats
 *       [ci][K order above over (co, tap_j)] (tap-major) or [ci][co*ntaps_c + j] (channel-major), classes concatenated */
size_t slv_conv_wf_elems(const int32_t* geom);
size_t slv_conv_wt_elems(const int32_t* geom);
int slv_conv_w_transform(const int32_t* geom, const float* w, float* wf /* nullable */, float* wt /* nullable */,
                         slv_stream_t stream);
/* The split-operand weight images of MANY layers in one launch.  slv_conv_w_jobs writes the job descriptors of one layer
 * (host memory: slv_conv_w_job_words() int32 each -- an image is cut into jobs of 16 384 slots, so a layer-4 conv has ~80 --; w / wf /
 * wt as slv_conv_w_transform: they must stay where they are for the table's lifetime) and returns their number (-1: max_jobs
 * too small or a bad argument); 0 = this layer does not use split-operand images (stems, native arithmetic): keep
 * slv_conv_w_transform for it.  blocks_per_job = 0: the library's choice.  The caller concatenates the jobs of its layers into ONE device buffer (once)
 * and calls slv_conv_w_transform_jobs on it every step. */
int32_t slv_conv_w_job_words(void);
int32_t slv_conv_w_jobs(const int32_t* geom, const float* w, float* wf, float* wt, int32_t* out_jobs, int32_t max_jobs);
int slv_conv_w_transform_jobs(const int32_t* jobs_dev, int32_t njobs, int32_t blocks_per_job, slv_stream_t stream);
/* Eval-mode forward with BatchNorm FOLDED into the weights (the SK feature pass, sk_utils.py:137-254; selavi_amd/infer32.py):
 * y = relu?(conv(x, w') + bias[co] (+ res)) in one launch of the split-operand kernel -- wf = slv_conv_w_transform's image of
 * w' = w * scale[co] (made once per pass, the weights do not change during it), bias = the BatchNorm shift, res = the block's
 * shortcut (y's shape) or NULL.  pieces = 3: the exact three-piece split of the training path (6 partial products); 2 (opt-in):
 * two pieces, 3 partial products -- 16-17 significand bits per product at half the matrix-core work.  Unsplit K (cfg's slice
 * count is ignored), no statistics, x read as stored.  Layers without a split-operand image (slv_conv_fwd_eval_ok == 0: the
 * 3 / 1-channel stems, the native arithmetic) keep slv_conv_fwd + slv_bn_act. */
int32_t slv_conv_fwd_eval_ok(const int32_t* geom);
int slv_conv_fwd_eval(const int32_t* geom, const float* x, const float* wf, const int32_t* tab, const float* bias /* [Cout], nullable */,
                      const float* res /* nullable */, int relu, int pieces, float* y, int32_t cfg, slv_stream_t stream);
int slv_conv_fwd(const int32_t* geom, const float* x, const float* w /* nullable if wf is used */,
                 const float* wf /* nullable if slv_conv_wf_elems() == 0 */, const int32_t* tab,
                 const float* in_scale_shift /* nullable */, int in_relu, float* y,
                 float* stat_sum /* nullable */, float* stat_sq, void* ws /* nullable if 0 bytes */,
                 size_t ws_bytes, int32_t cfg, slv_stream_t stream);
size_t slv_conv_dgrad_ws_bytes(const int32_t* geom, int32_t cfg);
/* Optional fused BatchNorm-backward reduction (the BN of the layer that produced this conv's input,
 * consumed through ReLU): with g = dx as stored, x = bnr_x (raw output of that layer, shape of dx) and
 * g' = g * (x*scale + shift > 0), every channel gets slv_conv_dgrad_bnr_slots() partial pairs
 * bnr_part[c][slot] = {sum g', sum g' * (x - mean) * invstd} -- the input of slv_bn_bwd_sums[_finalize];
 * replaces a separate slv_bn_bwd_reduce pass over g and x. */
int32_t slv_conv_dgrad_bnr_slots(const int32_t* geom, int32_t cfg);
int slv_conv_dgrad(const int32_t* geom, const float* dy, const float* wt, const int32_t* tab,
                   float* dx, const float* addend /* nullable, may alias dx */,
                   const float* bnr_x /* nullable: no fused reduction */,
                   const float* bnr_scale_shift /* [2][Cin] */, const float* bnr_mean_invstd /* [2][Cin] */,
                   float* bnr_part /* [Cin][slots][2] */, void* ws /* nullable if 0 bytes */,
                   size_t ws_bytes, int32_t cfg, slv_stream_t stream);
size_t slv_conv_wgrad_ws_bytes(const int32_t* geom, int32_t cfg);
/* dw = sum_p dy[co,p] * act(x_in)[ci, p*stride+tap-pad]; deterministic split-K via `ws`.  tab = fwd table */
int slv_conv_wgrad(const int32_t* geom, const float* dy, const float* x_in,
                   const float* in_scale_shift /* nullable */, int in_relu, const int32_t* tab,
                   float* dw, void* ws, size_t ws_bytes, int32_t cfg, slv_stream_t stream);
/* C[m][n] = sum_k A[m][k] * B[n][k] (+ bias[n]);  A [M][K], B [N][K] row-major (nn.Linear layout) */
int slv_gemm_nt(const float* A, const float* B, const float* bias /* nullable */, float* C, int M,
                int N, int K, int ldc, slv_stream_t stream);

/* ---------------------------------------------------------------- BatchNorm / pools / SGD -----
 * torch BatchNorm3d/2d train+eval, ReLU, residual add, AdaptiveAvgPool(1), MaxPool2d(3,2,1) of the
 * same nets; torch.optim.SGD(momentum, weight_decay) of main.py:132-137.  Tensors [Bn][C][P].
 * SyncBN (main.py:117-118): all-reduce(sum) the double `sums` buffers between the two calls.     */
int slv_bn_partials_to_sums(const float* psum, const float* psq, int nblk, int C,
                            double* sums /* [2][C] */, slv_stream_t stream);
int slv_bn_finalize(const double* sums, double count, const float* gamma, const float* beta,
                    float* running_mean /* nullable: no update */, float* running_var, float momentum,
                    float eps, float* mean_invstd /* [2][C] */, float* scale_shift /* [2][C] */, int C,
                    slv_stream_t stream);
/* single-process BatchNorm (no SyncBN exchange): the two calls above in one launch */
int slv_bn_stats_finalize(const float* psum, const float* psq, int nblk, double count, const float* gamma,
                          const float* beta, float* running_mean /* nullable */, float* running_var,
                          float momentum, float eps, float* mean_invstd, float* scale_shift, int C,
                          slv_stream_t stream);
/* SyncBN in one call on one stream: partials -> fp64 sums (sums_scratch, 2C doubles) -> all-reduce over `comm` ->
 * finalize with count_local * world.  The backward twin folds slv_bn_bwd_sums + all-reduce + slv_bn_bwd_finalize.
 * ASSUMES EQUAL PER-RANK COUNTS (count_global = count_local * world): true for the training step -- the reference's
 * loader drops the ragged last batch (main.py:94-101 drop_last=True) and every rank feeds the same clip shape; a caller
 * with ragged per-rank batches must use slv_bn_partials_to_sums + slv_comm_allreduce_f64 (count included) + slv_bn_finalize. */
int slv_bn_sync_finalize(slv_comm_t comm, const float* psum, const float* psq, int nblk, double count_local,
                         const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                         float eps, float* mean_invstd, float* scale_shift, int C, double* sums_scratch,
                         slv_stream_t stream);
int slv_bn_bwd_sync_finalize(slv_comm_t comm, const float* partial, int nsplit, double count_local, const float* gamma,
                             const float* mean_invstd, const float* scale_shift, float* bwd5, float* dgamma,
                             float* dbeta, int accumulate, int C, double* sums_scratch, slv_stream_t stream);
int slv_bn_eval_params(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* mean_invstd /* nullable */,
                       float* scale_shift, int C, slv_stream_t stream);
/* out = relu?(x*s+h + (res ? (res_ss ? res*rs+rh : res) : 0)) : block tail / activation materialise */
int slv_bn_act(const float* x, const float* scale_shift, const float* res /* nullable */,
               const float* res_scale_shift /* nullable */, int relu, float* out, int Bn, int C,
               int64_t P, slv_stream_t stream);
int32_t slv_bn_bwd_nsplit(int Bn, int C, int64_t P);
/* partial[c][nsplit][2] = {sum g', sum g' xhat}; g' = g masked by (scale_shift_mask: own BN output > 0)
 * or (v_mask > 0, masked gradient also written to g_out) or unmasked; optional second BN (x2) sharing g' */
int slv_bn_bwd_reduce(const float* g, const float* x, const float* mean_invstd,
                      const float* scale_shift_mask /* nullable */, const float* v_mask /* nullable */,
                      const float* x2 /* nullable */, const float* mean_invstd2, float* g_out,
                      float* partial, float* partial2, int Bn, int C, int64_t P, int nsplit,
                      slv_stream_t stream);
int slv_bn_bwd_sums(const float* partial, int nsplit, int C, double* sums /* [2][C] */,
                    slv_stream_t stream);
int slv_bn_bwd_finalize(const double* sums, double count, const float* gamma, const float* mean_invstd,
                        const float* scale_shift /* nullable: no relu mask */, float* bwd5 /* [5][C] */,
                        float* dgamma /* nullable */, float* dbeta, int accumulate, int C,
                        slv_stream_t stream);
/* single-process: slv_bn_bwd_sums + slv_bn_bwd_finalize in one launch */
int slv_bn_bwd_sums_finalize(const float* partial, int nsplit, double count, const float* gamma,
                             const float* mean_invstd, const float* scale_shift /* nullable */,
                             float* bwd5, float* dgamma /* nullable */, float* dbeta, int accumulate, int C,
                             slv_stream_t stream);
/* out = A1*mask*g + A2 + A3*x : the gradient w.r.t. the raw conv output, materialised (may alias g) */
int slv_bn_bwd_apply(const float* g, const float* x, const float* bwd5, int relu, float* out, int Bn,
                     int C, int64_t P, slv_stream_t stream);
int slv_avgpool_fwd(const float* v, float* out, int rows, int P, slv_stream_t stream);
int slv_avgpool_bwd(const float* dout, float* dv, int rows, int P, slv_stream_t stream);
int slv_bnrelu_maxpool_fwd(const float* x, const float* scale_shift, float* out, uint8_t* idx, int Bn,
                           int C, int H, int W, slv_stream_t stream);
int slv_maxpool_bwd(const float* dout, const uint8_t* idx, float* dy, int Bn, int C, int H, int W,
                    slv_stream_t stream);
/* params/grads/bufs/sizes: HOST arrays of n_tensors device pointers / element counts */
int slv_sgd_step(const void* const* params, const void* const* grads, const void* const* bufs,
                 const int64_t* sizes, int n_tensors, float lr, float momentum, float weight_decay,
                 int first_step, slv_stream_t stream);
/* out[i] = sum_s src[s][i], s ascending (weight gradients of batch slices, see ops.ConvPlan chunks) */
int slv_sum_slices(const float* src, float* out, int slices, int64_t n, slv_stream_t stream);
int slv_fill_f32(float* p, float value, int64_t n, slv_stream_t stream);

/* ---------------------------------------------------------------- grouped heads + loss ---------
 * All G = 2*headcount heads per launch (model.py:62-90,233-252; utils.py:377-387; main.py:291-293).
 * Activations are stacked [G][B][dim]; parameter pointers come as HOST arrays of G device pointers
 * (each head keeps its own nn.Parameter so state_dict keys match the reference).
 * x: shared_x ? [2][B][IN] (group g reads modality g / hc) : [G][B][IN].                          */
int slv_heads_linear_fwd(const float* x, int shared_x, int hc, const float* mask /* nullable [G][B][IN] */,
                         float mask_scale, const void* const* W, const void* const* bias /* nullable */,
                         float* out, int G, int B, int IN, int OUT, slv_stream_t stream);
int slv_heads_bn_stats(const float* h, double* sums /* [G][2][C] */, int G, int B, int C,
                       slv_stream_t stream);
int slv_heads_bn_apply(const float* h, const double* sums, double count, const void* const* gamma,
                       const void* const* beta, const void* const* running_mean,
                       const void* const* running_var, const float* mask2 /* nullable */,
                       float mask_scale, float momentum, float eps, int training, float* a,
                       float* mean_invstd /* [G][2][C] */, int G, int B, int C, slv_stream_t stream);
int slv_heads_ce(const float* logits, const int64_t* labels /* [B][label_stride], column g % hc */,
                 int label_stride, int hc, float* loss_rows /* [G*B] */, float* dlogits /* nullable */,
                 float grad_scale, int G, int B, int K, slv_stream_t stream);
/* *total = scale * sum(loss_rows) in one workgroup, fixed order (the mean of utils.py:377-387 without a host-side reduce) */
int slv_heads_ce_total(const float* loss_rows, int64_t n, float scale, float* total, slv_stream_t stream);
/* Dropout(0.3) keep-masks of the heads (model.py:79,85) from Philox4x32-10: element e of m1 | m2 = word e % 4 of the block
 * with counter (e / 4, offset) under key `seed`; 1.0 iff word >= p * 2^32.  Pure function of (seed, offset, e). */
int slv_dropout_masks(uint64_t seed, uint64_t offset, float p, float* m1, int64_t n1, float* m2 /* nullable */,
                      int64_t n2, slv_stream_t stream);
/* The same draw with (seed, offset) = state[0], state[1] read on the DEVICE, followed by state[1] += 1: the form a HIP
 * graph replays with fresh masks (host scalars are frozen into a captured launch; train.GraphedStep). */
int slv_dropout_masks_dev(uint64_t* state /* device, 2 words */, float p, float* m1, int64_t n1, float* m2 /* nullable */,
                          int64_t n2, slv_stream_t stream);
int slv_heads_linear_bwd_w(const float* dout, const float* x, int shared_x, int hc, const float* mask,
                           float mask_scale, float* dW /* [G][OUT][IN] */, float* dbias /* nullable */,
                           int G, int B, int IN, int OUT, slv_stream_t stream);
int slv_heads_linear_bwd_x(const float* dout, const void* const* W, const float* mask, float mask_scale,
                           float* dx /* [G][B][IN] */, int G, int B, int IN, int OUT, slv_stream_t stream);
int slv_heads_bn_bwd_stats(const float* da, const float* h, const float* mean_invstd,
                           const void* const* gamma, const void* const* beta, const float* mask2,
                           float mask_scale, double* sums /* [G][2][C] */, int G, int B, int C,
                           slv_stream_t stream);
int slv_heads_bn_bwd_apply(const float* da, const float* h, const float* mean_invstd,
                           const void* const* gamma, const void* const* beta, const float* mask2,
                           float mask_scale, const double* sums, double count, float* dh, float* dgamma,
                           float* dbeta, int G, int B, int C, slv_stream_t stream);
int slv_heads_sum_groups(const float* src /* [2*hc][n] */, float* out /* [2][n] */, int hc, int64_t n,
                         slv_stream_t stream);
int slv_rowwise_affine(const float* x, const float* scale_shift, int relu, float* y, int64_t rows, int C,
                       slv_stream_t stream);

/* ---- evaluation dumps (clustering_metrics.py:95-175) -----------------------------------------------------------
 * slv_av_argmax: labels[i] = argmax_k softmax64(lv[i])_k * softmax64(la[i])_k (:121-126 / :140-144), first index
 *   on ties -- the same arithmetic as slv_sk_prepare without materialising the N x K matrix.
 * slv_contingency: counts[a][b] = #{i : pred[i] == a and target[i] == b} (the K x K vote table _hungarian_match
 *   :41-56 builds with K*K masked sums); *bad is set to 1 if any index is out of range. */
int slv_av_argmax(const float* lv, const float* la, int64_t N, int K, int64_t* labels, slv_stream_t stream);
int slv_contingency(const int64_t* pred, const int64_t* target, int64_t N, int K1, int K2, int64_t* counts,
                    int32_t* bad, slv_stream_t stream);

/* ---- 16-bit MFMA path, first kernel (BASELINE configs[4]; main.py:151 --use_fp16 trains the convs in half
 * precision through apex O1).  bf16 CHANNELS-LAST activations [N][T][H][W][Cp] (Cp % 32 == 0, channels >= C are zero),
 * fp32 accumulation on v_mfma_f32_16x16x32_bf16, fused epilogue y = relu?(acc * scale + shift + residual) -> bf16.
 * slv_conv_cl16_fwd replaces nn.Conv3d/Conv2d forward (+ the eval-mode BatchNorm / ReLU / residual that follow it in
 *   torchvision's blocks) for one layer.  geom: 20 int32 = {N, Ti, Hi, Wi, Cin_p, Cout, Cout_p, To, Ho, Wo, kt, kh, kw,
 *   st, sh, sw, pt, ph, pw, Mrows}; mt in {4, 8, 9}: 16-row tiles per block, Mrows % (16*mt) == 0, Mrows >= Cout;
 *   w_layout_bf16: [kt*kh*kw][Cin_p/32][Mrows][32] bf16, zero padded (selavi_amd/ops16.py builds it);
 *   scale_shift: fp32 [2][Cout] or null; res_bf16: [P_out][Cout_p] or null.
 * slv_to_cl16: fp32 N,C,T,H,W (S = T*H*W) -> bf16 N,T,H,W,Cp. */
int slv_conv_cl16_fwd(const int32_t* geom, int mt, const void* x_bf16, const void* w_layout_bf16, void* y_bf16,
                      const float* scale_shift, const void* res_bf16, int relu, slv_stream_t stream);
int slv_to_cl16(const float* x, void* y_bf16, int64_t N, int C, int Cp, int64_t S, slv_stream_t stream);
/* stem input: fp32 N,C,T,H,W (TH = T*H) -> bf16 [N][T][H][Wo][32], channel dw*C + c = x[.., w = wo*sw - pw + dw] (zero
 * outside): a (1,kh,kw) conv over C <= 4 channels becomes a (1,kh,1) conv over 32 channels (kw * C <= 32) */
int slv_to_cl16_wpatch(const float* x, void* y_bf16, int64_t N, int C, int64_t TH, int W, int kw, int sw, int pw,
                       slv_stream_t stream);
/* The stem convs DIRECTLY from the fp32 clip / spectrogram (csrc/conv_cl16_stem.hip; round 5): Conv3d(Cin <= 3 -> 32 < Cout <= 64,
 * (1,7,7), stride (1,2,2), padding (0,3,3)) -- torchvision's R(2+1)D / ResNet stems, /root/reference/model.py:93-114 -- without
 * the W-patch tensor of slv_to_cl16_wpatch.  x: fp32 [N][Cin][T][H][W] (W <= 112); w: the fp32 master weights
 * [Cout][Cin][1][7][7]; y: bf16 [N][T][Ho][Wo][64]; stat_sum / stat_sq (both or neither): [Cout][slv_cl16_stem_nblk()] partial
 * sums / sums of squares of the rounded outputs (BatchNorm statistics, as slv_cl16_conv's).  slv_cl16_stem_ok: 1 when the
 * direct kernels take the geometry (SELAVI_CL16_STEM=0 switches them off: the W-patch path of rounds 1-4). */
int32_t slv_cl16_stem_ok(int N, int Cin, int T, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw);
int32_t slv_cl16_stem_nblk(int N, int Cin, int T, int H, int W, int Cout);
int slv_cl16_stem_fwd(const float* x, const float* w, void* y_bf16, float* stat_sum, float* stat_sq, int N, int Cin, int T,
                      int H, int W, int Cout, slv_stream_t stream);
/* ... and its weight gradient: dy bf16 [N][T][Ho][Wo][64] (the gradient w.r.t. the conv's raw output), dw fp32
 * [Cout][Cin][1][7][7] (the reference layout), ws: slv_cl16_stem_wgrad_ws_bytes() of scratch (per-workgroup partials,
 * summed in a fixed order: deterministic).  y_bf16 / bwd5 (nullable, together) / relu: dy is the gradient w.r.t. the ACTIVATED
 * output of the conv's BatchNorm and the kernel applies that BatchNorm's backward on load -- what slv_cl16_bn_bwd_apply(dy,
 * y, bwd5, relu) would have stored, bit for bit, without the pass (the stem's first conv has no backward-data launch). */
size_t slv_cl16_stem_wgrad_ws_bytes(int N, int Cin, int T, int H, int W, int Cout);
int slv_cl16_stem_wgrad(const float* x, const void* dy_bf16, float* dw, float* ws, size_t ws_bytes, int N, int Cin, int T, int H,
                        int W, int Cout, const void* y_bf16, const float* bwd5, int relu, slv_stream_t stream);
/* MaxPool2d(3, 2, 1) on [N][H][W][Cp] bf16; AdaptiveAvgPool(1)+flatten: [N][S][Cp] bf16 -> fp32 [N][C] */
int slv_maxpool_cl16(const void* x_bf16, void* y_bf16, int64_t N, int H, int W, int Cp, slv_stream_t stream);
int slv_avgpool_cl16(const void* x_bf16, float* y, int64_t N, int64_t S, int C, int Cp, slv_stream_t stream);

/* ---- 16-bit MFMA path, training (BASELINE configs[4]; main.py:151-153 apex.amp O1, :296-299 scaled-loss backward --
 * bf16 needs no loss scaling).  fp32 master weights / BatchNorm parameters / statistics, bf16 channels-last activations
 * and activation gradients, fp32 accumulation, fp32 weight gradients in the reference layout.
 *
 * slv_cl16_conv: the general launch of the bf16 implicit-GEMM kernel.  clconv: slv_cl16_conv_words() int32 =
 *   {N, Ti, Hi, Wi, Cin_p, Cin, Lt, Lh, Lw, bmt, bmh, bmw, bot, boh, bow, To, Ho, Wo, Cout, Cout_p, omt, omh, omw, oot,
 *    ooh, oow, Mrows, ntaps, tap[64], flags}: the block enumerates the lattice Lt x Lh x Lw per clip; activation rows are read
 *   at lattice*bm + bo + tap offset, the output row is lattice*om + oo; tap = (dt+8) | (dh+8) << 4 | (dw+8) << 8 |
 *   weight slab << 12; flags bit 0 = a forward launch (kernel choice only).  Forward conv: lattice = output, bm = stride, bo = -pad; backward data: one launch per
 *   stride-parity class of the input positions (selavi_amd/ops16.py builds the tables).
 *   in_scale_shift [2][Cin] (nullable): rows are read as relu(x*s + h), zero padding after the affine (train-mode
 *     BatchNorm + ReLU of the producing layer applied on load);
 *   stat_sum/stat_sq [Cout][slv_cl16_conv_nblk()] (nullable, together): per-channel partial sum / sum of squares of the
 *     bf16-rounded output -- the input of slv_bn_stats_finalize / slv_bn_partials_to_sums;
 *   otherwise as slv_conv_cl16_fwd (scale_shift / res / relu epilogue; res doubles as the backward-data addend);
 *   bnr_* (backward data inside a conv chain, cf. slv_conv_dgrad): with g = the gradient as stored, x = bnr_x (raw
 *     output of the layer that produced this conv's input) and g' = g * (x*scale + shift > 0), every channel gets
 *     slv_cl16_conv_nblk() partial pairs bnr_part[c][bnr_slot0 + tile] = {sum g', sum g' * (x - mean) * invstd} -- the
 *     input of slv_bn_bwd_sums[_finalize]; the parity classes of a strided layer use consecutive slot ranges.
 * slv_cl16_w_transform: fp32 [Cout][Cin][taps] -> forward layout [taps][Cin_p/32][mrows_fwd][32] and backward-data
 *   layout [taps][Cout_p/32][mrows_dgrad][32] (either nullable); patch_kw > 0: forward layout of the stem's W-patch conv.
 * slv_cl16_wgrad: dw[Cout][Cin][taps] (fp32, reference layout) = sum_pos dy[pos][co] * act(x)[pos*stride+tap-pad][ci];
 *   clw: slv_cl16_wgrad_words() int32 = {N, Ti, Hi, Wi, Cin_p, Cin, To, Ho, Wo, Cout_p, st, sh, sw, pt, ph, pw, kt, kh, kw,
 *   Ncols, mtiles, ntiles, kslices, kper}; tile = (32*wm) x (32*wn), wm, wn in 2..5; deterministic split-K through `ws`.
 *   Stride-1 (1,3,3) layers run on the rolling-patch kernel (csrc/wgrad_cl16_s3.hip), stride-1 (3,1,1) layers on the
 *   column-order kernel (csrc/wgrad_cl16_t.hip); each has its own tiling and K slices (slv_cl16_wgrad_ws_bytes covers all); the result does not depend on which kernel ran beyond fp32 summation order.
 * slv_cl16_bn_*: channels-last bf16 versions of slv_bn_act / slv_bn_bwd_reduce / slv_bn_bwd_apply on [P][Cp]. */
int32_t slv_cl16_conv_words(void);
int32_t slv_cl16_conv_nblk(const int32_t* clconv);
/* Which launches of slv_cl16_conv the 8-wave kernel of the wide layers takes (csrc/conv_cl16_g8.hip: 256 positions x 128 /
 * 256 / 288 channels per workgroup, weight layouts whose rows come in such blocks): 0 = none, 1 = the launches that fill
 * the chip (default; SELAVI_CL16_G8 sets the initial value), 2 = every launch it can express (tests).  mode < 0 only reads.
 * Returns the previous mode.  slv_cl16_conv_nblk() follows the mode: set it before plans are made. */
int32_t slv_cl16_g8_mode(int32_t mode);
int slv_cl16_conv(const int32_t* clconv, int mt, const void* x_bf16, const void* w_layout_bf16, void* y_bf16,
                  const float* in_scale_shift, const float* scale_shift, const void* res_bf16, int relu,
                  float* stat_sum, float* stat_sq, const void* bnr_x_bf16 /* nullable: no fused reduction */,
                  const float* bnr_scale_shift /* [2][Cout] */, const float* bnr_mean_invstd /* [2][Cout] */,
                  float* bnr_part /* [Cout][bnr_nslots][2] */, int bnr_slot0, int bnr_nslots, slv_stream_t stream);
/* Backward data of a stride-1 (3,1,1) conv with the BatchNorm-backward APPLY of the layer it feeds folded into its epilogue
 * (main.py:296-299): out = A1 * mask * g + A2 + A3 * x with g = this conv's gradient as it would have been stored (bf16),
 * x = src_x (raw output of the layer that produced this conv's input, same positions), mask = [x s + h > 0],
 * bwd5 = {s, h, A1, A2, A3}[Cout] as slv_cl16_bn_bwd_apply takes them -- bit for bit what that pass makes of the stored g,
 * without the pass (possible when the coefficients are known beforehand: slv_cl16_wgrad_bnr).  clconv: a backward-data
 * launch description; slv_cl16_conv_dgrad_bn_apply_ok says whether this path takes it (the register-resident column
 * kernel of the 64 -> 144 layer, csrc/conv_cl16_tr.hip). */
int32_t slv_cl16_conv_dgrad_bn_apply_ok(const int32_t* clconv);
int slv_cl16_conv_dgrad_bn_apply(const int32_t* clconv, const void* dy_bf16, const void* w_layout_bf16, void* out_bf16,
                                 const void* src_x_bf16, const float* bwd5, slv_stream_t stream);
int slv_cl16_w_transform(const float* w, void* wf_bf16, void* wt_bf16, int Cout, int Cin, int taps, int Cin_p,
                         int Cout_p, int mrows_fwd, int mrows_dgrad, int patch_kw, slv_stream_t stream);
/* slv_cl16_w_transform for MANY layers in one launch: jobs_dev = njobs x 18 int32 in device memory, one job =
 * {w, wf, wt (three 64-bit pointers, wf / wt nullable), Cout, Cin, taps, Cin_p, Cout_p, mrows_fwd, mrows_dgrad, patch_kw,
 * nf, nt, first, count} with nf / nt = the element counts of the two layouts (0 for a null one) -- the arguments of
 * slv_cl16_w_transform -- and [first, first + count) the elements of the nf + nt this job makes (equal jobs: the caller cuts
 * a large layer into several). */
int slv_cl16_w_transform_jobs(const int32_t* jobs_dev, int32_t njobs, int32_t blocks_per_job, slv_stream_t stream);
int32_t slv_cl16_wgrad_words(void);
size_t slv_cl16_wgrad_ws_bytes(const int32_t* clw, int wm, int wn);
int slv_cl16_wgrad(const int32_t* clw, int wm, int wn, const void* dy_bf16, const void* x_bf16,
                   const float* in_scale_shift, float* dw, int Cout, int patch_kw, void* ws, size_t ws_bytes,
                   slv_stream_t stream);
/* Weight gradient of a stride-1 (3,1,1) conv that ALSO returns the BatchNorm-backward sums of the layer the conv reads
 * (x = that layer's raw output, in_scale_shift / in_mean_invstd its BatchNorm; main.py:296-299): dw as slv_cl16_wgrad with
 * the BN + ReLU prologue, bn_part[C][2] = { sum g', sum g' xhat } with g' = the masked gradient w.r.t. relu(bn(x)) -- the
 * input of slv_bn_bwd_sums_finalize, without the slv_cl16_bn_bwd_reduce pass over the gradient and x.  w: the conv's fp32
 * master weights [Cout][Cin][3] (rounded to bf16 inside, as the backward-data conv uses them).
 * slv_cl16_wgrad_bnr_ws_bytes returns 0 for a geometry this path does not take. */
size_t slv_cl16_wgrad_bnr_ws_bytes(const int32_t* clw);
int slv_cl16_wgrad_bnr(const int32_t* clw, const void* dy_bf16, const void* x_bf16, const float* in_scale_shift,
                       const float* in_mean_invstd, const float* w, float* dw, float* bn_part, int Cout, void* ws,
                       size_t ws_bytes, slv_stream_t stream);
/* slv_cl16_bn_act: out = relu?(x*s + h + residual) on [P][Cp] bf16.  relu bit 0: the ReLU of the sum; bit 1 (with
 * res_scale_shift): the residual is relu(res*rs + rh) ROUNDED TO bf16 -- the activated output of its BatchNorm exactly as a
 * consumer's load prologue makes it from the raw tensor (the stem's un-materialised block output). */
int slv_cl16_bn_act(const void* x_bf16, const float* scale_shift, const void* res_bf16, const float* res_scale_shift,
                    int relu, void* out_bf16, int64_t P, int C, int Cp, slv_stream_t stream);
int32_t slv_cl16_bn_bwd_nsplit(int64_t P, int Cp);
int slv_cl16_bn_bwd_reduce(const void* g_bf16, const void* x_bf16, const float* mean_invstd, const float* scale_shift_mask,
                           const void* v_mask_bf16, const void* x2_bf16, const float* mean_invstd2, void* g_out_bf16,
                           float* partial, float* partial2, int64_t P, int C, int Cp, int nsplit, slv_stream_t stream);
int slv_cl16_bn_bwd_apply(const void* g_bf16, const void* x_bf16, const float* bwd5, int relu, void* out_bf16, int64_t P,
                          int C, int Cp, slv_stream_t stream);
int slv_cl16_avgpool_bwd(const float* dout, void* dv_bf16, int64_t N, int64_t S, int C, int Cp, slv_stream_t stream);
/* the audio trunk's stem on the 16-bit path (torchvision ResNet conv1-bn1-relu-maxpool: /root/reference/model.py:114-132
 * builds it, main.py:296-299 runs its backward; fp32 counterparts: slv_bnrelu_maxpool_fwd / slv_maxpool_bwd):
 * out = MaxPool2d(3, 2, 1)(relu(x * scale + shift)) on bf16 [N][H][W][Cp], idx [N][Ho][Wo][Cp] = winning tap 0..8 (first
 * maximum, as torch); backward: the gather form over the <= 4 windows of an input pixel, fp32 sum, one rounding */
int slv_cl16_bnrelu_maxpool_fwd(const void* x_bf16, const float* scale_shift, void* out_bf16, uint8_t* idx, int64_t N, int C,
                                int Cp, int H, int W, slv_stream_t stream);
int slv_cl16_maxpool_bwd(const void* dout_bf16, const uint8_t* idx, void* dy_bf16, int64_t N, int Cp, int H, int W,
                         slv_stream_t stream);
/* bf16 [N][S][Cp] -> fp32 N,C,S: the inverse of slv_to_cl16 (inspection, tests) */
int slv_from_cl16(const void* x_bf16, float* y, int64_t N, int C, int Cp, int64_t S, slv_stream_t stream);

/* ---- input pipeline: what the reference's DataLoader workers compute per clip on the CPU -------
 * slv_clip_augment replaces datasets/video_transforms.py:462-510 (clip_augmentation: /255, -mean, /std, THWC->TCHW,
 *   spatial_sampling :420-459 = bilinear short-side resize :35-80 + crop :101-134/:167-210 + flip :137-164, ->CTHW)
 *   for a batch of B clips in one launch.  desc: B x 8 int64 on the device per clip = {byte offset of the clip's
 *   T*H*W*3 uint8 frames inside frames_u8, H, W, resized H, resized W, crop y offset, crop x offset, flip}; the
 *   random draws stay on the host (selavi_amd/datasets/video_transforms.py makes them in the reference's order).
 *   out: [B][3][T][S][S] float32.  mean3/std3: host pointers.
 * slv_logfbank replaces datasets/audio_utils.py:46-72 (python_speech_features.logfbank 0.6 with winfunc = ones,
 *   lowfreq 0, highfreq samplerate/2) for B clips: wav_i16 [B][wav_stride] int16 PCM, start_i64[b] first sample of
 *   the clip's window, volume_f64 nullable per-clip factor (audio_utils.py:42-43), slen samples per window;
 *   twiddle_f64 = cos(2 pi j / nfft), j < nfft, then sin(...) (device, float64); bins_i32 = the nfilt + 2 filterbank
 *   bin edges floor((nfft+1) * mel2hz(linspace) / samplerate) (device).  out_f32: [B][1][nfilt][frames] with
 *   frames = slv_logfbank_frames(slen, frame_len, frame_step); z_normalize applies (x - 1.93) / 17.89 (:71-72). */
int slv_clip_augment(const void* frames_u8, const int64_t* desc, float* out, int B, int T, int S,
                     const float* mean3, const float* std3, slv_stream_t stream);
int32_t slv_logfbank_frames(int slen, int frame_len, int frame_step);   /* sigproc.framesig frame count; -1 on bad sizes */
int slv_logfbank(const void* wav_i16, const int64_t* start_i64, const double* volume_f64, int64_t wav_stride, int B,
                 int slen, int frame_len, int frame_step, int nfft, int nfilt, const double* twiddle_f64,
                 const int32_t* bins_i32, double preemph, int z_normalize, float* out_f32, slv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SELAVI_HIP_H_ */

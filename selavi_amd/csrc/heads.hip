// Grouped MLP heads + softmax cross-entropy: all 2*headcount heads of the model in ONE launch per
// stage (the reference loops over heads in Python, model.py:244-251, ~160 tiny kernels).
// Head g (g < hc: video, g >= hc: audio) is MLPv2 (model.py:62-90):
//   Dropout -> Linear(512,512,no bias) -> BatchNorm1d -> ReLU -> Dropout -> Linear(512,K)+bias
// or a plain Linear(512,K) when use_mlp is off.  Loss: utils.get_loss (utils.py:377-387) and the
// 0.5/0.5 mix of main.py:291-293 folded into the logits gradient scale.
// These layers are < 0.1 % of the step's FLOPs and are weight-bandwidth bound: one wave per output
// column (lanes stride the 512-long reduction), wave butterflies, no LDS tiling needed.
#include "common.hpp"
#include "../../include/selavi_hip.h"

namespace slv {

constexpr int MAXG = 40;
struct PtrTab {
  const float* p[MAXG];
};
struct PtrTabW {
  float* p[MAXG];
};

// out[g][b][n] = sum_k X(g)[b][k] * (mask ? mask[g][b][k]*msc : 1) * W[g][n][k] (+ bias[g][n])
// X(g) = xin + (shared_x ? (g / hc) : g) * B*IN
template <bool MASK>
__global__ __launch_bounds__(256) void heads_linear_fwd_kernel(const float* __restrict__ xin, int shared_x, int hc,
                                                              const float* __restrict__ mask, float msc,
                                                              const PtrTab W, const PtrTab bias, int has_bias,
                                                              float* __restrict__ out, int B, int IN, int OUT) {
  const int g = blockIdx.y;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= OUT) return;
  const int lane = threadIdx.x & 63;
  const float* __restrict__ w = W.p[g] + (size_t)n * IN;
  const float* __restrict__ x = xin + (size_t)(shared_x ? g / hc : g) * B * IN;
  const float* __restrict__ mk = MASK ? mask + (size_t)g * B * IN : nullptr;
  const float bv = has_bias ? bias.p[g][n] : 0.f;
  for (int b = 0; b < B; ++b) {
    float s = 0.f;
    for (int k = lane; k < IN; k += 64) {
      float xv = x[(size_t)b * IN + k];
      if (MASK) xv *= mk[(size_t)b * IN + k] * msc;
      s += xv * w[k];
    }
    s = wave_sum(s);
    if (lane == 0) out[((size_t)g * B + b) * OUT + n] = s + bv;
  }
}

// sums[g][0][c] = sum_b h ; sums[g][1][c] = sum_b h^2     (double, for the SyncBN all-reduce)
__global__ void heads_bn_stats_kernel(const float* __restrict__ h, double* __restrict__ sums, int G, int B, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G * C) return;
  const int g = i / C, c = i - g * C;
  double s = 0.0, q = 0.0;
  for (int b = 0; b < B; ++b) {
    const double v = (double)h[((size_t)g * B + b) * C + c];
    s += v;
    q += v * v;
  }
  sums[((size_t)g * 2) * C + c] = s;
  sums[((size_t)g * 2 + 1) * C + c] = q;
}

// a = relu(bn(h)) * mask2 * msc ; training: batch stats from sums + running update ; eval: running stats
__global__ void heads_bn_apply_kernel(const float* __restrict__ h, const double* __restrict__ sums, double count,
                                      const PtrTab gamma, const PtrTab beta, const PtrTabW rmean, const PtrTabW rvar,
                                      const float* __restrict__ mask2, float msc, float momentum, float eps,
                                      int training, float* __restrict__ a, float* __restrict__ mean_invstd, int G,
                                      int B, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G * C) return;
  const int g = i / C, c = i - g * C;
  float mean, invstd;
  if (training) {
    const double m = sums[((size_t)g * 2) * C + c] / count;
    double var = sums[((size_t)g * 2 + 1) * C + c] / count - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    invstd = (float)(1.0 / sqrt(var + (double)eps));
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean.p[g][c] = (1.f - momentum) * rmean.p[g][c] + momentum * mean;
    rvar.p[g][c] = (1.f - momentum) * rvar.p[g][c] + momentum * (float)unb;
  } else {
    mean = rmean.p[g][c];
    invstd = 1.f / sqrtf(rvar.p[g][c] + eps);
  }
  mean_invstd[((size_t)g * 2) * C + c] = mean;
  mean_invstd[((size_t)g * 2 + 1) * C + c] = invstd;
  const float sc = gamma.p[g][c] * invstd, sh = beta.p[g][c] - mean * sc;
  for (int b = 0; b < B; ++b) {
    const size_t ad = ((size_t)g * B + b) * C + c;
    float v = fmaxf(h[ad] * sc + sh, 0.f);
    if (mask2) v *= mask2[ad] * msc;
    a[ad] = v;
  }
}

// softmax cross-entropy per (g, b) row, one wave per row.
// loss_rows[g*B+b] = lse - logit[label] ; dlogits = (softmax - onehot) * gscale
__global__ __launch_bounds__(256) void heads_ce_kernel(const float* __restrict__ logits,
                                                      const int64_t* __restrict__ labels, int label_stride, int hc,
                                                      float* __restrict__ loss_rows, float* __restrict__ dlogits,
                                                      float gscale, int G, int B, int K) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= G * B) return;
  const int lane = threadIdx.x & 63;
  const int g = row / B, b = row - g * B;
  const int64_t lab = labels[(size_t)b * label_stride + (g % hc)];
  const float* __restrict__ z = logits + (size_t)row * K;
  float mx = -INFINITY;
  for (int k = lane; k < K; k += 64) mx = fmaxf(mx, z[k]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int k = lane; k < K; k += 64) se += expf(z[k] - mx);
  se = wave_sum(se);
  const float lse = mx + logf(se);
  if (lane == 0) loss_rows[row] = lse - z[lab];
  if (dlogits) {
    const float inv = 1.f / se;
    for (int k = lane; k < K; k += 64) {
      const float p = expf(z[k] - mx) * inv;
      dlogits[(size_t)row * K + k] = (p - (k == lab ? 1.f : 0.f)) * gscale;
    }
  }
}

// total = scale * sum_rows loss_rows (one workgroup, fixed order): the mean over heads and batch of utils.py:377-387
__global__ __launch_bounds__(256) void heads_ce_total_kernel(const float* __restrict__ loss_rows, int n, float scale,
                                                             float* __restrict__ total) {
  __shared__ float sh[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += loss_rows[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = sh[0] * scale;
}

// Dropout keep-masks (model.py:79,85: Dropout(0.3) before each Linear of MLPv2) from Philox4x32-10 (Salmon et al.,
// SC'11): element e of the concatenation m1 | m2 is word e % 4 of the block with counter (e / 4, offset) under key
// seed; keep iff word >= p * 2^32.  A mask is a pure function of (seed, offset, e): no generator state on the device.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// state != nullptr: (seed, offset) are read from device memory -- the form a HIP graph can replay with fresh masks (kernel
// arguments are frozen at capture; dropout_state_advance_kernel, captured behind this launch, bumps the offset)
__global__ __launch_bounds__(256) void dropout_masks_kernel(unsigned long long seed, unsigned long long offset,
                                                            const unsigned long long* __restrict__ state,
                                                            unsigned thresh, float* __restrict__ m1, size_t n1,
                                                            float* __restrict__ m2, size_t n2) {
  const size_t blk = (size_t)blockIdx.x * 256 + threadIdx.x;      // one Philox block = 4 elements per thread
  if (blk * 4 >= n1 + n2) return;
  if (state) {
    seed = state[0];
    offset = state[1];
  }
  unsigned r[4];
  philox4x32_10((unsigned)blk, (unsigned)(blk >> 32), (unsigned)offset, (unsigned)(offset >> 32), (unsigned)seed,
                (unsigned)(seed >> 32), r);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const size_t e = blk * 4 + i;
    const float keep = r[i] >= thresh ? 1.f : 0.f;
    if (e < n1) m1[e] = keep;
    else if (e < n1 + n2) m2[e - n1] = keep;
  }
}

__global__ void dropout_state_advance_kernel(unsigned long long* state) { state[1] += 1ull; }

// dW[g][n][k] = sum_b dout[g][b][n] * Xm(g)[b][k] ; dbias[g][n] = sum_b dout[g][b][n]
template <bool MASK>
__global__ __launch_bounds__(256) void heads_linear_bwd_w_kernel(const float* __restrict__ dout,
                                                                const float* __restrict__ xin, int shared_x, int hc,
                                                                const float* __restrict__ mask, float msc,
                                                                float* __restrict__ dW, float* __restrict__ dbias,
                                                                int B, int IN, int OUT) {
  const int g = blockIdx.y;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= OUT) return;
  const int lane = threadIdx.x & 63;
  const float* __restrict__ x = xin + (size_t)(shared_x ? g / hc : g) * B * IN;
  const float* __restrict__ mk = MASK ? mask + (size_t)g * B * IN : nullptr;
  float db = 0.f;
  for (int k = lane; k < IN; k += 64) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      const float d = dout[((size_t)g * B + b) * OUT + n];
      float xv = x[(size_t)b * IN + k];
      if (MASK) xv *= mk[(size_t)b * IN + k] * msc;
      s += d * xv;
    }
    dW[((size_t)g * OUT + n) * IN + k] = s;
  }
  if (dbias && lane == 0) {
    for (int b = 0; b < B; ++b) db += dout[((size_t)g * B + b) * OUT + n];
    dbias[(size_t)g * OUT + n] = db;
  }
}

// dx[g][b][k] = (mask ? mask*msc : 1) * sum_n dout[g][b][n] * W[g][n][k]
// Weight-bandwidth bound (20 heads x OUT x IN floats, 16 rows): a workgroup owns 32 consecutive k of one
// head; its 8 thread groups take n = r, r+8, ... (so 8 weight rows are in flight per step) and are
// combined through LDS in a fixed order.  BB rows of dout at a time.
template <bool MASK, int BB>
__global__ __launch_bounds__(256) void heads_linear_bwd_x_kernel(const float* __restrict__ dout, const PtrTab W,
                                                                const float* __restrict__ mask, float msc,
                                                                float* __restrict__ dx, int B, int IN, int OUT) {
  __shared__ float red[8][BB][33];
  const int g = blockIdx.y, b0 = blockIdx.z * BB;
  const int kl = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int k = blockIdx.x * 32 + kl;
  const float* __restrict__ w = W.p[g];
  float acc[BB];
#pragma unroll
  for (int i = 0; i < BB; ++i) acc[i] = 0.f;
  if (k < IN) {
    for (int n = r; n < OUT; n += 8) {
      const float wv = w[(size_t)n * IN + k];
#pragma unroll
      for (int i = 0; i < BB; ++i) {
        if (b0 + i < B) acc[i] += dout[((size_t)g * B + b0 + i) * OUT + n] * wv;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < BB; ++i) red[r][i][kl] = acc[i];
  __syncthreads();
  // 256 threads finish BB x 32 outputs: thread -> (row i = tid / 32 + 8*j, k = tid % 32)
  for (int i = r; i < BB; i += 8) {
    if (b0 + i < B && k < IN) {
      float v = red[0][i][kl];
#pragma unroll
      for (int q = 1; q < 8; ++q) v += red[q][i][kl];
      const size_t ad = ((size_t)g * B + b0 + i) * IN + k;
      dx[ad] = MASK ? v * mask[ad] * msc : v;
    }
  }
}

// BN1d backward, stage 1: g' = da * mask2*msc * (bnout > 0);  sums[g][0][c] = sum g', [1] = sum g' xhat
__global__ void heads_bn_bwd_stats_kernel(const float* __restrict__ da, const float* __restrict__ h,
                                          const float* __restrict__ mean_invstd, const PtrTab gamma,
                                          const PtrTab beta, const float* __restrict__ mask2, float msc,
                                          double* __restrict__ sums, int G, int B, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G * C) return;
  const int g = i / C, c = i - g * C;
  const float mean = mean_invstd[((size_t)g * 2) * C + c], invstd = mean_invstd[((size_t)g * 2 + 1) * C + c];
  const float ga = gamma.p[g][c], be = beta.p[g][c];
  double s = 0.0, q = 0.0;
  for (int b = 0; b < B; ++b) {
    const size_t ad = ((size_t)g * B + b) * C + c;
    const float xh = (h[ad] - mean) * invstd;
    float gv = da[ad];
    if (mask2) gv *= mask2[ad] * msc;
    if (!(xh * ga + be > 0.f)) gv = 0.f;
    s += (double)gv;
    q += (double)gv * (double)xh;
  }
  sums[((size_t)g * 2) * C + c] = s;
  sums[((size_t)g * 2 + 1) * C + c] = q;
}
// stage 2: dh = gamma*invstd*(g' - c1 - xhat*c2) ; dgamma = sum g' xhat ; dbeta = sum g'
__global__ void heads_bn_bwd_apply_kernel(const float* __restrict__ da, const float* __restrict__ h,
                                          const float* __restrict__ mean_invstd, const PtrTab gamma,
                                          const PtrTab beta, const float* __restrict__ mask2, float msc,
                                          const double* __restrict__ sums, double count, float* __restrict__ dh,
                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int G, int B,
                                          int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G * C) return;
  const int g = i / C, c = i - g * C;
  const float mean = mean_invstd[((size_t)g * 2) * C + c], invstd = mean_invstd[((size_t)g * 2 + 1) * C + c];
  const float ga = gamma.p[g][c], be = beta.p[g][c];
  const double sg = sums[((size_t)g * 2) * C + c], sgx = sums[((size_t)g * 2 + 1) * C + c];
  const float c1 = (float)(sg / count), c2 = (float)(sgx / count);
  const float A1 = ga * invstd;
  for (int b = 0; b < B; ++b) {
    const size_t ad = ((size_t)g * B + b) * C + c;
    const float xh = (h[ad] - mean) * invstd;
    float gv = da[ad];
    if (mask2) gv *= mask2[ad] * msc;
    if (!(xh * ga + be > 0.f)) gv = 0.f;
    dh[ad] = A1 * (gv - c1 - xh * c2);
  }
  dgamma[(size_t)g * C + c] = (float)sgx;
  dbeta[(size_t)g * C + c] = (float)sg;
}

// out[m][i] = sum_{g in modality m} src[g][i]   (fixed order)
__global__ void heads_sum_groups_kernel(const float* __restrict__ src, float* __restrict__ out, int hc, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int m = blockIdx.y;
  float s = 0.f;
  for (int g = 0; g < hc; ++g) s += src[((size_t)(m * hc + g)) * n + i];
  out[(size_t)m * n + i] = s;
}

// y[r][c] = relu?(x[r][c]*s[c] + h[c])    rows x C row-major (eval-mode BN1d of the SK feature bank)
__global__ void rowwise_affine_kernel(const float* __restrict__ x, const float* __restrict__ ss, int relu,
                                      float* __restrict__ y, int C, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const float v = x[i] * ss[c] + ss[C + c];
    y[i] = relu ? fmaxf(v, 0.f) : v;
  }
}

// ---------------------------------------------------------------------------------------- the same on the matrix cores
// The one-wave-per-output-column kernels above walk the batch rows serially (8 loads + a wave butterfly per (row, column)):
// 0.65 ms for the first Linear of the 20 heads at 128 clips, 30x the time the 34 MB of weights take to read.  The MFMA
// versions (v_mfma_f32_16x16x4_f32: fp32 in, fp32 out, only the summation order changes) give each wave a 16-wide output
// tile and stream both operands from memory / L2 in fragment layout -- no LDS: lane (i = lane & 15, q = lane >> 4) holds
// A[i][k = q], B[k = q][i]; a float4 per lane covers 16 values of K in 4 MFMAs (which 4 of the 16 a lane group gets is
// the same for both operands, which is all a sum over K needs).  Shapes they take: IN % 16 == 0 (the 512-wide layers of
// every configuration); anything else stays on the kernels above.
typedef float hf32x4 __attribute__((ext_vector_type(4)));

// out[g][b][n] = sum_k Xm(g)[b][k] W[g][n][k] (+ bias): wave = 16 columns n x up to 16 BT rows
template <bool MASK, int BT>
__global__ __launch_bounds__(256) void heads_linear_fwd_mfma_kernel(const float* __restrict__ xin, int shared_x, int hc,
                                                                   const float* __restrict__ mask, float msc,
                                                                   const PtrTab W, const PtrTab bias, int has_bias,
                                                                   float* __restrict__ out, int B, int IN, int OUT) {
  const int g = blockIdx.y, lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
  const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16, b00 = blockIdx.z * (16 * BT);
  if (n0 >= OUT) return;
  const float* __restrict__ x = xin + (size_t)(shared_x ? g / hc : g) * B * IN;
  const float* __restrict__ mk = MASK ? mask + (size_t)g * B * IN : nullptr;
  const bool nok = n0 + i < OUT;
  const float* __restrict__ wrow = W.p[g] + (size_t)(nok ? n0 + i : 0) * IN + 4 * q;
  hf32x4 acc[BT];
  size_t xo[BT];
  bool bok[BT];
#pragma unroll
  for (int t = 0; t < BT; ++t) {
    acc[t] = (hf32x4){0.f, 0.f, 0.f, 0.f};
    const int b = b00 + 16 * t + i;
    bok[t] = b < B;
    xo[t] = (size_t)(bok[t] ? b : 0) * IN + 4 * q;
  }
  for (int k0 = 0; k0 < IN; k0 += 16) {
    hf32x4 w4 = *(const hf32x4*)(wrow + k0);
    if (!nok) w4 = (hf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < BT; ++t) {
      hf32x4 x4 = *(const hf32x4*)(x + xo[t] + k0);
      if (MASK) {
        const hf32x4 m4 = *(const hf32x4*)(mk + xo[t] + k0);
#pragma unroll
        for (int c = 0; c < 4; ++c) x4[c] *= m4[c] * msc;
      }
      if (!bok[t]) x4 = (hf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4[c], w4[c], acc[t], 0, 0, 0);
    }
  }
  const float bv = (has_bias && nok) ? bias.p[g][n0 + i] : 0.f;
#pragma unroll
  for (int t = 0; t < BT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = b00 + 16 * t + 4 * q + r;                     // C/D: column (n) = lane & 15, rows (b) 4 q + r
      if (b < B && nok) out[((size_t)g * B + b) * OUT + n0 + i] = acc[t][r] + bv;
    }
}

// dW[g][n][k] = sum_b dout[g][b][n] Xm(g)[b][k]; dbias[g][n] = sum_b dout[g][b][n]: wave = 16 rows n x 64 columns k
template <bool MASK>
__global__ __launch_bounds__(256) void heads_linear_bwd_w_mfma_kernel(const float* __restrict__ dout,
                                                                     const float* __restrict__ xin, int shared_x, int hc,
                                                                     const float* __restrict__ mask, float msc,
                                                                     float* __restrict__ dW, float* __restrict__ dbias,
                                                                     int B, int IN, int OUT) {
  const int g = blockIdx.y, lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16, k0 = (blockIdx.z * 4 + (threadIdx.x >> 6)) * 64;
  if (k0 >= IN) return;
  const float* __restrict__ x = xin + (size_t)(shared_x ? g / hc : g) * B * IN;
  const float* __restrict__ mk = MASK ? mask + (size_t)g * B * IN : nullptr;
  const float* __restrict__ d = dout + (size_t)g * B * OUT;
  const bool nok = n0 + i < OUT;
  hf32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (hf32x4){0.f, 0.f, 0.f, 0.f};
  float dsum = 0.f;
  auto step = [&](int b, bool ok) __attribute__((always_inline)) {
    const float a = (ok && nok) ? d[(size_t)b * OUT + n0 + i] : 0.f;       // A[m = n][k = b]
    dsum += a;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = k0 + 16 * t + i;
      float xv = (ok && k < IN) ? x[(size_t)b * IN + k] : 0.f;              // B[k = b][n = k]
      if (MASK) xv *= (ok && k < IN) ? mk[(size_t)b * IN + k] * msc : 0.f;
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv, acc[t], 0, 0, 0);
    }
  };
  int b0 = 0;
  for (; b0 + 16 <= B; b0 += 16) {
#pragma unroll
    for (int u = 0; u < 4; ++u) step(b0 + 4 * u + q, true);
  }
  for (; b0 < B; b0 += 4) step(b0 + q, b0 + q < B);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 4 * q + r, k = k0 + 16 * t + i;                     // C/D: column (k) = lane & 15, rows (n) 4 q + r
      if (n < OUT && k < IN) dW[((size_t)g * OUT + n) * IN + k] = acc[t][r];
    }
  if (dbias && k0 == 0) {                                                    // lanes i, i + 16, i + 32, i + 48 hold b = q (mod 4)
    dsum += __shfl_xor(dsum, 16);
    dsum += __shfl_xor(dsum, 32);
    if (q == 0 && nok) dbias[(size_t)g * OUT + n0 + i] = dsum;
  }
}

// dx[g][b][k] = (mask ? mask * msc : 1) * sum_n dout[g][b][n] W[g][n][k]: wave = 16 rows b x 32 columns k
template <bool MASK>
__global__ __launch_bounds__(256) void heads_linear_bwd_x_mfma_kernel(const float* __restrict__ dout, const PtrTab W,
                                                                     const float* __restrict__ mask, float msc,
                                                                     float* __restrict__ dx, int B, int IN, int OUT) {
  const int g = blockIdx.y, lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
  const int k0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 32, b0 = blockIdx.z * 16;
  if (k0 >= IN) return;
  const float* __restrict__ w = W.p[g];
  const bool bok = b0 + i < B;
  const float* __restrict__ drow = dout + ((size_t)g * B + (bok ? b0 + i : 0)) * OUT;
  hf32x4 acc[2] = {(hf32x4){0.f, 0.f, 0.f, 0.f}, (hf32x4){0.f, 0.f, 0.f, 0.f}};
  const bool k_ok0 = k0 + i < IN, k_ok1 = k0 + 16 + i < IN;
  auto step = [&](int n, bool ok) __attribute__((always_inline)) {
    const float a = (ok && bok) ? drow[n] : 0.f;                            // A[m = b][k = n]
    const float w0 = (ok && k_ok0) ? w[(size_t)n * IN + k0 + i] : 0.f;       // B[k = n][n = k]
    const float w1 = (ok && k_ok1) ? w[(size_t)n * IN + k0 + 16 + i] : 0.f;
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w1, acc[1], 0, 0, 0);
  };
  int n0 = 0;
  for (; n0 + 32 <= OUT; n0 += 32) {                                        // 8 independent request pairs in flight
#pragma unroll
    for (int u = 0; u < 8; ++u) step(n0 + 4 * u + q, true);
  }
  for (; n0 < OUT; n0 += 4) step(n0 + q, n0 + q < OUT);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = b0 + 4 * q + r, k = k0 + 16 * t + i;
      if (b < B && k < IN) {
        const size_t ad = ((size_t)g * B + b) * IN + k;
        dx[ad] = MASK ? acc[t][r] * mask[ad] * msc : acc[t][r];
      }
    }
}

static bool heads_mfma_enabled() {
  static const bool on = []() {
    const char* e = getenv("SELAVI_HEADS_MFMA");
    return !(e && e[0] == '0');
  }();
  return on;
}

static int fill_tab(PtrTab& t, const void* const* host, int G) {
  if (G > MAXG) return -1;
  for (int g = 0; g < G; ++g) t.p[g] = host ? (const float*)host[g] : nullptr;
  return 0;
}
static int fill_tabw(PtrTabW& t, const void* const* host, int G) {
  if (G > MAXG) return -1;
  for (int g = 0; g < G; ++g) t.p[g] = host ? (float*)host[g] : nullptr;
  return 0;
}

}  // namespace slv

using namespace slv;

extern "C" {

int slv_heads_linear_fwd(const float* x, int shared_x, int hc, const float* mask, float mask_scale,
                         const void* const* W, const void* const* bias, float* out, int G, int B, int IN, int OUT,
                         slv_stream_t stream) {
  SLV_CHECK_ARG(x && W && out && G > 0 && G <= MAXG && B > 0 && IN > 0 && OUT > 0 && hc > 0, "bad argument");
  PtrTab tw, tb;
  fill_tab(tw, W, G);
  fill_tab(tb, bias, G);
  if (heads_mfma_enabled() && (IN & 15) == 0) {
    // rows per wave: 16 BT; the batch is cut into blockIdx.z slices of that many rows
#define SLV_HFWD(M_, BT_)                                                                                              \
  hipLaunchKernelGGL((heads_linear_fwd_mfma_kernel<M_, BT_>), dim3(((OUT + 15) / 16 + 3) / 4, G, (B + 16 * BT_ - 1) / (16 * BT_)), \
                     dim3(256), 0, (hipStream_t)stream, x, shared_x, hc, mask, mask_scale, tw, tb, bias != nullptr, out, B, IN, OUT)
    if (B <= 16) { if (mask) SLV_HFWD(true, 1); else SLV_HFWD(false, 1); }
    else if (B <= 32) { if (mask) SLV_HFWD(true, 2); else SLV_HFWD(false, 2); }
    else { if (mask) SLV_HFWD(true, 4); else SLV_HFWD(false, 4); }
#undef SLV_HFWD
    SLV_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid((OUT + 3) / 4, G);
  if (mask)
    hipLaunchKernelGGL((heads_linear_fwd_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, x, shared_x, hc,
                       mask, mask_scale, tw, tb, bias != nullptr, out, B, IN, OUT);
  else
    hipLaunchKernelGGL((heads_linear_fwd_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, x, shared_x, hc,
                       mask, mask_scale, tw, tb, bias != nullptr, out, B, IN, OUT);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_heads_bn_stats(const float* h, double* sums, int G, int B, int C, slv_stream_t stream) {
  SLV_CHECK_ARG(h && sums && G > 0 && B > 0 && C > 0, "bad argument");
  hipLaunchKernelGGL(heads_bn_stats_kernel, dim3((G * C + 127) / 128), dim3(128), 0, (hipStream_t)stream, h, sums, G,
                     B, C);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_heads_bn_apply(const float* h, const double* sums, double count, const void* const* gamma,
                       const void* const* beta, const void* const* running_mean, const void* const* running_var,
                       const float* mask2, float mask_scale, float momentum, float eps, int training, float* a,
                       float* mean_invstd, int G, int B, int C, slv_stream_t stream) {
  SLV_CHECK_ARG(h && gamma && beta && running_mean && running_var && a && mean_invstd && G > 0 && G <= MAXG,
                "bad argument");
  SLV_CHECK_ARG(!training || (sums && count > 0), "training mode needs batch sums");
  PtrTab tg, tb;
  PtrTabW tm, tv;
  fill_tab(tg, gamma, G);
  fill_tab(tb, beta, G);
  fill_tabw(tm, running_mean, G);
  fill_tabw(tv, running_var, G);
  hipLaunchKernelGGL(heads_bn_apply_kernel, dim3((G * C + 127) / 128), dim3(128), 0, (hipStream_t)stream, h, sums,
                     count, tg, tb, tm, tv, mask2, mask_scale, momentum, eps, training, a, mean_invstd, G, B, C);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_heads_ce(const float* logits, const int64_t* labels, int label_stride, int hc, float* loss_rows,
                 float* dlogits, float grad_scale, int G, int B, int K, slv_stream_t stream) {
  SLV_CHECK_ARG(logits && labels && loss_rows && G > 0 && B > 0 && K > 0 && hc > 0, "bad argument");
  hipLaunchKernelGGL(heads_ce_kernel, dim3((G * B + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, labels,
                     label_stride, hc, loss_rows, dlogits, grad_scale, G, B, K);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_heads_ce_total(const float* loss_rows, int64_t n, float scale, float* total, slv_stream_t stream) {
  SLV_CHECK_ARG(loss_rows && total && n > 0 && n < 0x7FFFFFFF, "bad argument");
  hipLaunchKernelGGL(heads_ce_total_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_rows, (int)n, scale, total);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_dropout_masks(uint64_t seed, uint64_t offset, float p, float* m1, int64_t n1, float* m2, int64_t n2,
                      slv_stream_t stream) {
  SLV_CHECK_ARG(m1 && n1 > 0 && n2 >= 0 && (m2 || n2 == 0) && p >= 0.f && p < 1.f, "bad argument");
  const double t = (double)(p * 4294967296.0f);             // float product, as the oracle forms it
  const unsigned thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
  const size_t blocks4 = ((size_t)(n1 + n2) + 3) / 4;
  hipLaunchKernelGGL(dropout_masks_kernel, dim3((unsigned)((blocks4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (unsigned long long)seed, (unsigned long long)offset, (const unsigned long long*)nullptr, thresh, m1,
                     (size_t)n1, m2, (size_t)n2);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_dropout_masks_dev(uint64_t* state, float p, float* m1, int64_t n1, float* m2, int64_t n2, slv_stream_t stream) {
  SLV_CHECK_ARG(state && m1 && n1 > 0 && n2 >= 0 && (m2 || n2 == 0) && p >= 0.f && p < 1.f, "bad argument");
  const double t = (double)(p * 4294967296.0f);
  const unsigned thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
  const size_t blocks4 = ((size_t)(n1 + n2) + 3) / 4;
  hipLaunchKernelGGL(dropout_masks_kernel, dim3((unsigned)((blocks4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     0ull, 0ull, (const unsigned long long*)state, thresh, m1, (size_t)n1, m2, (size_t)n2);
  hipLaunchKernelGGL(dropout_state_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)state);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_heads_linear_bwd_w(const float* dout, const float* x, int shared_x, int hc, const float* mask,
                           float mask_scale, float* dW, float* dbias, int G, int B, int IN, int OUT,
                           slv_stream_t stream) {
  SLV_CHECK_ARG(dout && x && dW && G > 0 && B > 0 && IN > 0 && OUT > 0 && hc > 0, "bad argument");
  if (heads_mfma_enabled() && (IN & 15) == 0) {
    const dim3 gm((OUT + 15) / 16, G, ((IN + 63) / 64 + 3) / 4);
    if (mask)
      hipLaunchKernelGGL((heads_linear_bwd_w_mfma_kernel<true>), gm, dim3(256), 0, (hipStream_t)stream, dout, x, shared_x, hc,
                         mask, mask_scale, dW, dbias, B, IN, OUT);
    else
      hipLaunchKernelGGL((heads_linear_bwd_w_mfma_kernel<false>), gm, dim3(256), 0, (hipStream_t)stream, dout, x, shared_x, hc,
                         mask, mask_scale, dW, dbias, B, IN, OUT);
    SLV_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid((OUT + 3) / 4, G);
  if (mask)
    hipLaunchKernelGGL((heads_linear_bwd_w_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, dout, x,
                       shared_x, hc, mask, mask_scale, dW, dbias, B, IN, OUT);
  else
    hipLaunchKernelGGL((heads_linear_bwd_w_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, dout, x,
                       shared_x, hc, mask, mask_scale, dW, dbias, B, IN, OUT);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_heads_linear_bwd_x(const float* dout, const void* const* W, const float* mask, float mask_scale, float* dx,
                           int G, int B, int IN, int OUT, slv_stream_t stream) {
  SLV_CHECK_ARG(dout && W && dx && G > 0 && G <= MAXG && B > 0 && IN > 0 && OUT > 0, "bad argument");
  PtrTab tw;
  fill_tab(tw, W, G);
  if (heads_mfma_enabled() && (IN & 15) == 0) {
    const dim3 gm(((IN + 31) / 32 + 3) / 4, G, (B + 15) / 16);
    if (mask)
      hipLaunchKernelGGL((heads_linear_bwd_x_mfma_kernel<true>), gm, dim3(256), 0, (hipStream_t)stream, dout, tw, mask,
                         mask_scale, dx, B, IN, OUT);
    else
      hipLaunchKernelGGL((heads_linear_bwd_x_mfma_kernel<false>), gm, dim3(256), 0, (hipStream_t)stream, dout, tw, mask,
                         mask_scale, dx, B, IN, OUT);
    SLV_LAUNCH_CHECK();
    return 0;
  }
  constexpr int BB = 16;
  dim3 grid((IN + 31) / 32, G, (B + BB - 1) / BB);
  if (mask)
    hipLaunchKernelGGL((heads_linear_bwd_x_kernel<true, BB>), grid, dim3(256), 0, (hipStream_t)stream, dout, tw,
                       mask, mask_scale, dx, B, IN, OUT);
  else
    hipLaunchKernelGGL((heads_linear_bwd_x_kernel<false, BB>), grid, dim3(256), 0, (hipStream_t)stream, dout, tw,
                       mask, mask_scale, dx, B, IN, OUT);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_heads_bn_bwd_stats(const float* da, const float* h, const float* mean_invstd, const void* const* gamma,
                           const void* const* beta, const float* mask2, float mask_scale, double* sums, int G, int B,
                           int C, slv_stream_t stream) {
  SLV_CHECK_ARG(da && h && mean_invstd && gamma && beta && sums && G > 0 && G <= MAXG, "bad argument");
  PtrTab tg, tb;
  fill_tab(tg, gamma, G);
  fill_tab(tb, beta, G);
  hipLaunchKernelGGL(heads_bn_bwd_stats_kernel, dim3((G * C + 127) / 128), dim3(128), 0, (hipStream_t)stream, da, h,
                     mean_invstd, tg, tb, mask2, mask_scale, sums, G, B, C);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_heads_bn_bwd_apply(const float* da, const float* h, const float* mean_invstd, const void* const* gamma,
                           const void* const* beta, const float* mask2, float mask_scale, const double* sums,
                           double count, float* dh, float* dgamma, float* dbeta, int G, int B, int C,
                           slv_stream_t stream) {
  SLV_CHECK_ARG(da && h && mean_invstd && gamma && beta && sums && dh && dgamma && dbeta && G > 0 && G <= MAXG,
                "bad argument");
  PtrTab tg, tb;
  fill_tab(tg, gamma, G);
  fill_tab(tb, beta, G);
  hipLaunchKernelGGL(heads_bn_bwd_apply_kernel, dim3((G * C + 127) / 128), dim3(128), 0, (hipStream_t)stream, da, h,
                     mean_invstd, tg, tb, mask2, mask_scale, sums, count, dh, dgamma, dbeta, G, B, C);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_heads_sum_groups(const float* src, float* out, int hc, int64_t n, slv_stream_t stream) {
  SLV_CHECK_ARG(src && out && hc > 0 && n > 0, "bad argument");
  hipLaunchKernelGGL(heads_sum_groups_kernel, dim3((unsigned)((n + 255) / 256), 2), dim3(256), 0,
                     (hipStream_t)stream, src, out, hc, (size_t)n);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_rowwise_affine(const float* x, const float* scale_shift, int relu, float* y, int64_t rows, int C,
                       slv_stream_t stream) {
  SLV_CHECK_ARG(x && scale_shift && y && rows > 0 && C > 0, "bad argument");
  const size_t total = (size_t)rows * C;
  size_t b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(rowwise_affine_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, x, scale_shift,
                     relu, y, C, total);
  SLV_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

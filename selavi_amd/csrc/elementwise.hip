// HBM-bound kernels around the convolutions: BatchNorm statistics/finalize, the block-tail
// "BN + residual + ReLU" materialisation, BN backward reductions, pools and the fused SGD step.
// Reference semantics: torch BatchNorm3d/2d (train + eval), ReLU, residual add, AdaptiveAvgPool,
// MaxPool2d(3,2,1) as used by the torchvision nets of /root/reference/model.py:95,114; SGD of
// main.py:132-137.
#include "common.hpp"
#include "../../include/selavi_hip.h"

namespace slv {

typedef float v4f __attribute__((ext_vector_type(4)));
// Streaming accesses of tensors far larger than the 4 MB L2 slices: non-temporal loads/stores keep them from
// evicting the operands of the neighbouring GEMMs (bn_bwd_apply on the 462 MB layer-1 tensors: 5.0 -> 5.9 TB/s).
// NT is a template flag chosen by tensor size; small late-layer tensors stay cacheable for their consumer.
constexpr size_t NT_MIN_BYTES = 32u << 20;
template <bool NT>
__device__ __forceinline__ void ld4(float* dst, const float* src) {
  if constexpr (NT) *(v4f*)dst = __builtin_nontemporal_load((const v4f*)src);
  else *(v4f*)dst = *(const v4f*)src;
}
template <bool NT>
__device__ __forceinline__ void st4(float* dst, const float* src) {
  if constexpr (NT) __builtin_nontemporal_store(*(const v4f*)src, (v4f*)dst);
  else *(v4f*)dst = *(const v4f*)src;
}

// ------------------------------------------------------------------ BN statistics
// partial[c][nblk] (float, written by the conv epilogue) -> sums[2C] (double): sum, sum of squares
__global__ __launch_bounds__(256) void bn_partials_to_sums_kernel(const float* __restrict__ ps,
                                                                 const float* __restrict__ pq, int nblk,
                                                                 int C, double* __restrict__ sums) {
  __shared__ double sh[2][256];
  const int c = blockIdx.x;
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) {
    s += (double)ps[(size_t)c * nblk + i];
    q += (double)pq[(size_t)c * nblk + i];
  }
  sh[0][threadIdx.x] = s;
  sh[1][threadIdx.x] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sums[c] = sh[0][0];
    sums[C + c] = sh[1][0];
  }
}

// sums -> mean/invstd (saved for backward), scale/shift (consumer prologue), running stats update
__device__ __forceinline__ void bn_finalize_one(int c, double sum, double sq, double count,
                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                                float eps, float* __restrict__ mean_invstd,
                                                float* __restrict__ scale_shift, int C) {
  const double mean = sum / count;
  double var = sq / count - mean * mean;  // biased
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  mean_invstd[c] = meanf;
  mean_invstd[C + c] = invstd;
  const float sc = gamma[c] * invstd;
  scale_shift[c] = sc;
  scale_shift[C + c] = beta[c] - meanf * sc;
  if (rmean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * meanf;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
  }
}
__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ rmean,
                                   float* __restrict__ rvar, float momentum, float eps,
                                   float* __restrict__ mean_invstd, float* __restrict__ scale_shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  bn_finalize_one(c, sums[c], sums[C + c], count, gamma, beta, rmean, rvar, momentum, eps, mean_invstd, scale_shift, C);
}
// single-process BatchNorm: partial reduction and finalize in one launch (one workgroup per channel)
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ ps,
                                                               const float* __restrict__ pq, int nblk, double count,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ rmean,
                                                               float* __restrict__ rvar, float momentum, float eps,
                                                               float* __restrict__ mean_invstd,
                                                               float* __restrict__ scale_shift, int C) {
  __shared__ double sh[2][256];
  const int c = blockIdx.x;
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) {
    s += (double)ps[(size_t)c * nblk + i];
    q += (double)pq[(size_t)c * nblk + i];
  }
  sh[0][threadIdx.x] = s;
  sh[1][threadIdx.x] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
    bn_finalize_one(c, sh[0][0], sh[1][0], count, gamma, beta, rmean, rvar, momentum, eps, mean_invstd, scale_shift, C);
}

__global__ void bn_eval_params_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rmean, const float* __restrict__ rvar, float eps,
                                      float* __restrict__ mean_invstd, float* __restrict__ scale_shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(rvar[c] + eps);
  const float sc = gamma[c] * invstd;
  if (mean_invstd) {
    mean_invstd[c] = rmean[c];
    mean_invstd[C + c] = invstd;
  }
  scale_shift[c] = sc;
  scale_shift[C + c] = beta[c] - rmean[c] * sc;
}

// ------------------------------------------------------------------ block tail (forward)
// out = relu?( x*s + h  +  (res ? (rss ? res*rs + rh : res) : 0) ),  tensors [Bn][C][P]
// V = 4: P % 4 == 0 and 16-byte aligned tensors -> float4 accesses (4 consecutive elements share a channel).
// 32-bit exact division by P and C (FastDiv): the flat index stays below 2^32 (checked by the entry point).
template <int V, bool NT>
__global__ __launch_bounds__(256) void bn_act_kernel(const float* __restrict__ x, const float* __restrict__ ss,
                                                    const float* __restrict__ res, const float* __restrict__ rss,
                                                    int relu, float* __restrict__ out, int C, const FastDiv dP,
                                                    const FastDiv dC, unsigned units) {
  for (unsigned u = blockIdx.x * 256u + threadIdx.x; u < units; u += gridDim.x * 256u) {
    const unsigned i = u * V;
    const unsigned row = fdiv(i, dP);
    const unsigned c = row - fdiv(row, dC) * (unsigned)C;
    const float s = ss[c], h = ss[C + c];
    float rs = 1.f, rh = 0.f;
    if (res && rss) { rs = rss[c]; rh = rss[C + c]; }
    float xv[V], rv[V], ov[V];
    if constexpr (V == 4) {
      ld4<NT>(xv, x + i);
      if (res) ld4<NT>(rv, res + i);
    } else {
      xv[0] = x[i];
      if (res) rv[0] = res[i];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float v = xv[j] * s + h;
      if (res) v += rss ? (rv[j] * rs + rh) : rv[j];
      ov[j] = relu ? fmaxf(v, 0.f) : v;
    }
    if constexpr (V == 4) *(float4*)(out + i) = *(float4*)ov;   // the block output feeds the next conv: cacheable
    else out[i] = ov[0];
  }
}

// ------------------------------------------------------------------ BN backward reductions
// For channel c (blockIdx.x) and slice blockIdx.y of the (b,p) space:
//   g' = mask * g ;  partial = { sum g', sum g' * xhat(x) [, sum g' * xhat2(x2)] }
// MASK 0: none; 1: own BN output > 0 (s*x+h); 2: external tensor v > 0 (and g' is written to gout)
template <int MASK, bool TWO, int V, bool NT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ gin,
                                                           const float* __restrict__ x,
                                                           const float* __restrict__ mi,  // mean,invstd [2C]
                                                           const float* __restrict__ ss,  // MASK 1
                                                           const float* __restrict__ v,   // MASK 2
                                                           const float* __restrict__ x2, const float* __restrict__ mi2,
                                                           float* __restrict__ gout, float* __restrict__ part,
                                                           float* __restrict__ part2, int C, unsigned P,
                                                           const FastDiv dP, unsigned tot, unsigned per, int nsplit) {
  __shared__ float sh[3][256];
  const int c = blockIdx.x, sp = blockIdx.y;
  const unsigned e0 = sp * per, e1 = (e0 + per < tot) ? e0 + per : tot;   // slice of the (b,p) space; per % V == 0
  const float mean = mi[c], invstd = mi[C + c];
  float mean2 = 0.f, invstd2 = 0.f, s_ = 0.f, h_ = 0.f;
  if (TWO) { mean2 = mi2[c]; invstd2 = mi2[C + c]; }
  if (MASK == 1) { s_ = ss[c]; h_ = ss[C + c]; }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (unsigned e = e0 + threadIdx.x * V; e < e1; e += 256 * V) {
    const unsigned b = fdiv(e, dP);
    const size_t ad = ((size_t)b * C + c) * P + (size_t)(e - b * P);
    float g[V], xv[V], vv[V], x2v[V];
    if constexpr (V == 4) {
      ld4<NT>(g, gin + ad);
      ld4<NT>(xv, x + ad);
      if (MASK == 2) ld4<NT>(vv, v + ad);
      if (TWO) ld4<NT>(x2v, x2 + ad);
    } else {
      g[0] = gin[ad];
      xv[0] = x[ad];
      if (MASK == 2) vv[0] = v[ad];
      if (TWO) x2v[0] = x2[ad];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      if (MASK == 1) g[j] = (xv[j] * s_ + h_ > 0.f) ? g[j] : 0.f;
      if (MASK == 2) g[j] = (vv[j] > 0.f) ? g[j] : 0.f;
      a0 += g[j];
      a1 += g[j] * ((xv[j] - mean) * invstd);
      if (TWO) a2 += g[j] * ((x2v[j] - mean2) * invstd2);
    }
    if (MASK == 2) {
      if constexpr (V == 4) *(float4*)(gout + ad) = *(float4*)g;
      else gout[ad] = g[0];
    }
  }
  sh[0][threadIdx.x] = a0;
  sh[1][threadIdx.x] = a1;
  sh[2][threadIdx.x] = a2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
      sh[2][threadIdx.x] += sh[2][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[((size_t)c * nsplit + sp) * 2 + 0] = sh[0][0];
    part[((size_t)c * nsplit + sp) * 2 + 1] = sh[1][0];
    if (TWO) {
      part2[((size_t)c * nsplit + sp) * 2 + 0] = sh[0][0];
      part2[((size_t)c * nsplit + sp) * 2 + 1] = sh[2][0];
    }
  }
}

// one workgroup per channel: fp64 tree over the slice partials (thousands of slots when they come from
// a dgrad epilogue); fixed order
__device__ __forceinline__ void bn_bwd_block_sums(const float* __restrict__ part, int nsplit, int c, double& a,
                                                  double& b) {
  __shared__ double sh[2][256];
  double x = 0.0, y = 0.0;
  for (int s = threadIdx.x; s < nsplit; s += 256) {
    const float2 v = *(const float2*)(part + ((size_t)c * nsplit + s) * 2);
    x += (double)v.x;
    y += (double)v.y;
  }
  sh[0][threadIdx.x] = x;
  sh[1][threadIdx.x] = y;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  a = sh[0][0];
  b = sh[1][0];
}
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* __restrict__ part, int nsplit, int C,
                                                         double* __restrict__ sums) {
  double a, b;
  bn_bwd_block_sums(part, nsplit, blockIdx.x, a, b);
  if (threadIdx.x == 0) {
    sums[blockIdx.x] = a;
    sums[C + blockIdx.x] = b;
  }
}

// dx = A1*g' + A2 + A3*x  with  A1 = gamma*invstd, A3 = -A1*c2*invstd, A2 = -A1*c1 - A3*mean,
// c1 = sum g'/n, c2 = sum g' xhat / n.   bwd5 = {s, h, A1, A2, A3} (s,h = forward scale/shift for the mask)
__device__ __forceinline__ void bn_bwd_finalize_one(int c, double sg, double sgx, double count,
                                                    const float* __restrict__ gamma, const float* __restrict__ mi,
                                                    const float* __restrict__ ss, float* __restrict__ bwd5,
                                                    float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                    int accumulate, int C) {
  const float invstd = mi[C + c], mean = mi[c];
  const float A1 = gamma[c] * invstd;
  const float c1 = (float)(sg / count), c2 = (float)(sgx / count);
  const float A3 = -A1 * c2 * invstd;
  const float A2 = -A1 * c1 - A3 * mean;
  bwd5[c] = ss ? ss[c] : 0.f;
  bwd5[C + c] = ss ? ss[C + c] : 0.f;
  bwd5[2 * C + c] = A1;
  bwd5[3 * C + c] = A2;
  bwd5[4 * C + c] = A3;
  if (dgamma) {
    if (accumulate) {
      dgamma[c] += (float)sgx;
      dbeta[c] += (float)sg;
    } else {
      dgamma[c] = (float)sgx;
      dbeta[c] = (float)sg;
    }
  }
}
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ sums, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ mi,
                                       const float* __restrict__ ss, float* __restrict__ bwd5,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                       int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  bn_bwd_finalize_one(c, sums[c], sums[C + c], count, gamma, mi, ss, bwd5, dgamma, dbeta, accumulate, C);
}
// single-process: slice partials -> sums -> coefficients in one launch (workgroup per channel)
__global__ __launch_bounds__(256) void bn_bwd_sums_finalize_kernel(const float* __restrict__ part, int nsplit,
                                                                  double count, const float* __restrict__ gamma,
                                                                  const float* __restrict__ mi,
                                                                  const float* __restrict__ ss, float* __restrict__ bwd5,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  int accumulate, int C) {
  double a, b;
  bn_bwd_block_sums(part, nsplit, blockIdx.x, a, b);
  if (threadIdx.x == 0) bn_bwd_finalize_one(blockIdx.x, a, b, count, gamma, mi, ss, bwd5, dgamma, dbeta, accumulate, C);
}

// materialise the gradient w.r.t. a raw conv output: out = A1*mask*g + A2 + A3*x  (bwd5 = s,h,A1,A2,A3)
template <int V, bool NT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                          const float* __restrict__ b5, int relu,
                                                          float* __restrict__ out, int C, const FastDiv dP,
                                                          const FastDiv dC, unsigned units) {
  for (unsigned u = blockIdx.x * 256u + threadIdx.x; u < units; u += gridDim.x * 256u) {
    const unsigned i = u * V;
    const unsigned row = fdiv(i, dP);
    const unsigned c = row - fdiv(row, dC) * (unsigned)C;
    const float s = b5[c], h = b5[C + c], a1 = b5[2 * C + c], a2 = b5[3 * C + c], a3 = b5[4 * C + c];
    float xv[V], gv[V], ov[V];
    if constexpr (V == 4) {
      ld4<NT>(xv, x + i);
      ld4<NT>(gv, g + i);
    } else {
      xv[0] = x[i];
      gv[0] = g[i];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float gm = (relu && !(xv[j] * s + h > 0.f)) ? 0.f : gv[j];
      ov[j] = a1 * gm + a2 + a3 * xv[j];
    }
    if constexpr (V == 4) st4<NT>(out + i, ov);
    else out[i] = ov[0];
  }
}

// ------------------------------------------------------------------ pools
// adaptive average pool to 1: out[row] = mean_p v[row][p], one wave per row
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ v, float* __restrict__ out,
                                                         int rows, int P) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int p = threadIdx.x & 63; p < P; p += 64) s += v[(size_t)row * P + p];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) out[row] = s / (float)P;
}
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dv,
                                                         int P, size_t total) {
  const float inv = 1.f / (float)P;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    dv[i] = dout[i / P] * inv;
}

// relu(bn(x)) -> MaxPool2d(3, stride 2, pad 1); idx = winning tap 0..8 (first max wins, like torch)
__global__ __launch_bounds__(256) void bnrelu_maxpool_fwd_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ ss,
                                                                float* __restrict__ out,
                                                                unsigned char* __restrict__ idx, int C, int H,
                                                                int W, int Ho, int Wo, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int wo = (int)(i % Wo);
    size_t r = i / Wo;
    const int ho = (int)(r % Ho);
    r /= Ho;  // r = b*C + c
    const int c = (int)(r % C);
    const float s = ss[c], h = ss[C + c];
    float best = -INFINITY;
    int bi = 0;
    bool any = false;
    for (int dh = 0; dh < 3; ++dh) {
      const int hh = ho * 2 - 1 + dh;
      if (hh < 0 || hh >= H) continue;
      for (int dw = 0; dw < 3; ++dw) {
        const int ww = wo * 2 - 1 + dw;
        if (ww < 0 || ww >= W) continue;
        const float v = fmaxf(x[(r * H + hh) * W + ww] * s + h, 0.f);
        if (!any || v > best || v != v) {
          best = v;
          bi = dh * 3 + dw;
          any = true;
        }
      }
    }
    out[i] = best;
    idx[i] = (unsigned char)bi;
  }
}
// gather form (deterministic): dy[b,c,h,w] = sum over the <=4 windows that contain (h,w) and chose it
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dout,
                                                         const unsigned char* __restrict__ idx,
                                                         float* __restrict__ dy, int H, int W, int Ho, int Wo,
                                                         size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    size_t r = i / W;
    const int h = (int)(r % H);
    r /= H;
    float g = 0.f;
    for (int ho = (h + 1) / 2 - 1; ho <= (h + 1) / 2; ++ho) {
      if (ho < 0 || ho >= Ho) continue;
      const int dh = h - (ho * 2 - 1);
      if (dh < 0 || dh > 2) continue;
      for (int wo = (w + 1) / 2 - 1; wo <= (w + 1) / 2; ++wo) {
        if (wo < 0 || wo >= Wo) continue;
        const int dw = w - (wo * 2 - 1);
        if (dw < 0 || dw > 2) continue;
        const size_t o = (r * Ho + ho) * Wo + wo;
        if (idx[o] == dh * 3 + dw) g += dout[o];
      }
    }
    dy[i] = g;
  }
}

// ------------------------------------------------------------------ fused multi-tensor SGD
// torch.optim.SGD (momentum, weight decay, no nesterov/dampening): d = g + wd*p ; buf = mu*buf + d
// (buf = d on the first step) ; p -= lr*buf.     Up to SGD_CHUNK tensors per launch.
constexpr int SGD_CHUNK = 48;
struct SgdTable {
  float* p[SGD_CHUNK];
  const float* g[SGD_CHUNK];
  float* m[SGD_CHUNK];
  long long n[SGD_CHUNK];
  int blk0[SGD_CHUNK + 1];  // first block of each tensor
  int count;
};
__global__ __launch_bounds__(256) void sgd_kernel(const SgdTable t, float lr, float mu, float wd, int first) {
  // find the tensor this block works on (count <= 48: linear scan by one lane is fine)
  int ti = 0;
  while (ti + 1 < t.count && (int)blockIdx.x >= t.blk0[ti + 1]) ++ti;
  const long long base = (long long)(blockIdx.x - t.blk0[ti]) * 4096;
  float* __restrict__ p = t.p[ti];
  const float* __restrict__ g = t.g[ti];
  float* __restrict__ m = t.m[ti];
  const long long n = t.n[ti];
#pragma unroll 4
  for (int j = 0; j < 16; ++j) {
    const long long i = base + j * 256 + threadIdx.x;
    if (i < n) {
      const float pv = p[i];
      const float d = g[i] + wd * pv;
      const float b = first ? d : mu * m[i] + d;
      m[i] = b;
      p[i] = pv - lr * b;
    }
  }
}

// out[i] = src[0][i] + src[1][i] + ... (fixed order): weight gradients of a batch that was convolved in slices
__global__ __launch_bounds__(256) void sum_slices_kernel(const float* __restrict__ src, float* __restrict__ out,
                                                         int slices, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    float v = src[i];
    for (int s = 1; s < slices; ++s) v += src[(size_t)s * n + i];
    out[i] = v;
  }
}

__global__ void fill_kernel(float* __restrict__ p, float v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

static inline bool aligned16(const void* p) { return ((size_t)p & 15) == 0; }   // null counts as aligned
static inline unsigned grid_for(size_t n, int cap = 4096) {
  size_t b = (n + 255) / 256;
  if (b > (size_t)cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace slv

using namespace slv;

extern "C" {

int slv_bn_partials_to_sums(const float* psum, const float* psq, int nblk, int C, double* sums,
                            slv_stream_t stream) {
  SLV_CHECK_ARG(psum && psq && sums && nblk > 0 && C > 0, "null pointer or empty shape");
  hipLaunchKernelGGL(bn_partials_to_sums_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, psum, psq, nblk, C,
                     sums);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, float* mean_invstd, float* scale_shift, int C,
                    slv_stream_t stream) {
  SLV_CHECK_ARG(sums && gamma && beta && mean_invstd && scale_shift && C > 0 && count > 0, "bad argument");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, sums, count,
                     gamma, beta, running_mean, running_var, momentum, eps, mean_invstd, scale_shift, C);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_bn_stats_finalize(const float* psum, const float* psq, int nblk, double count, const float* gamma,
                          const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                          float* mean_invstd, float* scale_shift, int C, slv_stream_t stream) {
  SLV_CHECK_ARG(psum && psq && nblk > 0 && gamma && beta && mean_invstd && scale_shift && C > 0 && count > 0,
                "bad argument");
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, psum, psq, nblk, count,
                     gamma, beta, running_mean, running_var, momentum, eps, mean_invstd, scale_shift, C);
  SLV_LAUNCH_CHECK();
  return 0;
}

// SyncBN forward (main.py:117-118): conv-epilogue partials -> fp64 sums -> all-reduce over RCCL -> finalize with the
// GLOBAL count, one host call, one stream (the exchange sits between two kernels of the caller's stream).
int slv_bn_sync_finalize(slv_comm_t comm, const float* psum, const float* psq, int nblk, double count_local,
                         const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                         float eps, float* mean_invstd, float* scale_shift, int C, double* sums_scratch,
                         slv_stream_t stream) {
  SLV_CHECK_ARG(comm && sums_scratch, "communicator / scratch");
  int rc = slv_bn_partials_to_sums(psum, psq, nblk, C, sums_scratch, stream);
  if (rc) return rc;
  rc = slv::comm_allreduce_sum_f64(comm, sums_scratch, (size_t)2 * C, (hipStream_t)stream);
  if (rc) return rc;
  return slv_bn_finalize(sums_scratch, count_local * slv::comm_world(comm), gamma, beta, running_mean, running_var,
                         momentum, eps, mean_invstd, scale_shift, C, stream);
}

// SyncBN backward: slice partials -> fp64 sums -> all-reduce -> folded coefficients, likewise
int slv_bn_bwd_sync_finalize(slv_comm_t comm, const float* partial, int nsplit, double count_local, const float* gamma,
                             const float* mean_invstd, const float* scale_shift, float* bwd5, float* dgamma,
                             float* dbeta, int accumulate, int C, double* sums_scratch, slv_stream_t stream) {
  SLV_CHECK_ARG(comm && sums_scratch, "communicator / scratch");
  int rc = slv_bn_bwd_sums(partial, nsplit, C, sums_scratch, stream);
  if (rc) return rc;
  rc = slv::comm_allreduce_sum_f64(comm, sums_scratch, (size_t)2 * C, (hipStream_t)stream);
  if (rc) return rc;
  return slv_bn_bwd_finalize(sums_scratch, count_local * slv::comm_world(comm), gamma, mean_invstd, scale_shift, bwd5,
                             dgamma, dbeta, accumulate, C, stream);
}

int slv_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, float* mean_invstd, float* scale_shift, int C, slv_stream_t stream) {
  SLV_CHECK_ARG(gamma && beta && running_mean && running_var && scale_shift && C > 0, "bad argument");
  hipLaunchKernelGGL(bn_eval_params_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, mean_invstd, scale_shift, C);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_bn_act(const float* x, const float* scale_shift, const float* res, const float* res_scale_shift, int relu,
               float* out, int Bn, int C, int64_t P, slv_stream_t stream) {
  SLV_CHECK_ARG(x && scale_shift && out && Bn > 0 && C > 0 && P > 0, "bad argument");
  const size_t total = (size_t)Bn * C * P;
  SLV_CHECK_ARG(total < (1ull << 32), "more than 2^32 elements");
  const FastDiv dP = make_fastdiv((unsigned)P), dC = make_fastdiv((unsigned)C);
  if (P % 4 == 0 && aligned16(x) && aligned16(out) && aligned16(res))
    if (total * 4 >= NT_MIN_BYTES)
      hipLaunchKernelGGL((bn_act_kernel<4, true>), dim3(grid_for(total / 4, 8192)), dim3(256), 0, (hipStream_t)stream, x,
                         scale_shift, res, res_scale_shift, relu, out, C, dP, dC, (unsigned)(total / 4));
    else
      hipLaunchKernelGGL((bn_act_kernel<4, false>), dim3(grid_for(total / 4, 8192)), dim3(256), 0, (hipStream_t)stream, x,
                         scale_shift, res, res_scale_shift, relu, out, C, dP, dC, (unsigned)(total / 4));
  else
    hipLaunchKernelGGL((bn_act_kernel<1, false>), dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream, x,
                       scale_shift, res, res_scale_shift, relu, out, C, dP, dC, (unsigned)total);
  SLV_LAUNCH_CHECK();
  return 0;
}

int32_t slv_bn_bwd_nsplit(int Bn, int C, int64_t P) {
  const long long tot = (long long)Bn * P;
  long long s = (2048 + C - 1) / C;          // ~2048 workgroups in total
  const long long maxs = (tot + 1023) / 1024;  // >= 1024 elements per slice
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  return (int32_t)s;
}

int slv_bn_bwd_reduce(const float* g, const float* x, const float* mean_invstd, const float* scale_shift_mask,
                      const float* v_mask, const float* x2, const float* mean_invstd2, float* g_out, float* partial,
                      float* partial2, int Bn, int C, int64_t P, int nsplit, slv_stream_t stream) {
  SLV_CHECK_ARG(g && x && mean_invstd && partial && Bn > 0 && C > 0 && P > 0 && nsplit > 0, "bad argument");
  SLV_CHECK_ARG(!(scale_shift_mask && v_mask), "choose one mask source");
  SLV_CHECK_ARG(!v_mask || g_out, "masked gradient output required with v_mask");
  SLV_CHECK_ARG(!x2 || (mean_invstd2 && partial2), "second BN needs its stats and partial buffer");
  dim3 grid(C, nsplit);
  hipStream_t st = (hipStream_t)stream;
  const unsigned long long tot64 = (unsigned long long)Bn * (unsigned long long)P;
  SLV_CHECK_ARG(tot64 < (1ull << 31), "more than 2^31 elements per channel");
  const unsigned tot = (unsigned)tot64;
  const bool vec = (P % 4 == 0) && aligned16(g) && aligned16(x) && aligned16(v_mask) && aligned16(x2) && aligned16(g_out);
  unsigned per = (tot + nsplit - 1) / nsplit;
  per = (per + 3u) & ~3u;                       // slices start on 4-element boundaries
  const FastDiv dP = make_fastdiv((unsigned)P);
  const bool big = (size_t)tot * C * 4 >= NT_MIN_BYTES;
#define SLV_RED2(MASK, TWO, V, NT_)                                                                          \
  hipLaunchKernelGGL((bn_bwd_reduce_kernel<MASK, TWO, V, NT_>), grid, dim3(256), 0, st, g, x, mean_invstd,    \
                     scale_shift_mask, v_mask, x2, mean_invstd2, g_out, partial, partial2, C, (unsigned)P, dP, tot, \
                     per, nsplit)
#define SLV_RED(MASK, TWO)                                                                     \
  do {                                                                                         \
    if (vec && big) SLV_RED2(MASK, TWO, 4, true);                                              \
    else if (vec) SLV_RED2(MASK, TWO, 4, false);                                               \
    else SLV_RED2(MASK, TWO, 1, false);                                                        \
  } while (0)
  if (v_mask) {
    if (x2) SLV_RED(2, true); else SLV_RED(2, false);
  } else if (scale_shift_mask) {
    if (x2) SLV_RED(1, true); else SLV_RED(1, false);
  } else {
    if (x2) SLV_RED(0, true); else SLV_RED(0, false);
  }
#undef SLV_RED2
#undef SLV_RED
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_bn_bwd_sums(const float* partial, int nsplit, int C, double* sums, slv_stream_t stream) {
  SLV_CHECK_ARG(partial && sums && nsplit > 0 && C > 0, "bad argument");
  hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, nsplit, C, sums);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_bn_bwd_finalize(const double* sums, double count, const float* gamma, const float* mean_invstd,
                        const float* scale_shift, float* bwd5, float* dgamma, float* dbeta, int accumulate, int C,
                        slv_stream_t stream) {
  SLV_CHECK_ARG(sums && gamma && mean_invstd && bwd5 && C > 0 && count > 0, "bad argument");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, sums, count,
                     gamma, mean_invstd, scale_shift, bwd5, dgamma, dbeta, accumulate, C);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_bn_bwd_sums_finalize(const float* partial, int nsplit, double count, const float* gamma,
                             const float* mean_invstd, const float* scale_shift, float* bwd5, float* dgamma,
                             float* dbeta, int accumulate, int C, slv_stream_t stream) {
  SLV_CHECK_ARG(partial && nsplit > 0 && gamma && mean_invstd && bwd5 && C > 0 && count > 0, "bad argument");
  hipLaunchKernelGGL(bn_bwd_sums_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, nsplit, count,
                     gamma, mean_invstd, scale_shift, bwd5, dgamma, dbeta, accumulate, C);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_bn_bwd_apply(const float* g, const float* x, const float* bwd5, int relu, float* out, int Bn, int C,
                     int64_t P, slv_stream_t stream) {
  SLV_CHECK_ARG(g && x && bwd5 && out && Bn > 0 && C > 0 && P > 0, "bad argument");
  const size_t total = (size_t)Bn * C * P;
  SLV_CHECK_ARG(total < (1ull << 32), "more than 2^32 elements");
  const FastDiv dP = make_fastdiv((unsigned)P), dC = make_fastdiv((unsigned)C);
  if (P % 4 == 0 && aligned16(g) && aligned16(x) && aligned16(out))
    if (total * 4 >= NT_MIN_BYTES)
      hipLaunchKernelGGL((bn_bwd_apply_kernel<4, true>), dim3(grid_for(total / 4, 8192)), dim3(256), 0,
                         (hipStream_t)stream, g, x, bwd5, relu, out, C, dP, dC, (unsigned)(total / 4));
    else
      hipLaunchKernelGGL((bn_bwd_apply_kernel<4, false>), dim3(grid_for(total / 4, 8192)), dim3(256), 0,
                         (hipStream_t)stream, g, x, bwd5, relu, out, C, dP, dC, (unsigned)(total / 4));
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<1, false>), dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream,
                       g, x, bwd5, relu, out, C, dP, dC, (unsigned)total);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_avgpool_fwd(const float* v, float* out, int rows, int P, slv_stream_t stream) {
  SLV_CHECK_ARG(v && out && rows > 0 && P > 0, "bad argument");
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, v, out, rows, P);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_avgpool_bwd(const float* dout, float* dv, int rows, int P, slv_stream_t stream) {
  SLV_CHECK_ARG(dout && dv && rows > 0 && P > 0, "bad argument");
  const size_t total = (size_t)rows * P;
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dout, dv, P,
                     total);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_bnrelu_maxpool_fwd(const float* x, const float* scale_shift, float* out, uint8_t* idx, int Bn, int C,
                           int H, int W, slv_stream_t stream) {
  SLV_CHECK_ARG(x && scale_shift && out && idx && Bn > 0 && C > 0 && H > 0 && W > 0, "bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)Bn * C * Ho * Wo;
  hipLaunchKernelGGL(bnrelu_maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x,
                     scale_shift, out, idx, C, H, W, Ho, Wo, total);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_maxpool_bwd(const float* dout, const uint8_t* idx, float* dy, int Bn, int C, int H, int W,
                    slv_stream_t stream) {
  SLV_CHECK_ARG(dout && idx && dy && Bn > 0 && C > 0 && H > 0 && W > 0, "bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)Bn * C * H * W;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dout, idx, dy, H,
                     W, Ho, Wo, total);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_sgd_step(const void* const* params, const void* const* grads, const void* const* bufs,
                 const int64_t* sizes, int n_tensors, float lr, float momentum, float weight_decay, int first_step,
                 slv_stream_t stream) {
  SLV_CHECK_ARG(params && grads && bufs && sizes && n_tensors >= 0, "null pointer (host arrays expected)");
  int i = 0;
  while (i < n_tensors) {
    SgdTable t;
    t.count = 0;
    int blocks = 0;
    while (i < n_tensors && t.count < SGD_CHUNK) {
      if (sizes[i] > 0) {
        const int k = t.count++;
        t.p[k] = (float*)params[i];
        t.g[k] = (const float*)grads[i];
        t.m[k] = (float*)bufs[i];
        t.n[k] = sizes[i];
        t.blk0[k] = blocks;
        blocks += (int)((sizes[i] + 4095) / 4096);
      }
      ++i;
    }
    t.blk0[t.count] = blocks;
    if (t.count == 0) break;
    hipLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, lr, momentum, weight_decay,
                       first_step);
    SLV_LAUNCH_CHECK();
  }
  return 0;
}

int slv_sum_slices(const float* src, float* out, int slices, int64_t n, slv_stream_t stream) {
  SLV_CHECK_ARG(src && out && slices > 0 && n >= 0, "bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(sum_slices_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, src, out, slices,
                     (size_t)n);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_fill_f32(float* p, float value, int64_t n, slv_stream_t stream) {
  SLV_CHECK_ARG(p && n >= 0, "bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, p, value, (size_t)n);
  SLV_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

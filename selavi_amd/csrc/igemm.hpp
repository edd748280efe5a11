// Implicit-GEMM convolution / GEMM core for gfx950 (MI355X), fp32 in / fp32 accumulate on the
// matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, 157 TF peak = fp32 vector peak).
//
// One kernel template covers the four GEMM-shaped ops of the SeLaVi training step
// (reference: torchvision Conv3d/Conv2d + autograd called from model.py:145-166, main.py:284-301):
//   MODE_FWD    Y[co, p]      = sum_{ci,tap} W[co, ci, tap]  * act(X)[ci, p*stride + tap - pad]
//   MODE_DGRAD  dX[ci, q]     = sum_{co,tap} Wt[ci, co, tap] * dXout[co, (q + pad - tap)/stride]
//   MODE_WGRAD  dW[co, ci,tap] = sum_p dXout[co, p] * act(X)[ci, p*stride + tap - pad]
//   MODE_GEMM   C[m, n]       = sum_k A[m, k] * B[n, k]          (dense "NT" GEMM for the heads)
// with the BatchNorm that surrounds every conv of the model fused into the operand loaders:
//   PRO_ACT  v = relu?(x*scale[c] + shift[c])                (consumer-side BN apply + ReLU)
//   PRO_BWD  v = A1[c]*mask*g + A2[c] + A3[c]*x              (BN backward of the conv's own output)
// so activated tensors and BN input gradients are never materialised in HBM, and the forward
// epilogue emits per-channel sum / sum-of-squares partials (training-mode batch statistics).
//
// Tiling (wave64, 256 threads = 4 waves): block tile (MT*16) x (NT*64) x 16; each wave owns all
// MT*16 rows and NT*16 of the columns, i.e. MT x NT accumulators of 16x16 (4 VGPRs each).  The M
// tile is chosen per layer from {64,128,144,240} so the odd channel counts of R(2+1)D-18
// (45/144/230/460/921) waste < 5 % of the MFMA work.  LDS tiles are [rows][16+2] for K-contiguous
// operands and [16][BN+16] for position-contiguous operands; both give conflict-free ds_read_b32
// fragment reads (bank = (a/4) % 32 inside each 32-lane half).  Global->LDS is register staged and
// double buffered: loads for chunk c+1 are issued before the MFMAs of chunk c, one barrier per chunk.
#pragma once
#include "common.hpp"

namespace slv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2, MODE_GEMM = 3 };
enum { PRO_NONE = 0, PRO_ACT = 1, PRO_BWD = 2 };

struct IgemmArgs {
  const float* A;    // FWD/DGRAD/GEMM: dense [M][Kd] ; WGRAD: gradient tensor on the conv-output side
  const float* A2;   // WGRAD + PRO_BWD: raw conv output x (same shape as A)
  const float* B;    // FWD/WGRAD: conv input tensor ; DGRAD: gradient on the conv-output side ; GEMM: dense [N][Kd]
  const float* B2;   // DGRAD + PRO_BWD: raw conv output x
  const float* pa;   // per-channel prologue params of A (WGRAD): [5][Cout] = s, h, A1, A2, A3
  const float* pb;   // per-channel prologue params of B: PRO_ACT [2][C] = s, h ; PRO_BWD [5][C]
  const int2* tab;   // FWD/WGRAD: (ci,tap) table ; DGRAD: (co,tap) table  -> {offset, dt|dh<<4|dw<<8|chan<<12}
  float* C;          // output
  const float* E;    // optional epilogue addend, same shape as C (DGRAD residual / accumulate)
  const float* bias; // GEMM: optional per-column bias
  float* stat_sum;   // FWD: [M][nblkN] per-channel partial sums of the output (or null)
  float* stat_sq;
  int M, Kd;
  long long Ntot;    // columns: positions (FWD/DGRAD) or Cin*taps (WGRAD) or N (GEMM)
  int nblkM, nblkN;
  int Bn, Cin, Ti, Hi, Wi, Cout, To, Ho, Wo;
  int st, sh, sw, pt, ph, pw;
  int a_pro, b_pro, a_relu, b_relu;
  int chunks_per_split;  // WGRAD: 16-position chunks per grid.y slice
  long long Ptot;        // WGRAD: Bn*To*Ho*Wo
  int ldc;               // WGRAD/GEMM: leading dimension of C
};

__device__ __forceinline__ float apply_act(float x, float s, float h, int relu) {
  float v = x * s + h;
  return relu ? fmaxf(v, 0.f) : v;
}
__device__ __forceinline__ float apply_bwd(float g, float x, float s, float h, float a1, float a2,
                                           float a3, int relu) {
  if (relu) g = (x * s + h > 0.f) ? g : 0.f;
  return a1 * g + a2 + a3 * x;
}

template <int MODE, int MT, int NT>
__global__ __launch_bounds__(256, (MT >= 15 ? 2 : 3)) void igemm_kernel(const IgemmArgs g) {
  constexpr int BM = MT * 16, BN = NT * 64;
  constexpr int AS = 18;
  constexpr bool BKF = (MODE == MODE_WGRAD || MODE == MODE_GEMM);  // B tile K-contiguous?
  constexpr int BS = BKF ? 18 : (BN + 16);
  constexpr int A_ELEMS = BM * AS;
  constexpr int B_ELEMS = BKF ? BN * 18 : 16 * BS;
  constexpr int BROWS = BN / 16;  // B staging registers per thread (both layouts)
  __shared__ float smem[2 * (A_ELEMS + B_ELEMS)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- XCD-aware block remap (bijective): the nblkM blocks of one column tile share an L2
  int mblk, nblk;
  {
    const int nb = gridDim.x, id = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = id & 7, loc = id >> 3;
    const int nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    mblk = nid % g.nblkM;
    nblk = nid / g.nblkM;
  }
  const int m0 = mblk * BM;
  const long long n0 = (long long)nblk * BN;
  const int mrem = g.M - m0;  // valid rows in this block (may exceed BM)

  const int HWi = g.Hi * g.Wi, THWi = g.Ti * HWi;
  const int HoWo = g.Ho * g.Wo, Pout = g.To * HoWo;

  // ---- per-thread constants of the loaders
  const int a_kk = tid & 15, a_r = tid >> 4;  // K-contiguous loader: 16 k's x 16 rows per pass

  // position-contiguous B loader (FWD/DGRAD): one column per thread, BROWS k's
  int nl = 0, kg = 0;
  bool nvalid = false;
  int pb_, t0 = 0, h0 = 0, w0 = 0;
  long long bbase = 0;
  (void)pb_;
  if constexpr (MODE == MODE_FWD || MODE == MODE_DGRAD) {
    nl = tid % BN;
    kg = __builtin_amdgcn_readfirstlane(tid / BN);
    const long long n = n0 + nl;
    nvalid = n < g.Ntot;
    const long long nn = nvalid ? n : 0;
    if constexpr (MODE == MODE_FWD) {
      const int b = (int)(nn / Pout);
      int rem = (int)(nn - (long long)b * Pout);
      const int to = rem / HoWo;
      rem -= to * HoWo;
      const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
      t0 = to * g.st - g.pt;
      h0 = ho * g.sh - g.ph;
      w0 = wo * g.sw - g.pw;
      bbase = (long long)b * g.Cin * THWi + (long long)t0 * HWi + h0 * g.Wi + w0;
    } else {
      const int b = (int)(nn / THWi);
      int rem = (int)(nn - (long long)b * THWi);
      const int ti = rem / HWi;
      rem -= ti * HWi;
      const int hi = rem / g.Wi, wi = rem - hi * g.Wi;
      t0 = ti + g.pt;
      h0 = hi + g.ph;
      w0 = wi + g.pw;
      bbase = (long long)b * g.Cout * Pout;
    }
  }
  // WGRAD position walker: this thread's reduction position p = pcur + a_kk
  int wb = 0, wto = 0, who = 0, wwo = 0;
  long long wp = 0, wp_end = 0;
  int nchunks;
  if constexpr (MODE == MODE_WGRAD) {
    const long long c0 = (long long)blockIdx.y * g.chunks_per_split;
    const long long call = (g.Ptot + 15) / 16;
    long long c1 = c0 + g.chunks_per_split;
    if (c1 > call) c1 = call;
    nchunks = (int)(c1 > c0 ? c1 - c0 : 0);
    wp = c0 * 16 + a_kk;
    wp_end = g.Ptot;
    const long long pp = wp < wp_end ? wp : 0;
    wb = (int)(pp / Pout);
    int rem = (int)(pp - (long long)wb * Pout);
    wto = rem / HoWo;
    rem -= wto * HoWo;
    who = rem / g.Wo;
    wwo = rem - who * g.Wo;
  } else {
    nchunks = (g.Kd + 15) / 16;
  }

  float ra[MT];
  float rb[BROWS];

  // ---------------------------------------------------------------- global -> registers
  auto load_chunk = [&](int c) {
    const int k0 = c * 16;
    if constexpr (MODE != MODE_WGRAD) {
      // A: dense [M][Kd]
      const int k = k0 + a_kk;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = a_r + 16 * i;
        ra[i] = (m < mrem && k < g.Kd) ? g.A[(size_t)(m0 + m) * g.Kd + k] : 0.f;
      }
    }
    if constexpr (MODE == MODE_GEMM) {
      const int k = k0 + a_kk;
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        const long long n = n0 + a_r + 16 * i;
        rb[i] = (n < g.Ntot && k < g.Kd) ? g.B[(size_t)n * g.Kd + k] : 0.f;
      }
    }
    if constexpr (MODE == MODE_FWD) {
#pragma unroll
      for (int q = 0; q < BROWS; ++q) {
        const int k = k0 + kg * BROWS + q;  // wave-uniform
        float v = 0.f;
        if (k < g.Kd) {
          const int2 e = g.tab[k];
          const int dt = e.y & 15, dh = (e.y >> 4) & 15, dw = (e.y >> 8) & 15, ch = e.y >> 12;
          const bool ok = nvalid && (unsigned)(t0 + dt) < (unsigned)g.Ti &&
                          (unsigned)(h0 + dh) < (unsigned)g.Hi && (unsigned)(w0 + dw) < (unsigned)g.Wi;
          if (ok) {
            v = g.B[bbase + e.x];
            if (g.b_pro == PRO_ACT) v = apply_act(v, g.pb[ch], g.pb[g.Cin + ch], g.b_relu);
          }
        }
        rb[q] = v;
      }
    }
    if constexpr (MODE == MODE_DGRAD) {
      const int mt_ = g.st - 1, mh_ = g.sh - 1, mw_ = g.sw - 1;  // strides are 1 or 2
      const int lt_ = g.st >> 1, lh_ = g.sh >> 1, lw_ = g.sw >> 1;
#pragma unroll
      for (int q = 0; q < BROWS; ++q) {
        const int k = k0 + kg * BROWS + q;
        float v = 0.f;
        if (k < g.Kd) {
          const int2 e = g.tab[k];
          const int dt = e.y & 15, dh = (e.y >> 4) & 15, dw = (e.y >> 8) & 15, ch = e.y >> 12;
          const int tt = t0 - dt, hh = h0 - dh, ww = w0 - dw;
          const int to = tt >> lt_, ho = hh >> lh_, wo = ww >> lw_;
          const bool ok = nvalid && tt >= 0 && hh >= 0 && ww >= 0 && ((tt & mt_) | (hh & mh_) | (ww & mw_)) == 0 &&
                          to < g.To && ho < g.Ho && wo < g.Wo;
          if (ok) {
            const long long ad = bbase + e.x + (long long)to * HoWo + ho * g.Wo + wo;
            v = g.B[ad];
            if (g.b_pro == PRO_BWD) {
              const int C_ = g.Cout;
              v = apply_bwd(v, g.B2[ad], g.pb[ch], g.pb[C_ + ch], g.pb[2 * C_ + ch], g.pb[3 * C_ + ch],
                            g.pb[4 * C_ + ch], g.b_relu);
            }
          }
        }
        rb[q] = v;
      }
    }
    if constexpr (MODE == MODE_WGRAD) {
      const bool pok = wp < wp_end;
      const int rem = wto * HoWo + who * g.Wo + wwo;
      // A[m][p] = dXout[b][m][p]
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = a_r + 16 * i;
        float v = 0.f;
        if (pok && m < mrem) {
          const int ch = m0 + m;
          const size_t ad = ((size_t)wb * g.Cout + ch) * Pout + rem;
          v = g.A[ad];
          if (g.a_pro == PRO_BWD) {
            const int C_ = g.Cout;
            v = apply_bwd(v, g.A2[ad], g.pa[ch], g.pa[C_ + ch], g.pa[2 * C_ + ch], g.pa[3 * C_ + ch],
                          g.pa[4 * C_ + ch], g.a_relu);
          }
        }
        ra[i] = v;
      }
      // B[n][p] = act(X)[b][ci][in_pos(p, tap)]
      const int ti0 = wto * g.st - g.pt, hi0 = who * g.sh - g.ph, wi0 = wwo * g.sw - g.pw;
      const long long xb = (long long)wb * g.Cin * THWi + (long long)ti0 * HWi + hi0 * g.Wi + wi0;
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        const long long n = n0 + a_r + 16 * i;
        float v = 0.f;
        if (pok && n < g.Ntot) {
          const int2 e = g.tab[n];
          const int dt = e.y & 15, dh = (e.y >> 4) & 15, dw = (e.y >> 8) & 15, ch = e.y >> 12;
          const bool ok = (unsigned)(ti0 + dt) < (unsigned)g.Ti && (unsigned)(hi0 + dh) < (unsigned)g.Hi &&
                          (unsigned)(wi0 + dw) < (unsigned)g.Wi;
          if (ok) {
            v = g.B[xb + e.x];
            if (g.b_pro == PRO_ACT) v = apply_act(v, g.pb[ch], g.pb[g.Cin + ch], g.b_relu);
          }
        }
        rb[i] = v;
      }
      // advance the walker by one chunk (16 positions)
      wp += 16;
      wwo += 16;
      while (wwo >= g.Wo) { wwo -= g.Wo; ++who; }
      while (who >= g.Ho) { who -= g.Ho; ++wto; }
      while (wto >= g.To) { wto -= g.To; ++wb; }
    }
  };

  // ---------------------------------------------------------------- registers -> LDS
  auto store_chunk = [&](int buf) {
    float* As = smem + buf * (A_ELEMS + B_ELEMS);
    float* Bs = As + A_ELEMS;
#pragma unroll
    for (int i = 0; i < MT; ++i) As[(a_r + 16 * i) * AS + a_kk] = ra[i];
    if constexpr (BKF) {
#pragma unroll
      for (int i = 0; i < BROWS; ++i) Bs[(a_r + 16 * i) * 18 + a_kk] = rb[i];
    } else {
#pragma unroll
      for (int q = 0; q < BROWS; ++q) Bs[(kg * BROWS + q) * BS + nl] = rb[q];
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fi = lane & 15, fk = lane >> 4;

  auto compute = [&](int buf) {
    const float* As = smem + buf * (A_ELEMS + B_ELEMS);
    const float* Bs = As + A_ELEMS;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = As[(i * 16 + fi) * AS + kk * 4 + fk];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if constexpr (BKF) b[j] = Bs[((wave * NT + j) * 16 + fi) * 18 + kk * 4 + fk];
        else b[j] = Bs[(kk * 4 + fk) * BS + (wave * NT + j) * 16 + fi];
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (i * 16 < mrem) {  // wave-uniform: skip row tiles beyond M
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  };

  // ---------------------------------------------------------------- main loop
  if (nchunks > 0) {
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      const bool more = c + 1 < nchunks;
      if (more) load_chunk(c + 1);
      compute(c & 1);
      if (more) store_chunk((c + 1) & 1);
      __syncthreads();
    }
  }

  // ---------------------------------------------------------------- epilogue
  // accumulator layout: row = i*16 + fk*4 + r, col = (wave*NT + j)*16 + fi
  if constexpr (MODE == MODE_FWD || MODE == MODE_DGRAD) {
    const int Pc = (MODE == MODE_FWD) ? Pout : THWi;  // positions per sample on the output side
    size_t obase[NT];
    bool cok[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const long long n = n0 + (wave * NT + j) * 16 + fi;
      cok[j] = n < g.Ntot;
      const long long nn = cok[j] ? n : 0;
      const long long b = nn / Pc;
      obase[j] = (size_t)b * g.M * Pc + (size_t)(nn - b * Pc);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (i * 16 < mrem) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = i * 16 + fk * 4 + r;
          if (m < mrem) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              if (cok[j]) {
                const size_t ad = obase[j] + (size_t)(m0 + m) * Pc;
                float v = acc[i][j][r];
                if (g.E) v += g.E[ad];
                g.C[ad] = v;
              }
            }
          }
        }
      }
    }
    if constexpr (MODE == MODE_FWD) {
      if (g.stat_sum) {
        // per-channel partial statistics of this block's columns (fixed order -> deterministic)
        float* red = smem;  // [2][4 waves][BM]   (main loop ended with a barrier)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              const float v = cok[j] ? acc[i][j][r] : 0.f;
              s += v;
              q += v * v;
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
              s += __shfl_xor(s, o, 64);
              q += __shfl_xor(q, o, 64);
            }
            if (fi == 0) {
              const int m = i * 16 + fk * 4 + r;
              red[wave * BM + m] = s;
              red[(4 + wave) * BM + m] = q;
            }
          }
        }
        __syncthreads();
        for (int m = tid; m < BM; m += 256) {
          if (m < mrem) {
            const float s = ((red[m] + red[BM + m]) + red[2 * BM + m]) + red[3 * BM + m];
            const float q = ((red[4 * BM + m] + red[5 * BM + m]) + red[6 * BM + m]) + red[7 * BM + m];
            g.stat_sum[(size_t)(m0 + m) * g.nblkN + nblk] = s;
            g.stat_sq[(size_t)(m0 + m) * g.nblkN + nblk] = q;
          }
        }
      }
    }
  } else {
    // WGRAD / GEMM: C[(split)][m][n], row-major with leading dimension ldc
    float* Cout_ = g.C + (MODE == MODE_WGRAD ? (size_t)blockIdx.y * g.M * g.ldc : 0);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (i * 16 < mrem) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = i * 16 + fk * 4 + r;
          if (m < mrem) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              const long long n = n0 + (wave * NT + j) * 16 + fi;
              if (n < g.Ntot) {
                float v = acc[i][j][r];
                if constexpr (MODE == MODE_GEMM) {
                  if (g.bias) v += g.bias[n];
                }
                Cout_[(size_t)(m0 + m) * g.ldc + n] = v;
              }
            }
          }
        }
      }
    }
  }
}

template <int MODE, int MT, int NT>
inline void launch_igemm(const IgemmArgs& a, int splits, hipStream_t st) {
  dim3 grid(a.nblkM * a.nblkN, splits, 1);
  hipLaunchKernelGGL((igemm_kernel<MODE, MT, NT>), grid, dim3(256), 0, st, a);
}

// choose the row-tile: returns MT for a given M (rows) -- see header comment
inline int pick_mt(int M) {
  if (M <= 64) return 4;
  if (M <= 128) return 8;
  if (M <= 144) return 9;
  if (M <= 240) return 15;
  // multi-block: minimise padded rows, prefer the larger tile on ties
  const int cand[4] = {15, 9, 8, 4};
  int best = 8;
  long bestw = 1L << 60;
  for (int c : cand) {
    const int bm = c * 16;
    const long padded = (long)((M + bm - 1) / bm) * bm;
    if (padded < bestw) { bestw = padded; best = c; }
  }
  return best;
}

}  // namespace slv

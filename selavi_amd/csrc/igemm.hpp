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
// fragment reads (bank = (a/4) % 32 inside each 32-lane half).
//
// Pipeline: global->LDS is register staged and double buffered.  Loads of chunk c+1 are issued
// (unconditionally, from clamped addresses -- no divergent branches) before the MFMAs of chunk c;
// validity masking and the BN prologue math run when the registers are written to LDS after those
// MFMAs, so the global latency hides under the matrix pipe.  One barrier per chunk.
#pragma once
#include <type_traits>

#include "common.hpp"

namespace slv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2, MODE_GEMM = 3 };
enum { PRO_NONE = 0, PRO_ACT = 1, PRO_BWD = 2 };

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

struct IgemmArgs {
  const float* A;    // FWD/DGRAD/GEMM: dense [M][Kd] ; WGRAD: gradient tensor on the conv-output side
  const float* A2;   // WGRAD + PRO_BWD: raw conv output x (same shape as A)
  const float* B;    // FWD/WGRAD: conv input tensor ; DGRAD: gradient on the conv-output side ; GEMM: dense [N][Kd]
  const float* B2;   // DGRAD + PRO_BWD: raw conv output x
  const float* pa;   // per-channel prologue params of A (WGRAD): [5][Cout] = s, h, A1, A2, A3
  const float* pb;   // per-channel prologue params of B: PRO_ACT [2][C] = s, h ; PRO_BWD [5][C]
  const int2* tab;   // (c,tap) table, padded with invalid entries (y < 0) to a multiple of 16 (+16):
                     //   {offset, dt | dh<<4 | dw<<8 | c<<12}
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
  int chunks_per_split;  // WGRAD: 16-position chunks per K-slice
  long long Ptot;        // WGRAD: Bn*To*Ho*Wo
  int ldc;               // WGRAD/GEMM: leading dimension of C
  FastDiv dPout, dHoWo, dWo;  // WGRAD position decode
};

__device__ __forceinline__ float apply_act(float x, float s, float h, int relu) {
  const float v = x * s + h;
  return relu ? fmaxf(v, 0.f) : v;
}
__device__ __forceinline__ float apply_bwd(float g, float x, float s, float h, float a1, float a2,
                                           float a3, int relu) {
  const float gm = (relu && !(x * s + h > 0.f)) ? 0.f : g;
  return a1 * gm + a2 + a3 * x;
}

template <int MODE, int MT, int NT>
__global__ __launch_bounds__(256, ((MT >= 15 || MODE == MODE_WGRAD || (MODE == MODE_DGRAD && NT == 2)) ? 2 : 3))
void igemm_kernel(const IgemmArgs g) {
  constexpr int BM = MT * 16, BN = NT * 64;
  constexpr int AS = 18;
  constexpr bool BKF = (MODE == MODE_WGRAD || MODE == MODE_GEMM);  // B tile K-contiguous?
  constexpr int BS = BKF ? 18 : (BN + 16);
  constexpr int A_ELEMS = BM * AS;
  constexpr int B_ELEMS = BKF ? BN * 18 : 16 * BS;
  constexpr int BROWS = BN / 16;  // B staging registers per thread (both layouts)
  constexpr int P_ELEMS = (MODE == MODE_WGRAD) ? (5 * BM + 2 * BN) : 0;
  __shared__ float smem[2 * (A_ELEMS + B_ELEMS) + P_ELEMS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- XCD-aware block remap (bijective): the tiles that share operands share an XCD's L2
  int mblk, nblk, split;
  {
    const int nb = gridDim.x, id = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = id & 7, loc = id >> 3;
    const int nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int per = g.nblkM * g.nblkN;
    split = nid / per;  // WGRAD: K-slice
    const int rem_ = nid - split * per;
    mblk = rem_ % g.nblkM;
    nblk = rem_ / g.nblkM;
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
  int t0 = 0, h0 = 0, w0 = 0;
  long long bbase = 0;
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
  // WGRAD: this thread's reduction position p = chunk*16 + a_kk
  unsigned wp = 0;
  int nchunks;
  if constexpr (MODE == MODE_WGRAD) {
    const long long c0 = (long long)split * g.chunks_per_split;
    const long long call = (g.Ptot + 15) / 16;
    long long c1 = c0 + g.chunks_per_split;
    if (c1 > call) c1 = call;
    nchunks = (int)(c1 > c0 ? c1 - c0 : 0);
    wp = (unsigned)(c0 * 16 + a_kk);
  } else {
    nchunks = (g.Kd + 15) / 16;
  }

  // staging registers hold RAW loaded values; masking + prologue math run in store_chunk
  float ra[MT], ra2[MT];
  float rb[BROWS], rb2[BROWS];
  unsigned okA = 0, okB = 0;  // per-element validity bits

  (void)ra2; (void)rb2; (void)okA; (void)okB;

  // WGRAD: loop-invariant table entries of this thread's B rows and per-channel params in LDS
  int2 wte[BROWS];
  (void)wte;
  float* pAs = smem + 2 * (A_ELEMS + B_ELEMS);  // [5][BM]
  float* pBs = pAs + 5 * BM;                    // [2][BN]
  if constexpr (MODE == MODE_WGRAD) {
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      const long long n = n0 + a_r + 16 * i;
      wte[i] = g.tab[n < g.Ntot ? n : g.Ntot];  // entry Ntot is an invalid (y < 0) pad entry
    }
    for (int i = tid; i < 5 * BM; i += 256) {
      const int which = i / BM, m = i - which * BM;
      pAs[i] = (g.a_pro == PRO_BWD && m < mrem) ? g.pa[which * g.Cout + m0 + m] : 0.f;
    }
    for (int i = tid; i < 2 * BN; i += 256) {
      const int which = i / BN, nn = i - which * BN;
      const long long n = n0 + nn;
      float v = 0.f;
      if (g.b_pro == PRO_ACT && n < g.Ntot) v = g.pb[which * g.Cin + ((g.tab[n].y >> 12) & 0x7FFFF)];
      pBs[i] = v;
    }
  }

  // ---------------------------------------------------------------- global -> registers (raw, branch-free)
  // NB: every lambda is force-inlined -- an out-of-line lambda captures the register arrays by
  // reference and drags the accumulators into scratch memory (and turns global loads into flat ones).
  auto load_chunk = [&](int c) __attribute__((always_inline)) {
    const int k0 = c * 16;
    if constexpr (MODE != MODE_WGRAD) {
      // A: dense [M][Kd]
      const int k = k0 + a_kk;
      const bool kok = k < g.Kd;
      okA = 0;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = a_r + 16 * i;
        const bool ok = kok && m < mrem;
        ra[i] = g.A[ok ? (size_t)(m0 + m) * g.Kd + k : 0];
        okA |= (ok ? 1u : 0u) << i;
      }
    }
    if constexpr (MODE == MODE_GEMM) {
      const int k = k0 + a_kk;
      okB = 0;
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        const long long n = n0 + a_r + 16 * i;
        const bool ok = n < g.Ntot && k < g.Kd;
        rb[i] = g.B[ok ? (size_t)n * g.Kd + k : 0];
        okB |= (ok ? 1u : 0u) << i;
      }
    }
    if constexpr (MODE == MODE_FWD) {
      okB = 0;
#pragma unroll
      for (int q = 0; q < BROWS; ++q) {
        const int k = k0 + kg * BROWS + q;  // wave-uniform; table is padded, no bound check needed
        const int2 e = g.tab[k];
        const int dt = e.y & 15, dh = (e.y >> 4) & 15, dw = (e.y >> 8) & 15, ch = (e.y >> 12) & 0x7FFFF;
        const bool ok = nvalid && e.y >= 0 && (unsigned)(t0 + dt) < (unsigned)g.Ti &&
                        (unsigned)(h0 + dh) < (unsigned)g.Hi && (unsigned)(w0 + dw) < (unsigned)g.Wi;
        rb[q] = g.B[ok ? bbase + e.x : 0];
        okB |= (ok ? 1u : 0u) << q;
        (void)ch;
      }
    }
    if constexpr (MODE == MODE_DGRAD) {
      const int mt_ = g.st - 1, mh_ = g.sh - 1, mw_ = g.sw - 1;  // strides are 1 or 2
      const int lt_ = g.st >> 1, lh_ = g.sh >> 1, lw_ = g.sw >> 1;
      okB = 0;
#pragma unroll
      for (int q = 0; q < BROWS; ++q) {
        const int k = k0 + kg * BROWS + q;
        const int2 e = g.tab[k];
        const int dt = e.y & 15, dh = (e.y >> 4) & 15, dw = (e.y >> 8) & 15, ch = (e.y >> 12) & 0x7FFFF;
        const int tt = t0 - dt, hh = h0 - dh, ww = w0 - dw;
        const int to = tt >> lt_, ho = hh >> lh_, wo = ww >> lw_;
        const bool ok = nvalid && e.y >= 0 && (tt | hh | ww) >= 0 && ((tt & mt_) | (hh & mh_) | (ww & mw_)) == 0 &&
                        to < g.To && ho < g.Ho && wo < g.Wo;
        const long long ad = ok ? bbase + e.x + (long long)to * HoWo + ho * g.Wo + wo : 0;
        rb[q] = g.B[ad];
        if (g.b_pro == PRO_BWD) rb2[q] = g.B2[ad];
        (void)ch;
        okB |= (ok ? 1u : 0u) << q;
      }
    }
    if constexpr (MODE == MODE_WGRAD) {
      const unsigned p = wp + (unsigned)c * 16u;
      const bool pok = (long long)p < g.Ptot;
      const unsigned pp = pok ? p : 0u;
      const unsigned b = fdiv(pp, g.dPout);
      const unsigned rem = pp - b * (unsigned)Pout;
      const unsigned to = fdiv(rem, g.dHoWo);
      const unsigned r2 = rem - to * (unsigned)HoWo;
      const unsigned ho = fdiv(r2, g.dWo);
      const unsigned wo = r2 - ho * (unsigned)g.Wo;
      okA = 0;
      okB = 0;
      // A[m][p] = dXout[b][m][p]
      const size_t abase = (size_t)b * g.Cout * Pout + rem;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = a_r + 16 * i;
        const bool ok = pok && m < mrem;
        const size_t ad = ok ? abase + (size_t)(m0 + m) * Pout : 0;
        ra[i] = g.A[ad];
        if (g.a_pro == PRO_BWD) ra2[i] = g.A2[ad];
        okA |= (ok ? 1u : 0u) << i;
      }
      // B[n][p] = act(X)[b][ci][in_pos(p, tap)]
      const int ti0 = (int)to * g.st - g.pt, hi0 = (int)ho * g.sh - g.ph, wi0 = (int)wo * g.sw - g.pw;
      const long long xb = (long long)b * g.Cin * THWi + (long long)ti0 * HWi + hi0 * g.Wi + wi0;
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        const int2 e = wte[i];
        const int dt = e.y & 15, dh = (e.y >> 4) & 15, dw = (e.y >> 8) & 15;
        const bool ok = pok && e.y >= 0 && (unsigned)(ti0 + dt) < (unsigned)g.Ti &&
                        (unsigned)(hi0 + dh) < (unsigned)g.Hi && (unsigned)(wi0 + dw) < (unsigned)g.Wi;
        rb[i] = g.B[ok ? xb + e.x : 0];
        okB |= (ok ? 1u : 0u) << i;
      }
    }
  };

  // ---------------------------------------------------------------- registers -> (mask, prologue) -> LDS
  auto store_chunk = [&](int buf, int c) __attribute__((always_inline)) {
    (void)c;
    float* As = smem + buf * (A_ELEMS + B_ELEMS);
    float* Bs = As + A_ELEMS;
    if constexpr (MODE == MODE_WGRAD) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = a_r + 16 * i;
        float v = ra[i];
        if (g.a_pro == PRO_BWD)
          v = apply_bwd(v, ra2[i], pAs[m], pAs[BM + m], pAs[2 * BM + m], pAs[3 * BM + m], pAs[4 * BM + m], g.a_relu);
        As[m * AS + a_kk] = ((okA >> i) & 1u) ? v : 0.f;
      }
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        const int nn = a_r + 16 * i;
        float v = rb[i];
        if (g.b_pro == PRO_ACT) v = apply_act(v, pBs[nn], pBs[BN + nn], g.b_relu);
        Bs[nn * 18 + a_kk] = ((okB >> i) & 1u) ? v : 0.f;
      }
    } else {
#pragma unroll
      for (int i = 0; i < MT; ++i) As[(a_r + 16 * i) * AS + a_kk] = ((okA >> i) & 1u) ? ra[i] : 0.f;
      if constexpr (MODE == MODE_GEMM) {
#pragma unroll
        for (int i = 0; i < BROWS; ++i) Bs[(a_r + 16 * i) * 18 + a_kk] = ((okB >> i) & 1u) ? rb[i] : 0.f;
      } else {
        // per-k channel params are wave-uniform: short-lived scalar loads here (keeping them live
        // across the MFMA phase costs 16-40 registers and spills)
#pragma unroll
        for (int q = 0; q < BROWS; ++q) {
          float v = rb[q];
          if (g.b_pro != PRO_NONE) {
            const int ch = (g.tab[c * 16 + kg * BROWS + q].y >> 12) & 0x7FFFF;
            if constexpr (MODE == MODE_FWD) {
              v = apply_act(v, g.pb[ch], g.pb[g.Cin + ch], g.b_relu);
            } else {
              const int C_ = g.Cout;
              v = apply_bwd(v, rb2[q], g.pb[ch], g.pb[C_ + ch], g.pb[2 * C_ + ch], g.pb[3 * C_ + ch],
                            g.pb[4 * C_ + ch], g.b_relu);
            }
          }
          Bs[(kg * BROWS + q) * BS + nl] = ((okB >> q) & 1u) ? v : 0.f;
        }
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fi = lane & 15, fk = lane >> 4;
  const int mtv = __builtin_amdgcn_readfirstlane((mrem + 15) >> 4);  // valid 16-row tiles (wave-uniform)

  auto compute = [&](int buf, auto full_tag) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
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
        if (FULL || i < mtv) {  // wave-uniform: skip row tiles beyond M in ragged blocks
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  };

  // ---------------------------------------------------------------- main loop
  auto main_loop = [&](auto full_tag) __attribute__((always_inline)) {
    load_chunk(0);
    store_chunk(0, 0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      const bool more = c + 1 < nchunks;
      if (more) load_chunk(c + 1);
      compute(c & 1, full_tag);
      if (more) store_chunk((c + 1) & 1, c + 1);
      __syncthreads();
    }
  };
  if (nchunks > 0) {
    if (mtv >= MT) main_loop(std::true_type{});
    else main_loop(std::false_type{});
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
      if (i < mtv) {
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
    float* Cout_ = g.C + (MODE == MODE_WGRAD ? (size_t)split * g.M * g.ldc : 0);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (i < mtv) {
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
  dim3 grid(a.nblkM * a.nblkN * splits, 1, 1);
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

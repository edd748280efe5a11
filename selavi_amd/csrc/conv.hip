// C-ABI entry points of the implicit-GEMM convolution family (see igemm.hpp for the kernel).
// Replaces cuDNN conv3d/conv2d forward, backward-data and backward-weight as reached from the
// torchvision nets instantiated by /root/reference/model.py:95,114 and their autograd backward
// (main.py:301).
#include <stdlib.h>

#include "igemm.hpp"
#include "../../include/selavi_hip.h"

namespace slv {

struct Geom {
  int Bn, Cin, Ti, Hi, Wi, Cout, To, Ho, Wo, kt, kh, kw, st, sh, sw, pt, ph, pw;
};

static int read_geom(const int32_t* p, Geom& g) {
  if (!p) return -1;
  memcpy(&g, p, sizeof(Geom));
  if (g.Bn <= 0 || g.Cin <= 0 || g.Cout <= 0 || g.Ti <= 0 || g.Hi <= 0 || g.Wi <= 0) return -1;
  if (g.kt <= 0 || g.kh <= 0 || g.kw <= 0 || g.kt * g.kh * g.kw > 63) return -1;
  if (g.kt > 15 || g.kh > 15 || g.kw > 15 || g.pt > 15 || g.ph > 15 || g.pw > 15) return -1;
  if ((g.st != 1 && g.st != 2) || (g.sh != 1 && g.sh != 2) || (g.sw != 1 && g.sw != 2)) return -1;
  const int To = (g.Ti + 2 * g.pt - g.kt) / g.st + 1, Ho = (g.Hi + 2 * g.ph - g.kh) / g.sh + 1,
            Wo = (g.Wi + 2 * g.pw - g.kw) / g.sw + 1;
  if (To != g.To || Ho != g.Ho || Wo != g.Wo || To <= 0 || Ho <= 0 || Wo <= 0) return -1;
  // 32-bit byte offsets inside the buffer descriptors
  if ((long long)g.Bn * g.Cin * g.Ti * g.Hi * g.Wi * 4 >= 0xFFFFFFF0LL) return -1;
  if ((long long)g.Bn * g.Cout * g.To * g.Ho * g.Wo * 4 >= 0xFFFFFFF0LL) return -1;
  if (g.Cin >= (1 << 22) || g.Cout >= (1 << 22)) return -1;
  return 0;
}

// One MODE_CONV launch: the forward conv, or one stride-parity class of the backward-data conv.
struct Desc {
  int M, C, Kd, ntaps;     // Kd: K extent of the launch (C*ntaps, or ntaps*Cp with tap-major K)
  int kord, Cp;            // K ordering (igemm.hpp) and padded channel count of the tap-major layout
  int taps[64];            // linear tap ids (kt,kh,kw order) in this launch's k order
  int delta[64][3];        // source-coordinate delta of each tap
  int Q[3], mul[3], S[3];  // lattice dims, source multipliers, source dims
  int dmul[3], dorg[3], D[3];
  long long Ntot;
  size_t tab_words;        // int32 words of this launch's table block
  size_t tab_off;          // word offset inside the layer's table buffer
  size_t gen_words;        // forward only: words of the channel-major table that precedes a tap-major one
                           // (the weight-gradient kernel always reads the channel-major table)
  size_t wt_off;           // backward-data: float offset of this class' weight matrix
};

// the loader waves run two chunks past the end (branch-free schedule): 48 invalid pad entries
static int kpad(int Kd) { return ((Kd + 15) / 16) * 16 + 48; }

static bool want_tap_major(int C) {
  static const int off = getenv("SLV_KORD_CHAN") ? 1 : 0;   // A/B switch: force the channel-major path
  const int cp = (C + 15) / 16 * 16;
  return !off && C >= 16 && cp * 10 <= C * 11;              // at most 10 % zero padding of K
}
static size_t tap_table_words(int nchunks) { return (size_t)2 * (nchunks + 4) + 64; }
static void finish(Desc& d, const Geom& g, bool with_generic) {
  d.Ntot = (long long)g.Bn * d.Q[0] * d.Q[1] * d.Q[2];
  d.Cp = (d.C + 15) / 16 * 16;
  d.kord = want_tap_major(d.C) ? KORD_TAP : KORD_CHAN;
  const size_t gen = (size_t)2 * kpad(d.C * d.ntaps) + 64;
  if (d.kord == KORD_TAP) {
    d.Kd = d.ntaps * d.Cp;
    d.gen_words = with_generic ? gen : 0;
    d.tab_words = d.gen_words + tap_table_words(d.Kd / 16);
  } else {
    d.Kd = d.C * d.ntaps;
    d.gen_words = 0;
    d.tab_words = gen;
  }
}

static Desc fwd_desc(const Geom& g) {
  Desc d;
  memset(&d, 0, sizeof(d));
  d.M = g.Cout;
  d.C = g.Cin;
  const int k[3] = {g.kt, g.kh, g.kw}, p[3] = {g.pt, g.ph, g.pw}, s[3] = {g.st, g.sh, g.sw};
  const int in[3] = {g.Ti, g.Hi, g.Wi}, out[3] = {g.To, g.Ho, g.Wo};
  for (int a = 0; a < k[0]; ++a)
    for (int b = 0; b < k[1]; ++b)
      for (int c = 0; c < k[2]; ++c) {
        const int j = d.ntaps++;
        d.taps[j] = (a * k[1] + b) * k[2] + c;
        d.delta[j][0] = a - p[0];
        d.delta[j][1] = b - p[1];
        d.delta[j][2] = c - p[2];
      }
  for (int i = 0; i < 3; ++i) {
    d.Q[i] = out[i]; d.mul[i] = s[i]; d.S[i] = in[i];
    d.dmul[i] = 1; d.dorg[i] = 0; d.D[i] = out[i];
  }
  finish(d, g, true);
  return d;
}

// class index c in [0, st*sh*sw): parity (c0,c1,c2) of the conv-INPUT position
static int dgrad_descs(const Geom& g, Desc* out8) {
  const int k[3] = {g.kt, g.kh, g.kw}, p[3] = {g.pt, g.ph, g.pw}, s[3] = {g.st, g.sh, g.sw};
  const int in[3] = {g.Ti, g.Hi, g.Wi}, out[3] = {g.To, g.Ho, g.Wo};
  int n = 0;
  size_t tab_off = 0, wt_off = 0;
  for (int c0 = 0; c0 < s[0]; ++c0)
    for (int c1 = 0; c1 < s[1]; ++c1)
      for (int c2 = 0; c2 < s[2]; ++c2) {
        const int cls[3] = {c0, c1, c2};
        Desc d;
        memset(&d, 0, sizeof(d));
        d.M = g.Cin;
        d.C = g.Cout;
        bool empty = false;
        for (int i = 0; i < 3; ++i) {
          d.Q[i] = (in[i] - cls[i] + s[i] - 1) / s[i];
          if (d.Q[i] <= 0) empty = true;
          d.mul[i] = 1; d.S[i] = out[i];
          d.dmul[i] = s[i]; d.dorg[i] = cls[i]; d.D[i] = in[i];
        }
        if (empty) continue;
        for (int a = 0; a < k[0]; ++a)
          for (int b = 0; b < k[1]; ++b)
            for (int c = 0; c < k[2]; ++c) {
              const int kk[3] = {a, b, c};
              bool ok = true;
              int dl[3];
              for (int i = 0; i < 3; ++i) {
                const int num = cls[i] + p[i] - kk[i];
                if (((num % s[i]) + s[i]) % s[i] != 0) ok = false;
                dl[i] = (num >= 0) ? num / s[i] : -((-num) / s[i]);
              }
              if (!ok) continue;
              const int j = d.ntaps++;
              d.taps[j] = (a * k[1] + b) * k[2] + c;
              for (int i = 0; i < 3; ++i) d.delta[j][i] = dl[i];
            }
        finish(d, g, false);
        d.tab_off = tab_off;
        d.wt_off = wt_off;
        tab_off += d.tab_words;
        wt_off += (size_t)g.Cin * d.Kd;
        out8[n++] = d;
      }
  return n;
}

static void fill_tapd(const Desc& d, int32_t* td) {
  for (int j = 0; j < 64; ++j)
    td[j] = j < d.ntaps ? ((d.delta[j][0] + 64) | ((d.delta[j][1] + 64) << 8) | ((d.delta[j][2] + 64) << 16)) : 0;
}
// channel-major table: one {offset, tap | chan << 8} entry per k = c*ntaps + j
static void fill_table_generic(const Desc& d, int32_t* w) {
  const int Sprod = d.S[0] * d.S[1] * d.S[2];
  const int Kg = d.C * d.ntaps, KP = kpad(Kg);
  for (int c = 0; c < d.C; ++c)
    for (int j = 0; j < d.ntaps; ++j) {
      const int kidx = c * d.ntaps + j;
      w[2 * kidx] = c * Sprod + d.delta[j][0] * d.S[1] * d.S[2] + d.delta[j][1] * d.S[2] + d.delta[j][2];
      w[2 * kidx + 1] = j | (c << 8);
    }
  for (int kidx = Kg; kidx < KP; ++kidx) {
    w[2 * kidx] = 0;
    w[2 * kidx + 1] = 63;  // tap 63 is never valid
  }
  fill_tapd(d, w + 2 * KP);
}
// tap-major table: one entry per 16-deep chunk {offset of (tap j, channel c0), j | c0 << 8}.
// K order k = ((c/16)*ntaps + j)*16 + c%16: the chunks of one 16-channel group visit its taps back to back,
// so the shifted re-reads of the same input rows hit L1/L2 (a whole-tensor tap sweep between them cost
// 4.1 GB of L2 misses per launch on the layer-1 backward-data conv, PMC FETCH_SIZE).
static void fill_table_tap(const Desc& d, int32_t* w) {
  const int Sprod = d.S[0] * d.S[1] * d.S[2];
  const int nch = d.Kd / 16;
  for (int ch = 0; ch < nch; ++ch) {
    const int j = ch % d.ntaps, c0 = (ch / d.ntaps) * 16;
    w[2 * ch] = c0 * Sprod + d.delta[j][0] * d.S[1] * d.S[2] + d.delta[j][1] * d.S[2] + d.delta[j][2];
    w[2 * ch + 1] = j | (c0 << 8);
  }
  for (int ch = nch; ch < nch + 4; ++ch) {
    w[2 * ch] = 0;
    w[2 * ch + 1] = 63;
  }
  fill_tapd(d, w + 2 * (nch + 4));
}
static void fill_table(const Desc& d, int32_t* w) {
  if (d.kord == KORD_TAP) {
    if (d.gen_words) fill_table_generic(d, w);
    fill_table_tap(d, w + d.gen_words);
  } else {
    fill_table_generic(d, w);
  }
}

// Tile choice for the M x ncols output: minimise  padded work / (tile efficiency * chip fill).
// Big tiles amortise the operand loads best but the small late layers (B*T*H*W = 1568 columns at
// cfg2's layer4) would leave most of the 256 CUs idle with them; measured relative efficiencies.
static void pick_tile(int M, long long ncols, int* mt_out, int* nt_out) {
  static const struct { int mt, nt; double eff; } cand[] = {
      {9, 2, 1.00}, {8, 2, 0.97}, {15, 1, 0.90}, {4, 2, 0.90}, {9, 1, 0.85}, {8, 1, 0.82}, {4, 1, 0.70}};
  {  // large outputs: least-padding row tile, 128-column tile (many workgroups per CU anyway)
    const int mt = pick_mt(M), nt = mt >= 15 ? 1 : 2;
    const long long blocks = (long long)((M + mt * 16 - 1) / (mt * 16)) * ((ncols + nt * 64 - 1) / (nt * 64));
    *mt_out = mt;
    *nt_out = nt;
    if (blocks >= 768) return;
  }
  double best = 1e300;
  for (const auto& c : cand) {
    const long long bm = c.mt * 16, bn = c.nt * 64;
    const long long nbm = (M + bm - 1) / bm, nbn = (ncols + bn - 1) / bn;
    const double blocks = (double)(nbm * nbn);
    const double slots = 512.0;  // ~2 resident workgroups per CU
    const double waves = (double)((long long)((blocks + slots - 1) / slots));
    const double fill = blocks / (waves * slots);
    const double cost = (double)(nbm * bm) * (double)(nbn * bn) / (c.eff * fill);
    if (cost < best) { best = cost; *mt_out = c.mt; *nt_out = c.nt; }
  }
}

// Split-K for the CONV launches of the late layers: with B*T*H*W of a few thousand columns even the
// smallest tile leaves most CUs idle, but K = Cin*taps is thousands deep.  The K range is cut into
// `splits` slices, each slice writes a private partial output (workspace) and a fixed-order reduce
// kernel sums them (+ residual addend / BN statistics).  Deterministic; no atomics.
static int conv_max_splits() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("SLV_CONV_SPLITS");
    v = e ? atoi(e) : 8;
    if (v < 1) v = 1;
    if (v > 16) v = 16;
  }
  return v;
}
static void plan_conv(int M, long long ncols, int Kd, int* mt_out, int* nt_out, int* splits_out) {
  *splits_out = 1;
  const int mt = pick_mt(M), nt = mt >= 15 ? 1 : 2;
  const long long blocks = (long long)((M + mt * 16 - 1) / (mt * 16)) * ((ncols + nt * 64 - 1) / (nt * 64));
  if (blocks > 0 && blocks < 768) {
    const int chunks = (Kd + 15) / 16;
    static const int target = getenv("SLV_SPLIT_TARGET") ? atoi(getenv("SLV_SPLIT_TARGET")) : 1536;
    static const int minch = getenv("SLV_SPLIT_MINCH") ? atoi(getenv("SLV_SPLIT_MINCH")) : 32;
    long long sp = (target + blocks - 1) / blocks;
    if (sp > chunks / minch) sp = chunks / minch;   // >= 32 chunks (512 k) per slice
    if (sp > conv_max_splits()) sp = conv_max_splits();
    if (sp >= 2) {
      const int cps = (chunks + (int)sp - 1) / (int)sp;
      *mt_out = mt;
      *nt_out = nt;
      *splits_out = (chunks + cps - 1) / cps;
      return;
    }
  }
  pick_tile(M, ncols, mt_out, nt_out);
}

// out[i] = sum_s partial[s][i] (+ addend[i])   (fixed order; addend may alias out)
__global__ void conv_splitk_reduce_kernel(const float* __restrict__ part, const float* addend, float* out, size_t n,
                                          int splits) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = part[i];
    for (int s = 1; s < splits; ++s) v += part[(size_t)s * n + i];
    if (addend) v += addend[i];
    out[i] = v;
  }
}
// forward conv: y = sum of the K-slice partials + the per-channel statistics partials the fused epilogue
// would have produced.  One wave per (channel, column block of BN lattice columns).
__global__ __launch_bounds__(256) void conv_splitk_reduce_stats_kernel(const float* __restrict__ part,
                                                                        float* __restrict__ y, float* __restrict__ ssum,
                                                                        float* __restrict__ ssq, int M, long long Ntot,
                                                                        int P, int BN, int nblkN, size_t total,
                                                                        int splits) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const long long n0 = (long long)blockIdx.x * BN;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < BN; c += 64) {
    const long long n = n0 + c;
    if (n < Ntot) {
      const long long b = n / P;
      const size_t ad = ((size_t)b * M + m) * P + (size_t)(n - b * P);
      float v = part[ad];
      for (int s = 1; s < splits; ++s) v += part[(size_t)s * total + ad];
      y[ad] = v;
      s1 += v;
      s2 += v * v;
    }
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) {
    ssum[(size_t)m * nblkN + blockIdx.x] = s1;
    ssq[(size_t)m * nblkN + blockIdx.x] = s2;
  }
}

// backward-data split-K reduce that also forms the BatchNorm-backward partial sums of the layer that
// produced the conv input (see IgemmArgs::R): one wave per (channel, 256 flattened (b,p) positions).
__global__ __launch_bounds__(256) void conv_splitk_reduce_bnr_kernel(const float* __restrict__ part,
                                                                      const float* addend, float* out,
                                                                      const float* __restrict__ R,
                                                                      const float* __restrict__ rss,
                                                                      const float* __restrict__ rmi,
                                                                      float* __restrict__ rpart, int M, unsigned P,
                                                                      const FastDiv dP, unsigned ntot, size_t total,
                                                                      int splits, int rslots) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float ps = rss[m], ph = rss[M + m], pm = rmi[m], pi = rmi[M + m];
  const unsigned n0 = blockIdx.x * 256u;
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned n = n0 + q * 64 + lane;
    if (n < ntot) {
      const unsigned b = fdiv(n, dP);
      const size_t ad = ((size_t)b * M + m) * P + (size_t)(n - b * P);
      float v = part[ad];
      for (int s = 1; s < splits; ++s) v += part[(size_t)s * total + ad];
      if (addend) v += addend[ad];
      out[ad] = v;
      const float xv = R[ad];
      const float gm = (xv * ps + ph > 0.f) ? v : 0.f;
      s0 += gm;
      s1 += gm * ((xv - pm) * pi);
    }
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if (lane == 0) {
    float* o = rpart + ((size_t)m * rslots + blockIdx.x) * 2;
    o[0] = s0;
    o[1] = s1;
  }
}

template <int MODE>
static int dispatch(const IgemmArgs& a0, int mt, int nt, int splits, hipStream_t st, int mf = 0) {
  IgemmArgs a = a0;
  const int bm = mt * 16, bn = nt * 64;
  a.nblkM = (a.M + bm - 1) / bm;
  a.nblkN = (int)((a.Ntot + bn - 1) / bn);
  if (a.nblkN == 0) return 0;
  // 16-byte A loads when the layout allows it (see igemm.hpp, template flag VA)
  bool vec_a;
  if (MODE == MODE_WGRAD) vec_a = ((a.To * a.Ho * a.Wo) % 4 == 0) && (a.Ptot % 4 == 0);
  else vec_a = (a.Kd % 4 == 0) && (((size_t)a.A & 15) == 0);
  if (getenv("SLV_NO_VECA")) vec_a = false;
#define SLV_CASE(MT_, NT_) \
  if (mt == MT_ && nt == NT_) { launch_igemm<MODE, MT_, NT_>(a, splits, vec_a, st); return 0; }
#define SLV_CASE_MF(MT_, MT32_) \
  if (mt == MT_ && nt == 2) { launch_igemm<MODE, MT32_, 1, 1>(a, splits, vec_a, st); return 0; }
  if (mf) {
    SLV_CASE_MF(4, 2) SLV_CASE_MF(6, 3) SLV_CASE_MF(8, 4)
    return -1;
  }
  SLV_CASE(4, 1) SLV_CASE(4, 2) SLV_CASE(8, 1) SLV_CASE(8, 2)
  SLV_CASE(9, 1) SLV_CASE(9, 2) SLV_CASE(15, 1)
#undef SLV_CASE
#undef SLV_CASE_MF
  return -1;
}

static void conv_args(IgemmArgs& a, const Geom& g, const Desc& d, const int32_t* tab_dev) {
  memset(&a, 0, sizeof(a));
  a.M = d.M; a.Kd = d.Kd; a.Ntot = d.Ntot; a.Cb = d.C; a.ntaps = d.ntaps;
  a.kord = d.kord;
  a.sprod4 = (unsigned)(d.S[0] * d.S[1] * d.S[2]) * 4u;
  const int32_t* t = tab_dev + d.tab_off + d.gen_words;
  a.tab = (const int2*)t;
  a.tapd = (const int*)(t + (d.kord == KORD_TAP ? 2 * (d.Kd / 16 + 4) : 2 * kpad(d.Kd)));
  a.Q0 = d.Q[0]; a.Q1 = d.Q[1]; a.Q2 = d.Q[2];
  a.mul0 = d.mul[0]; a.mul1 = d.mul[1]; a.mul2 = d.mul[2];
  a.S0 = d.S[0]; a.S1 = d.S[1]; a.S2 = d.S[2];
  a.sbatch = (long long)d.C * d.S[0] * d.S[1] * d.S[2];
  a.dmul0 = d.dmul[0]; a.dmul1 = d.dmul[1]; a.dmul2 = d.dmul[2];
  a.dorg0 = d.dorg[0]; a.dorg1 = d.dorg[1]; a.dorg2 = d.dorg[2];
  a.D0 = d.D[0]; a.D1 = d.D[1]; a.D2 = d.D[2];
  a.A_bytes = (unsigned)((size_t)d.M * d.Kd * 4);
  a.B_bytes = (unsigned)((size_t)g.Bn * a.sbatch * 4);
}

struct TapMap {
  int off[64], nt[64], j[64];
};
// Per-step weight re-layouts (one read of w):
//   wf (forward, tap-major layers only)  wf[co][((ci/16)*taps + tap)*16 + ci%16]        = w[co][ci][tap]
//   wt (backward-data), per parity class  channel-major: wt_c[ci][co*nt_c + j]           = w[co][ci][tap_j]
//                                         tap-major: wt_c[ci][((co/16)*nt_c + j)*16 + co%16] = w[co][ci][tap_j]
// Padding channels (ci >= Cin resp. co >= Cout) stay zero (buffers are cleared first when padded).
__global__ void w_transform_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wt,
                                   int Cout, int Cin, int taps, const TapMap tm, int CpIn, int CpOut) {
  const size_t n = (size_t)Cout * Cin * taps;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const size_t r = i / taps;
    const int ci = (int)(r % Cin), co = (int)(r / Cin);
    const float v = w[i];
    if (wf) wf[(size_t)co * taps * CpIn + ((size_t)(ci >> 4) * taps + tap) * 16 + (ci & 15)] = v;
    if (wt && tm.nt[tap] != 0) {
      if (CpOut) wt[(size_t)tm.off[tap] + (size_t)ci * tm.nt[tap] * CpOut + ((size_t)(co >> 4) * tm.nt[tap] + tm.j[tap]) * 16 + (co & 15)] = v;
      else wt[(size_t)tm.off[tap] + ((size_t)ci * Cout + co) * tm.nt[tap] + tm.j[tap]] = v;
    }
  }
}

// dW[i] = sum_s partial[s][i], fixed order.  The weight gradients have few elements (83 k for layer1)
// but up to hundreds of K-slices, so the slice loop is spread over G waves per element group and
// unrolled 8-fold (independent loads in flight); partial sums are combined in a fixed tree.
template <int G>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                            size_t n, int splits) {
  constexpr int EPB = 256 / G;   // elements per block
  __shared__ float red[256];
  const int e = threadIdx.x % EPB, grp = threadIdx.x / EPB;
  const size_t i = (size_t)blockIdx.x * EPB + e;
  const int per = (splits + G - 1) / G;
  const int s0 = grp * per, s1 = (s0 + per < splits) ? s0 + per : splits;
  float v = 0.f;
  if (i < n) {
    const float* p = part + i;
    int s = s0;
    for (; s + 8 <= s1; s += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = p[(size_t)(s + u) * n];
      v += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    for (; s < s1; ++s) v += p[(size_t)s * n];
  }
  if constexpr (G == 1) {
    if (i < n) out[i] = v;
  } else {
    red[threadIdx.x] = v;
    __syncthreads();
    if (grp == 0 && i < n) {
      float r = red[e];
#pragma unroll
      for (int g2 = 1; g2 < G; ++g2) r += red[g2 * EPB + e];
      out[i] = r;
    }
  }
}
static void launch_splitk_reduce(const float* ws, float* out, size_t nel, int splits, hipStream_t st) {
  if (splits >= 16) {
    hipLaunchKernelGGL((splitk_reduce_kernel<4>), dim3((unsigned)((nel + 63) / 64)), dim3(256), 0, st, ws, out, nel, splits);
  } else {
    hipLaunchKernelGGL((splitk_reduce_kernel<1>), dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, ws, out, nel, splits);
  }
}

static int wgrad_splits(const Geom& g, int mt, int nt) {
  const long long Ptot = (long long)g.Bn * g.To * g.Ho * g.Wo;
  const long long chunks = (Ptot + 15) / 16;
  const int taps = g.kt * g.kh * g.kw;
  const long long tiles = (long long)((g.Cout + mt * 16 - 1) / (mt * 16)) * ((g.Cin * taps + nt * 64 - 1) / (nt * 64));
  long long s = 768 / tiles;                     // one full round of 3 workgroups per CU (measured best: 765 of 768)
  const long long maxs = (chunks + 15) / 16;     // at least 16 chunks (256 positions) per slice
  if (s > maxs) s = maxs;
  if (s > 512) s = 512;
  if (s < 1) s = 1;
  return (int)s;
}

}  // namespace slv

using namespace slv;

extern "C" {

// int32 words of the table buffer: dgrad == 0 -> forward/weight-gradient table, 1 -> all parity classes
int32_t slv_conv_table_len(const int32_t* geom, int dgrad) {
  Geom g;
  if (read_geom(geom, g) != 0) return -1;
  if (!dgrad) return (int32_t)fwd_desc(g).tab_words;
  Desc ds[8];
  const int n = dgrad_descs(g, ds);
  size_t t = 0;
  for (int i = 0; i < n; ++i) t += ds[i].tab_words;
  return (int32_t)t;
}

int slv_conv_table(const int32_t* geom, int dgrad, int32_t* tab_host_out) {
  Geom g;
  SLV_CHECK_ARG(read_geom(geom, g) == 0 && tab_host_out, "invalid geometry");
  if (!dgrad) {
    fill_table(fwd_desc(g), tab_host_out);
    return 0;
  }
  Desc ds[8];
  const int n = dgrad_descs(g, ds);
  for (int i = 0; i < n; ++i) fill_table(ds[i], tab_host_out + ds[i].tab_off);
  return 0;
}

// ---- launch configurations --------------------------------------------------------------------
// cfg == 0: built-in heuristic.  Otherwise mt | nt << 8 | splits << 16, one of slv_conv_configs():
// the host may time the candidates once per layer shape, which is what the reference does through
// cudnn.benchmark = True (main.py:187).
// mt / nt: block tile in units of 16 rows / 64 columns.  mf = 1: the 32x32x2 MFMA variant of that tile
// (block tiles 64x128, 96x128, 128x128 = (4,2), (6,2), (8,2)).
struct Cfg {
  int mt, nt, sp, mf;
};
static bool tile_ok(int mt, int nt, int mf) {
  if (mf) return nt == 2 && (mt == 4 || mt == 6 || mt == 8);
  return ((mt == 4 || mt == 8 || mt == 9) && (nt == 1 || nt == 2)) || (mt == 15 && nt == 1);
}
static int32_t pack_cfg(int mt, int nt, int sp, int mf) { return mt | (nt << 8) | (mf << 12) | (sp << 16); }
static int unpack_cfg(int32_t cfg, Cfg& c) {
  c.mt = cfg & 255; c.nt = (cfg >> 8) & 15; c.mf = (cfg >> 12) & 15; c.sp = (cfg >> 16) & 0x7FFF;
  return (c.mf <= 1 && tile_ok(c.mt, c.nt, c.mf) && c.sp >= 1) ? 0 : -1;
}
static int clamp_splits(int sp, long long chunks) {
  if (sp > chunks) sp = (int)(chunks > 0 ? chunks : 1);
  if (sp < 1) sp = 1;
  const long long cps = (chunks + sp - 1) / sp;
  return cps > 0 ? (int)((chunks + cps - 1) / cps) : 1;   // no empty slices
}
static int fwd_cfg(const Geom& g, int32_t cfg, Cfg& c) {
  const long long P = (long long)g.Bn * g.To * g.Ho * g.Wo;
  const int Kd = g.Cin * g.kt * g.kh * g.kw;
  c.mf = 0;
  if (cfg == 0) { plan_conv(g.Cout, P, Kd, &c.mt, &c.nt, &c.sp); return 0; }
  if (unpack_cfg(cfg, c) != 0 || c.sp > 64) return -1;
  c.sp = clamp_splits(c.sp, (Kd + 15) / 16);
  return 0;
}
// backward-data: per parity class tile (heuristic) or one tile for all (cfg); one common slice count
static int dgrad_cfg(const Geom& g, const Desc* ds, int n, int32_t cfg, Cfg* per_class, int* sp_out) {
  int sp = 1;
  if (cfg == 0) {
    for (int i = 0; i < n; ++i) {
      per_class[i].mf = 0;
      plan_conv(ds[i].M, ds[i].Ntot, ds[i].Kd, &per_class[i].mt, &per_class[i].nt, &per_class[i].sp);
      if (per_class[i].sp > sp) sp = per_class[i].sp;
    }
  } else {
    Cfg c;
    if (unpack_cfg(cfg, c) != 0 || c.sp > 64) return -1;
    int maxchunks = 1;
    for (int i = 0; i < n; ++i) {
      per_class[i] = c;
      if ((ds[i].Kd + 15) / 16 > maxchunks) maxchunks = (ds[i].Kd + 15) / 16;
    }
    sp = clamp_splits(c.sp, maxchunks);
  }
  *sp_out = sp;
  return 0;
}
static int wgrad_cfg(const Geom& g, int32_t cfg, Cfg& c) {
  const long long chunks = ((long long)g.Bn * g.To * g.Ho * g.Wo + 15) / 16;
  c.mf = 0;
  if (cfg == 0) {
    c.mt = pick_mt(g.Cout);
    c.nt = c.mt >= 15 ? 1 : 2;
    c.sp = wgrad_splits(g, c.mt, c.nt);
    return 0;
  }
  if (unpack_cfg(cfg, c) != 0 || c.sp > 1024) return -1;
  c.sp = clamp_splits(c.sp, chunks);
  return 0;
}

int32_t slv_conv_configs(const int32_t* geom, int op, int32_t* out, int32_t max_out) {
  Geom g;
  if (read_geom(geom, g) != 0 || !out || max_out <= 0 || op < 0 || op > 2) return -1;
  static const int tiles[10][3] = {{9, 2, 0}, {8, 2, 0}, {15, 1, 0}, {4, 2, 0}, {9, 1, 0}, {8, 1, 0}, {4, 1, 0},
                                   {8, 2, 1}, {6, 2, 1}, {4, 2, 1}};
  const int taps = g.kt * g.kh * g.kw;
  int M;
  long long N, chunks;
  if (op == 0) { M = g.Cout; N = (long long)g.Bn * g.To * g.Ho * g.Wo; chunks = (g.Cin * taps + 15) / 16; }
  else if (op == 1) {
    Desc ds[8];
    const int n = dgrad_descs(g, ds);
    M = g.Cin; N = 0; chunks = 1;
    for (int i = 0; i < n; ++i) {
      if (ds[i].Ntot > N) N = ds[i].Ntot;
      if ((ds[i].Kd + 15) / 16 > chunks) chunks = (ds[i].Kd + 15) / 16;
    }
  } else { M = g.Cout; N = (long long)g.Cin * taps; chunks = ((long long)g.Bn * g.To * g.Ho * g.Wo + 15) / 16; }
  long long minpad = 1LL << 62;
  for (const auto& t : tiles) {
    const long long bm = t[0] * 16, bn = t[1] * 64;
    const long long pad = ((M + bm - 1) / bm) * bm * (((N + bn - 1) / bn) * bn);
    if (pad < minpad) minpad = pad;
  }
  int cnt = 0;
  for (const auto& t : tiles) {
    const long long bm = t[0] * 16, bn = t[1] * 64;
    const long long nb = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    if ((double)(nb * bm * bn) > 1.35 * (double)minpad) continue;   // too much padded work
    int cand[12], nc = 0;
    if (op != 2) {
      static const int sps[8] = {1, 2, 3, 4, 6, 8, 12, 16};
      for (int sp : sps) {
        if (sp > 1 && (nb >= 1536 || chunks / sp < 8 || nb * sp > 8192)) continue;
        cand[nc++] = sp;
      }
    } else {
      const int s0 = wgrad_splits(g, t[0], t[1]);
      const int raw[5] = {s0 / 2, (s0 * 3) / 4, s0, (s0 * 3) / 2, s0 * 2};
      for (int r : raw) {
        const int sp = clamp_splits(r, chunks);
        bool dup = false;
        for (int j = 0; j < nc; ++j) dup |= cand[j] == sp;
        if (!dup) cand[nc++] = sp;
      }
    }
    for (int j = 0; j < nc && cnt < max_out; ++j) out[cnt++] = pack_cfg(t[0], t[1], cand[j], t[2]);
  }
  return cnt;
}

int32_t slv_conv_fwd_nblk(const int32_t* geom, int32_t cfg) {
  Geom g;
  Cfg c;
  if (read_geom(geom, g) != 0 || fwd_cfg(g, cfg, c) != 0) return -1;
  const long long P = (long long)g.Bn * g.To * g.Ho * g.Wo;
  return (int32_t)((P + c.nt * 64 - 1) / (c.nt * 64));
}

size_t slv_conv_fwd_ws_bytes(const int32_t* geom, int32_t cfg) {
  Geom g;
  Cfg c;
  if (read_geom(geom, g) != 0 || fwd_cfg(g, cfg, c) != 0) return 0;
  const long long P = (long long)g.Bn * g.To * g.Ho * g.Wo;
  return c.sp > 1 ? sizeof(float) * (size_t)c.sp * g.Cout * (size_t)P : 0;
}

int slv_conv_fwd(const int32_t* geom, const float* x, const float* w, const float* wf, const int32_t* tab,
                 const float* in_scale_shift, int in_relu, float* y, float* stat_sum, float* stat_sq,
                 void* ws, size_t ws_bytes, int32_t cfg, slv_stream_t stream) {
  Geom g;
  SLV_CHECK_ARG(read_geom(geom, g) == 0, "invalid geometry");
  SLV_CHECK_ARG(x && tab && y, "null pointer");
  Cfg c;
  SLV_CHECK_ARG(fwd_cfg(g, cfg, c) == 0, "invalid launch configuration");
  const Desc d = fwd_desc(g);
  IgemmArgs a;
  conv_args(a, g, d, tab);
  if (d.kord == KORD_TAP) SLV_CHECK_ARG(wf, "this layer reads the tap-major weights: pass wf (slv_conv_w_transform)");
  else SLV_CHECK_ARG(w, "null weight pointer");
  a.A = d.kord == KORD_TAP ? wf : w; a.B = x; a.C = y;
  a.pb = in_scale_shift; a.b_pro = in_scale_shift ? PRO_ACT : PRO_NONE; a.b_relu = in_relu;
  a.stat_sum = stat_sum; a.stat_sq = stat_sq;
  const int mt = c.mt, nt = c.nt, sp = c.sp;
  const size_t total = (size_t)g.Cout * (size_t)a.Ntot;
  if (sp > 1) {
    SLV_CHECK_ARG(ws && ws_bytes >= sizeof(float) * total * sp, "workspace too small (slv_conv_fwd_ws_bytes)");
    a.C = (float*)ws; a.stat_sum = a.stat_sq = nullptr;
    a.split_stride = (long long)total;
    const int chunks = (a.Kd + 15) / 16;
    a.chunks_per_split = (chunks + sp - 1) / sp;
  }
  SLV_CHECK_ARG(dispatch<MODE_CONV>(a, mt, nt, sp, (hipStream_t)stream, c.mf) == 0, "no kernel for tile");
  SLV_LAUNCH_CHECK();
  if (sp > 1) {
    if (stat_sum) {
      const int bn = nt * 64, nblkN = (int)((a.Ntot + bn - 1) / bn);
      hipLaunchKernelGGL(conv_splitk_reduce_stats_kernel, dim3(nblkN, (g.Cout + 3) / 4), dim3(256), 0,
                         (hipStream_t)stream, (const float*)ws, y, stat_sum, stat_sq, g.Cout, a.Ntot,
                         g.To * g.Ho * g.Wo, bn, nblkN, total, sp);
    } else {
      hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)),
                         dim3(256), 0, (hipStream_t)stream, (const float*)ws, (const float*)nullptr, y, total, sp);
    }
    SLV_LAUNCH_CHECK();
  }
  return 0;
}

// LDS-tiled variant for <= 9 taps: a workgroup (16 x 16 threads) owns a 16 (co) x 16 (ci) tile = one channel
// group of either target layout; reads are 16*taps-float runs, writes 64-byte runs; padding channels are
// written as zeros (no memset); no integer divisions.
__global__ __launch_bounds__(256) void w_transform_tiled_kernel(const float* __restrict__ w, float* __restrict__ wf,
                                                               float* __restrict__ wt, int Cout, int Cin, int taps,
                                                               const TapMap tm, int CpIn, int CpOut) {
  __shared__ float t[16][16 * 9 + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int ci0 = blockIdx.x * 16, co0 = blockIdx.y * 16;
  const int row = 16 * taps;
  {
    const int co = co0 + ty;
    const int cin_here = Cin - ci0 < 16 ? Cin - ci0 : 16;          // may be <= 0 in a pure padding tile
    const int rlim = (co < Cout && cin_here > 0) ? cin_here * taps : 0;
    const float* src = w + ((size_t)(co < Cout ? co : 0) * Cin + (ci0 < Cin ? ci0 : 0)) * taps;
    for (int r = tx; r < row; r += 16) t[ty][r] = r < rlim ? src[r] : 0.f;
  }
  __syncthreads();
  if (wf) {   // thread (tx = ci, ty = co)
    const int co = co0 + ty, ci = ci0 + tx;
    if (co < Cout && ci < CpIn) {
      float* dst = wf + (size_t)co * taps * CpIn + (size_t)(ci0 >> 4) * taps * 16 + tx;
      for (int tap = 0; tap < taps; ++tap) dst[tap * 16] = t[ty][tx * taps + tap];
    }
  }
  if (wt) {   // thread (tx = co, ty = ci)
    const int co = co0 + tx, ci = ci0 + ty;
    const int colim = CpOut ? CpOut : Cout;
    if (ci < Cin && co < colim) {
      for (int tap = 0; tap < taps; ++tap) {
        const int nt = tm.nt[tap];
        if (nt == 0) continue;
        const float v = t[tx][ty * taps + tap];
        if (CpOut) wt[(size_t)tm.off[tap] + (size_t)ci * nt * CpOut + ((size_t)(co0 >> 4) * nt + tm.j[tap]) * 16 + tx] = v;
        else wt[(size_t)tm.off[tap] + ((size_t)ci * Cout + co) * nt + tm.j[tap]] = v;
      }
    }
  }
}

size_t slv_conv_wf_elems(const int32_t* geom) {
  Geom g;
  if (read_geom(geom, g) != 0) return 0;
  const Desc d = fwd_desc(g);
  return d.kord == KORD_TAP ? (size_t)d.M * d.Kd : 0;
}

size_t slv_conv_wt_elems(const int32_t* geom) {
  Geom g;
  if (read_geom(geom, g) != 0) return 0;
  Desc ds[8];
  const int n = dgrad_descs(g, ds);
  size_t t = 0;
  for (int i = 0; i < n; ++i) t += (size_t)g.Cin * ds[i].Kd;
  return t > 0 ? t : 1;
}

int slv_conv_w_transform(const int32_t* geom, const float* w, float* wf, float* wt, slv_stream_t stream) {
  Geom g;
  SLV_CHECK_ARG(read_geom(geom, g) == 0 && w && (wf || wt), "invalid geometry or null pointer");
  const Desc df = fwd_desc(g);
  if (df.kord != KORD_TAP) wf = nullptr;   // the forward conv of this layer reads w directly
  Desc ds[8];
  const int n = dgrad_descs(g, ds);
  TapMap tm;
  memset(&tm, 0, sizeof(tm));
  const int taps = g.kt * g.kh * g.kw;
  int cp_out = 0;
  size_t wt_elems = 0;
  for (int i = 0; i < n; ++i) {
    wt_elems += (size_t)g.Cin * ds[i].Kd;
    if (ds[i].kord == KORD_TAP) cp_out = ds[i].Cp;
    for (int j = 0; j < ds[i].ntaps; ++j) {
      const int t = ds[i].taps[j];
      tm.off[t] = (int)ds[i].wt_off;
      tm.nt[t] = ds[i].ntaps;
      tm.j[t] = j;
    }
  }
  // taps whose parity class has an empty lattice (input extent smaller than the stride) keep nt = 0:
  // no input position ever sees them, the kernel skips them
  hipStream_t st = (hipStream_t)stream;
  if (taps <= 9) {
    const int cin_ext = wf ? df.Cp : g.Cin, cout_ext = cp_out ? cp_out : g.Cout;
    hipLaunchKernelGGL(w_transform_tiled_kernel, dim3((cin_ext + 15) / 16, (cout_ext + 15) / 16), dim3(256), 0, st, w,
                       wf, wt, g.Cout, g.Cin, taps, tm, df.Cp, cp_out);
    SLV_LAUNCH_CHECK();
    return 0;
  }
  if (wf && df.Cp != g.Cin) SLV_HIP(hipMemsetAsync(wf, 0, sizeof(float) * (size_t)df.M * df.Kd, st));
  if (wt && cp_out && cp_out != g.Cout) SLV_HIP(hipMemsetAsync(wt, 0, sizeof(float) * wt_elems, st));
  const size_t nel = (size_t)g.Cout * g.Cin * taps;
  hipLaunchKernelGGL(w_transform_kernel, dim3((unsigned)((nel + 255) / 256 < 4096 ? (nel + 255) / 256 : 4096)),
                     dim3(256), 0, st, w, wf, wt, g.Cout, g.Cin, taps, tm, df.Cp, cp_out);
  SLV_LAUNCH_CHECK();
  return 0;
}

size_t slv_conv_dgrad_ws_bytes(const int32_t* geom, int32_t cfg) {
  Geom g;
  if (read_geom(geom, g) != 0) return 0;
  Desc ds[8];
  Cfg pc[8];
  int sp;
  const int n = dgrad_descs(g, ds);
  if (dgrad_cfg(g, ds, n, cfg, pc, &sp) != 0) return 0;
  return sp > 1 ? sizeof(float) * (size_t)sp * g.Bn * g.Cin * g.Ti * g.Hi * g.Wi : 0;
}

// number of partial-sum slots per channel that slv_conv_dgrad writes into bnr_part
int32_t slv_conv_dgrad_bnr_slots(const int32_t* geom, int32_t cfg) {
  Geom g;
  if (read_geom(geom, g) != 0) return -1;
  Desc ds[8];
  Cfg pc[8];
  int sp;
  const int n = dgrad_descs(g, ds);
  if (dgrad_cfg(g, ds, n, cfg, pc, &sp) != 0) return -1;
  if (sp > 1) return (int32_t)(((long long)g.Bn * g.Ti * g.Hi * g.Wi + 255) / 256);
  long long t = 0;
  for (int i = 0; i < n; ++i) t += (ds[i].Ntot + pc[i].nt * 64 - 1) / (pc[i].nt * 64);
  return (int32_t)t;
}

int slv_conv_dgrad(const int32_t* geom, const float* dy, const float* wt, const int32_t* tab, float* dx,
                   const float* addend, const float* bnr_x, const float* bnr_scale_shift,
                   const float* bnr_mean_invstd, float* bnr_part, void* ws, size_t ws_bytes, int32_t cfg,
                   slv_stream_t stream) {
  Geom g;
  SLV_CHECK_ARG(read_geom(geom, g) == 0, "invalid geometry");
  SLV_CHECK_ARG(dy && wt && tab && dx, "null pointer");
  SLV_CHECK_ARG(!bnr_x || (bnr_scale_shift && bnr_mean_invstd && bnr_part), "BN-backward reduction needs all of its arguments");
  Desc ds[8];
  Cfg pc[8];
  int sp;
  const int n = dgrad_descs(g, ds);
  SLV_CHECK_ARG(dgrad_cfg(g, ds, n, cfg, pc, &sp) == 0, "invalid launch configuration");
  const size_t total = (size_t)g.Bn * g.Cin * g.Ti * g.Hi * g.Wi;
  if (sp > 1) SLV_CHECK_ARG(ws && ws_bytes >= sizeof(float) * total * sp, "workspace too small (slv_conv_dgrad_ws_bytes)");
  int rslots = 0;
  if (bnr_x && sp == 1)
    for (int i = 0; i < n; ++i) rslots += (int)((ds[i].Ntot + pc[i].nt * 64 - 1) / (pc[i].nt * 64));
  int rslot0 = 0;
  for (int i = 0; i < n; ++i) {
    const Desc& d = ds[i];
    IgemmArgs a;
    conv_args(a, g, d, tab);
    a.A = wt + d.wt_off; a.B = dy; a.C = dx; a.E = addend;
    a.b_pro = PRO_NONE;
    if (sp > 1) {
      a.C = (float*)ws; a.E = nullptr;
      a.split_stride = (long long)total;
      const int chunks = (a.Kd + 15) / 16;
      a.chunks_per_split = chunks > 0 ? (chunks + sp - 1) / sp : 1;
    } else if (bnr_x) {
      a.R = bnr_x; a.rss = bnr_scale_shift; a.rmi = bnr_mean_invstd; a.rpart = bnr_part;
      a.rslots = rslots; a.rslot0 = rslot0;
      rslot0 += (int)((d.Ntot + pc[i].nt * 64 - 1) / (pc[i].nt * 64));
    }
    SLV_CHECK_ARG(dispatch<MODE_CONV>(a, pc[i].mt, pc[i].nt, sp, (hipStream_t)stream, pc[i].mf) == 0, "no kernel for tile");
    SLV_LAUNCH_CHECK();
  }
  if (sp > 1) {
    if (bnr_x) {
      const unsigned P = (unsigned)(g.Ti * g.Hi * g.Wi);
      const unsigned ntot = (unsigned)g.Bn * P;
      const int slots = (int)((ntot + 255u) / 256u);
      hipLaunchKernelGGL(conv_splitk_reduce_bnr_kernel, dim3(slots, (g.Cin + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                         (const float*)ws, addend, dx, bnr_x, bnr_scale_shift, bnr_mean_invstd, bnr_part, g.Cin, P,
                         make_fastdiv(P), ntot, total, sp, slots);
    } else {
      hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)),
                         dim3(256), 0, (hipStream_t)stream, (const float*)ws, addend, dx, total, sp);
    }
    SLV_LAUNCH_CHECK();
  }
  return 0;
}

size_t slv_conv_wgrad_ws_bytes(const int32_t* geom, int32_t cfg) {
  Geom g;
  Cfg c;
  if (read_geom(geom, g) != 0 || wgrad_cfg(g, cfg, c) != 0) return 0;
  return c.sp > 1 ? sizeof(float) * (size_t)c.sp * g.Cout * g.Cin * g.kt * g.kh * g.kw : 0;
}

int slv_conv_wgrad(const int32_t* geom, const float* dy, const float* x_in, const float* in_scale_shift,
                   int in_relu, const int32_t* tab, float* dw, void* ws, size_t ws_bytes, int32_t cfg,
                   slv_stream_t stream) {
  Geom g;
  SLV_CHECK_ARG(read_geom(geom, g) == 0, "invalid geometry");
  SLV_CHECK_ARG(dy && x_in && tab && dw, "null pointer");
  Cfg c;
  SLV_CHECK_ARG(wgrad_cfg(g, cfg, c) == 0, "invalid launch configuration");
  const Desc d = fwd_desc(g);
  IgemmArgs a;
  memset(&a, 0, sizeof(a));
  const int taps = g.kt * g.kh * g.kw;
  a.tab = (const int2*)tab;                                  // channel-major table (first block of the forward table)
  a.tapd = (const int*)(tab + 2 * kpad(g.Cin * taps));
  a.Cin = g.Cin; a.Ti = g.Ti; a.Hi = g.Hi; a.Wi = g.Wi; a.Cout = g.Cout; a.To = g.To; a.Ho = g.Ho; a.Wo = g.Wo;
  a.st = g.st; a.sh = g.sh; a.sw = g.sw; a.pt = g.pt; a.ph = g.ph; a.pw = g.pw;
  a.A = dy;
  a.B = x_in; a.pb = in_scale_shift; a.b_pro = in_scale_shift ? PRO_ACT : PRO_NONE; a.b_relu = in_relu;
  a.A_bytes = (unsigned)((size_t)g.Bn * g.Cout * g.To * g.Ho * g.Wo * 4);
  a.B_bytes = (unsigned)((size_t)g.Bn * g.Cin * g.Ti * g.Hi * g.Wi * 4);
  a.M = g.Cout; a.Kd = 0; a.Ntot = (long long)g.Cin * taps; a.ldc = g.Cin * taps;
  a.Ptot = (long long)g.Bn * g.To * g.Ho * g.Wo;
  SLV_CHECK_ARG(a.Ptot < (1LL << 31), "more than 2^31 output positions");
  a.dPout = make_fastdiv((unsigned)(g.To * g.Ho * g.Wo));
  a.dHoWo = make_fastdiv((unsigned)(g.Ho * g.Wo));
  a.dWo = make_fastdiv((unsigned)g.Wo);
  // both operands by 16-byte loads when a quad of output positions maps to 4 consecutive aligned input elements
  a.vec_b = 0;
  if (g.sh == 1 && g.sw == 1 && (((size_t)x_in) & 15) == 0 && !getenv("SLV_NO_VECB")) {
    if (g.kh == 1 && g.kw == 1 && g.ph == 0 && g.pw == 0 && (g.Ho * g.Wo) % 4 == 0) a.vec_b = 1;
    else if ((g.kh == 1 || g.kh == 3) && (g.kw == 1 || g.kw == 3) && g.ph == g.kh / 2 && g.pw == g.kw / 2 &&
             g.Wo % 4 == 0) a.vec_b = 2;
  }
  const int mt = c.mt, nt = c.nt, splits = c.sp;
  const long long chunks = (a.Ptot + 15) / 16;
  a.chunks_per_split = (int)((chunks + splits - 1) / splits);
  const size_t nel = (size_t)g.Cout * g.Cin * taps;
  if (splits > 1) {
    SLV_CHECK_ARG(ws && ws_bytes >= sizeof(float) * nel * splits, "workspace too small");
    a.C = (float*)ws;
  } else {
    a.C = dw;
  }
  SLV_CHECK_ARG(dispatch<MODE_WGRAD>(a, mt, nt, splits, (hipStream_t)stream, c.mf) == 0, "no kernel for tile");
  SLV_LAUNCH_CHECK();
  if (splits > 1) {
    launch_splitk_reduce((const float*)ws, dw, nel, splits, (hipStream_t)stream);
    SLV_LAUNCH_CHECK();
  }
  return 0;
}

int slv_gemm_nt(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int ldc,
                slv_stream_t stream) {
  SLV_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && ldc >= N, "null pointer or empty shape");
  SLV_CHECK_ARG((long long)M * K * 4 < 0xFFFFFFF0LL && (long long)N * K * 4 < 0xFFFFFFF0LL, "operand larger than 4 GiB");
  IgemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.B = B; a.bias = bias; a.C = C; a.M = M; a.Kd = K; a.Ntot = N; a.ldc = ldc;
  a.A_bytes = (unsigned)((size_t)M * K * 4);
  a.B_bytes = (unsigned)((size_t)N * K * 4);
  int mt, nt;
  pick_tile(M, N, &mt, &nt);
  SLV_CHECK_ARG(dispatch<MODE_GEMM>(a, mt, nt, 1, (hipStream_t)stream) == 0, "no kernel for tile");
  SLV_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

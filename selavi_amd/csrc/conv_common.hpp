// Shared host-side pieces of the implicit-GEMM convolution entry points (geometry, gather tables, tile / split-K
// planning, launch configurations, kernel dispatch).  Included by conv_fwd.hip, conv_dgrad.hip, conv_wgrad.hip and
// gemm.hip: the translation units are split only to compile the ~200 kernel instantiations in parallel.
#pragma once
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
  int x3;                  // this launch runs on the split-operand kernels (igemm3.hpp): the weight image is three bf16 planes
  size_t a_floats;         // size of the launch's weight image in floats (M*Kd fp32, or the x3 image)
};

// the split-operand kernels (igemm3.hpp), instantiated in conv_x3_*.hip: 0 = launched, -1 = no such tile
int x3_enabled();
struct IgemmArgs;
// x3 weight image: [K chunk of 32][plane 3][Mp = round16(M) rows][32 k] bf16, rows of 64 bytes with the 16-byte slots
// XOR-swizzled (cl_swz) -- the LDS image of igemm3_kernel's A tile, copied by LDS-DMA
static size_t x3_image_floats(int M, int Kd) { return (size_t)((Kd + 31) / 32) * 3 * ((M + 15) / 16 * 16) * 16; }

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
  d.x3 = d.kord == KORD_TAP && x3_enabled();
  d.a_floats = d.x3 ? x3_image_floats(d.M, d.Kd) : (size_t)d.M * d.Kd;
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
        wt_off += d.a_floats;
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
// Tile of a split-operand conv launch (igemm3.hpp) when the host did not time the candidates.  What benchmark mode picks
// on the trunks' shapes (profiles/r04_notes.md, table of tuned configurations): 144 x 128 where 144-row blocks pad no more
// than 128-row ones (M = 144, 288, 576, 921), else 128 x 64 -- the lean 128-row pipeline at three workgroups per CU; its
// 128-column form spills under that register cap and the 240-row tile is the slowest per FLOP -- and 64 x 128 up to 64 rows.
static void pick_tile_x3(int M, int Kd, int* mt_out, int* nt_out) {
  if (M <= 64) { *mt_out = 4; *nt_out = 2; return; }
  const int pad9 = (M + 143) / 144 * 144, pad8 = (M + 127) / 128 * 128;
  if (pad9 < pad8 || (pad9 == pad8 && M <= 144)) { *mt_out = 9; *nt_out = 2; }
  else if (Kd >= 4096) { *mt_out = 8; *nt_out = 4; }      // layer 4.1 spatial: K = 4608, 1568 columns -- the 8-wave 128 x 256 tile
  else { *mt_out = 8; *nt_out = 1; }
}
static void plan_conv(int M, long long ncols, int Kd, int* mt_out, int* nt_out, int* splits_out, bool x3 = false) {
  *splits_out = 1;
  int mt = pick_mt(M), nt = mt >= 15 ? 1 : 2;
  if (x3) pick_tile_x3(M, Kd, &mt, &nt);
  const long long blocks = (long long)((M + mt * 16 - 1) / (mt * 16)) * ((ncols + nt * 64 - 1) / (nt * 64));
  if (blocks > 0 && blocks < 768) {
    const int chunks = (Kd + 15) / 16;
    // tuning overrides for experiments; clamped so that a bad value cannot divide by zero
    static const int target = getenv("SLV_SPLIT_TARGET") ? (atoi(getenv("SLV_SPLIT_TARGET")) > 0 ? atoi(getenv("SLV_SPLIT_TARGET")) : 1) : 1536;
    static const int minch = getenv("SLV_SPLIT_MINCH") ? (atoi(getenv("SLV_SPLIT_MINCH")) > 0 ? atoi(getenv("SLV_SPLIT_MINCH")) : 1) : 32;
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
  if (x3) { *mt_out = mt; *nt_out = nt; return; }
  pick_tile(M, ncols, mt_out, nt_out);
}

int launch_x3_fwd(const IgemmArgs& a, int mt, int nt, int splits, hipStream_t st);
int launch_x3_dgrad(const IgemmArgs& a, int mt, int nt, int splits, hipStream_t st);
int launch_x3_wgrad(const IgemmArgs& a, int mt, int nt, int splits, bool vec_a, hipStream_t st);

// out[i] = sum_s partial[s][i] (+ addend[i])   (fixed order; addend may alias out)
template <int MODE, int SUBSET = SUB_ALL>
static int dispatch(const IgemmArgs& a0, int mt, int nt, int splits, hipStream_t st, int mf = 0) {
  IgemmArgs a = a0;
  const int bm = mt * 16, bn = nt * 64;
  a.nblkM = (a.M + bm - 1) / bm;
  a.nblkN = (int)((a.Ntot + bn - 1) / bn);
  if (a.nblkN == 0) return 0;
  if constexpr (MODE == MODE_CONV && (SUBSET == SUB_FWD || SUBSET == SUB_DGRAD)) {
    // tap-major layers (all but the stems): fp32 on the bf16 matrix cores with split operands (igemm3.hpp)
    if (a.kord == KORD_TAP && !mf && x3_enabled())
      return SUBSET == SUB_FWD ? launch_x3_fwd(a, mt, nt, splits, st) : launch_x3_dgrad(a, mt, nt, splits, st);
  }
  // 16-byte A loads when the layout allows it (see igemm.hpp, template flag VA)
  bool vec_a;
  if (MODE == MODE_WGRAD) vec_a = ((a.To * a.Ho * a.Wo) % 4 == 0) && (a.Ptot % 4 == 0);
  else vec_a = (a.Kd % 4 == 0) && (((size_t)a.A & 15) == 0);
  if (getenv("SLV_NO_VECA")) vec_a = false;
  if constexpr (MODE == MODE_WGRAD) {
    // split operands on the bf16 matrix cores (igemm3.hpp: quads of positions where the geometry allows, else element-wise)
    if (!mf && x3_enabled()) return launch_x3_wgrad(a, mt, nt, splits, vec_a, st);
  }
#define SLV_CASE(MT_, NT_) \
  if (mt == MT_ && nt == NT_) { launch_igemm<MODE, MT_, NT_, 0, SUBSET>(a, splits, vec_a, st); return 0; }
#define SLV_CASE_MF(MT_, MT32_) \
  if (mt == MT_ && nt == 2) { launch_igemm<MODE, MT32_, 1, 1, SUBSET>(a, splits, vec_a, st); return 0; }
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
  a.A_bytes = (unsigned)(d.a_floats * 4);
  a.B_bytes = (unsigned)((size_t)g.Bn * a.sbatch * 4);
}

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
  // one full round of 3 workgroups per CU (measured best: 765 of 768); the 144-row split-operand tile runs 2 per CU: three rounds
  // of 512 (what benchmark mode picks: 512 slices for the 3 tiles of layer 1, 84 for the 18 of layer 2.1, 42 for the 36 of 3.0)
  long long s = (x3_enabled() && mt == 9 ? 1536 : 768) / tiles;
  const long long maxs = (chunks + 15) / 16;     // at least 16 chunks (256 positions) per slice
  if (s > maxs) s = maxs;
  if (s > 512) s = 512;
  if (s < 1) s = 1;
  return (int)s;
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
  if (nt == 4) return mt == 4 || mt == 8 || mt == 9;      // 8-wave workgroups of the split-operand kernels (igemm3.hpp) only
  if (nt == 3) return mt == 4 || mt == 8 || mt == 9;      // 192-column tiles: split-operand weight gradient only (N = 576 = 3 x 192)
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
  if (cfg == 0) { plan_conv(g.Cout, P, Kd, &c.mt, &c.nt, &c.sp, x3_enabled() && want_tap_major(g.Cin)); return 0; }
  if (unpack_cfg(cfg, c) != 0 || c.sp > 64) return -1;
  // a split-operand layer's weight buffers hold the three-plane bf16 image (slv_conv_w_transform): the native 32x32x2 tiles
  // (mf = 1) would read it as an fp32 matrix -- such a configuration (a tune cache written for the native kernels) is invalid
  if (c.mf && x3_enabled() && want_tap_major(g.Cin)) return -1;
  c.sp = clamp_splits(c.sp, (Kd + 15) / 16);
  return 0;
}
// backward-data: per parity class tile (heuristic) or one tile for all (cfg); one common slice count
static int dgrad_cfg(const Geom& g, const Desc* ds, int n, int32_t cfg, Cfg* per_class, int* sp_out) {
  int sp = 1;
  if (cfg == 0) {
    for (int i = 0; i < n; ++i) {
      per_class[i].mf = 0;
      plan_conv(ds[i].M, ds[i].Ntot, ds[i].Kd, &per_class[i].mt, &per_class[i].nt, &per_class[i].sp, ds[i].x3 != 0);
      if (per_class[i].sp > sp) sp = per_class[i].sp;
    }
  } else {
    Cfg c;
    if (unpack_cfg(cfg, c) != 0 || c.sp > 64) return -1;
    int maxchunks = 1;
    for (int i = 0; i < n; ++i) {
      if (c.mf && ds[i].x3) return -1;      // (see fwd_cfg: the class reads an x3 weight image)
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
    if (x3_enabled()) pick_tile_x3(g.Cout, 0, &c.mt, &c.nt), c.nt = 2;      // (no 240-row tile: the slowest per FLOP)
    if (c.nt == 2 && x3_enabled()) {        // split-operand kernel: 192-column tiles where they pad less (Cin * taps = 576)
      const long long N = (long long)g.Cin * g.kt * g.kh * g.kw;
      if ((N + 191) / 192 * 192 < (N + 127) / 128 * 128) c.nt = 3;
    }
    c.sp = wgrad_splits(g, c.mt, c.nt);
    return 0;
  }
  if (unpack_cfg(cfg, c) != 0 || c.sp > 1024) return -1;
  c.sp = clamp_splits(c.sp, chunks);
  return 0;
}


static __global__ void conv_splitk_reduce_kernel(const float* __restrict__ part, const float* addend, float* out, size_t n,
                                          int splits) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = part[i];
    for (int s = 1; s < splits; ++s) v += part[(size_t)s * n + i];
    if (addend) v += addend[i];
    out[i] = v;
  }
}

}  // namespace slv

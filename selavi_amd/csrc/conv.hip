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
  int M, C, Kd, ntaps;
  int taps[64];            // linear tap ids (kt,kh,kw order) in this launch's k order
  int delta[64][3];        // source-coordinate delta of each tap
  int Q[3], mul[3], S[3];  // lattice dims, source multipliers, source dims
  int dmul[3], dorg[3], D[3];
  long long Ntot;
  size_t tab_words;        // int32 words of this launch's table block
  size_t tab_off;          // word offset inside the layer's table buffer
  size_t wt_off;           // backward-data: float offset of this class' weight matrix
};

// the loader waves run two chunks past the end (branch-free schedule): 48 invalid pad entries
static int kpad(int Kd) { return ((Kd + 15) / 16) * 16 + 48; }

static void finish(Desc& d, const Geom& g) {
  d.Kd = d.C * d.ntaps;
  d.Ntot = (long long)g.Bn * d.Q[0] * d.Q[1] * d.Q[2];
  d.tab_words = (size_t)2 * kpad(d.Kd) + 64;
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
  finish(d, g);
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
        finish(d, g);
        d.tab_off = tab_off;
        d.wt_off = wt_off;
        tab_off += d.tab_words;
        wt_off += (size_t)g.Cin * d.Kd;
        out8[n++] = d;
      }
  return n;
}

static void fill_table(const Desc& d, int32_t* w) {
  const int Sprod = d.S[0] * d.S[1] * d.S[2];
  const int KP = kpad(d.Kd);
  for (int c = 0; c < d.C; ++c)
    for (int j = 0; j < d.ntaps; ++j) {
      const int kidx = c * d.ntaps + j;
      w[2 * kidx] = c * Sprod + d.delta[j][0] * d.S[1] * d.S[2] + d.delta[j][1] * d.S[2] + d.delta[j][2];
      w[2 * kidx + 1] = j | (c << 8);
    }
  for (int kidx = d.Kd; kidx < KP; ++kidx) {
    w[2 * kidx] = 0;
    w[2 * kidx + 1] = 63;  // tap 63 is never valid
  }
  int32_t* td = w + 2 * KP;
  for (int j = 0; j < 64; ++j)
    td[j] = j < d.ntaps ? ((d.delta[j][0] + 64) | ((d.delta[j][1] + 64) << 8) | ((d.delta[j][2] + 64) << 16)) : 0;
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

template <int MODE>
static int dispatch(const IgemmArgs& a0, int mt, int nt, int splits, hipStream_t st) {
  IgemmArgs a = a0;
  const int bm = mt * 16, bn = nt * 64;
  a.nblkM = (a.M + bm - 1) / bm;
  a.nblkN = (int)((a.Ntot + bn - 1) / bn);
  if (a.nblkN == 0) return 0;
  // 16-byte A loads when the layout allows it (see igemm.hpp, template flag VA)
  bool vec_a;
  if (MODE == MODE_WGRAD) vec_a = ((a.To * a.Ho * a.Wo) % 4 == 0) && a.a_pro == PRO_NONE && (a.Ptot % 4 == 0);
  else vec_a = (a.Kd % 4 == 0) && (((size_t)a.A & 15) == 0);
  if (getenv("SLV_NO_VECA")) vec_a = false;
#define SLV_CASE(MT_, NT_) \
  if (mt == MT_ && nt == NT_) { launch_igemm<MODE, MT_, NT_>(a, splits, vec_a, st); return 0; }
  SLV_CASE(4, 1) SLV_CASE(4, 2) SLV_CASE(8, 1) SLV_CASE(8, 2)
  SLV_CASE(9, 1) SLV_CASE(9, 2) SLV_CASE(15, 1)
#undef SLV_CASE
  return -1;
}

static void conv_args(IgemmArgs& a, const Geom& g, const Desc& d, const int32_t* tab_dev) {
  memset(&a, 0, sizeof(a));
  a.M = d.M; a.Kd = d.Kd; a.Ntot = d.Ntot; a.Cb = d.C; a.ntaps = d.ntaps;
  a.tab = (const int2*)(tab_dev + d.tab_off);
  a.tapd = (const int*)(tab_dev + d.tab_off + 2 * kpad(d.Kd));
  a.Q0 = d.Q[0]; a.Q1 = d.Q[1]; a.Q2 = d.Q[2];
  a.mul0 = d.mul[0]; a.mul1 = d.mul[1]; a.mul2 = d.mul[2];
  a.S0 = d.S[0]; a.S1 = d.S[1]; a.S2 = d.S[2];
  a.sbatch = (long long)d.C * d.S[0] * d.S[1] * d.S[2];
  a.dmul0 = d.dmul[0]; a.dmul1 = d.dmul[1]; a.dmul2 = d.dmul[2];
  a.dorg0 = d.dorg[0]; a.dorg1 = d.dorg[1]; a.dorg2 = d.dorg[2];
  a.D0 = d.D[0]; a.D1 = d.D[1]; a.D2 = d.D[2];
  a.A_bytes = (unsigned)((size_t)d.M * d.Kd * 4);
  a.B_bytes = (unsigned)((size_t)g.Bn * a.sbatch * 4);
  a.B2_bytes = a.B_bytes;
}

struct TapMap {
  int off[64], nt[64], j[64];
};
// wt[class][ci][co*nt + j] = w[co][ci][tap]
__global__ void wt_transform_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin,
                                    int taps, const TapMap tm) {
  const size_t n = (size_t)Cout * Cin * taps;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    if (tm.nt[tap] == 0) continue;
    const size_t r = i / taps;
    const int ci = (int)(r % Cin), co = (int)(r / Cin);
    wt[(size_t)tm.off[tap] + ((size_t)ci * Cout + co) * tm.nt[tap] + tm.j[tap]] = w[i];
  }
}

// dW[i] = sum_s partial[s][i]   (fixed order)
__global__ void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, size_t n,
                                     int splits) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += part[(size_t)s * n + i];
    out[i] = v;
  }
}

static int wgrad_splits(const Geom& g, int mt, int nt) {
  const long long Ptot = (long long)g.Bn * g.To * g.Ho * g.Wo;
  const long long chunks = (Ptot + 15) / 16;
  const int taps = g.kt * g.kh * g.kw;
  const long long tiles = (long long)((g.Cout + mt * 16 - 1) / (mt * 16)) * ((g.Cin * taps + nt * 64 - 1) / (nt * 64));
  long long s = (1024 + tiles - 1) / tiles;      // aim at ~1024 workgroups
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

int32_t slv_conv_fwd_nblk(const int32_t* geom) {
  Geom g;
  if (read_geom(geom, g) != 0) return -1;
  const long long P = (long long)g.Bn * g.To * g.Ho * g.Wo;
  int mt, nt;
  pick_tile(g.Cout, P, &mt, &nt);
  return (int32_t)((P + nt * 64 - 1) / (nt * 64));
}

int slv_conv_fwd(const int32_t* geom, const float* x, const float* w, const int32_t* tab,
                 const float* in_scale_shift, int in_relu, float* y, float* stat_sum, float* stat_sq,
                 slv_stream_t stream) {
  Geom g;
  SLV_CHECK_ARG(read_geom(geom, g) == 0, "invalid geometry");
  SLV_CHECK_ARG(x && w && tab && y, "null pointer");
  const Desc d = fwd_desc(g);
  IgemmArgs a;
  conv_args(a, g, d, tab);
  a.A = w; a.B = x; a.C = y;
  a.pb = in_scale_shift; a.b_pro = in_scale_shift ? PRO_ACT : PRO_NONE; a.b_relu = in_relu;
  a.stat_sum = stat_sum; a.stat_sq = stat_sq;
  int mt, nt;
  pick_tile(a.M, a.Ntot, &mt, &nt);
  SLV_CHECK_ARG(dispatch<MODE_CONV>(a, mt, nt, 1, (hipStream_t)stream) == 0, "no kernel for tile");
  SLV_LAUNCH_CHECK();
  return 0;
}

/* wt: per stride-parity class c a [Cin][Cout*ntaps_c] matrix, classes concatenated (same total size as w) */
int slv_conv_wt_transform(const int32_t* geom, const float* w, float* wt, slv_stream_t stream) {
  Geom g;
  SLV_CHECK_ARG(read_geom(geom, g) == 0 && w && wt, "invalid geometry or null pointer");
  Desc ds[8];
  const int n = dgrad_descs(g, ds);
  TapMap tm;
  memset(&tm, 0, sizeof(tm));
  const int taps = g.kt * g.kh * g.kw;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < ds[i].ntaps; ++j) {
      const int t = ds[i].taps[j];
      tm.off[t] = (int)ds[i].wt_off;
      tm.nt[t] = ds[i].ntaps;
      tm.j[t] = j;
    }
  // taps whose parity class has an empty lattice (input extent smaller than the stride) keep nt = 0:
  // no input position ever sees them, the kernel skips them
  const size_t nel = (size_t)g.Cout * g.Cin * taps;
  hipLaunchKernelGGL(wt_transform_kernel, dim3((unsigned)((nel + 255) / 256 < 4096 ? (nel + 255) / 256 : 4096)),
                     dim3(256), 0, (hipStream_t)stream, w, wt, g.Cout, g.Cin, taps, tm);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_conv_dgrad(const int32_t* geom, const float* dy, const float* x_out, const float* wt,
                   const int32_t* tab, const float* bwd5, int relu, float* dx, const float* addend,
                   slv_stream_t stream) {
  Geom g;
  SLV_CHECK_ARG(read_geom(geom, g) == 0, "invalid geometry");
  SLV_CHECK_ARG(dy && wt && tab && dx, "null pointer");
  SLV_CHECK_ARG(!bwd5 && !x_out, "the on-load BN-backward prologue was removed: materialise dXout with slv_bn_bwd_apply");
  (void)relu;
  Desc ds[8];
  const int n = dgrad_descs(g, ds);
  for (int i = 0; i < n; ++i) {
    const Desc& d = ds[i];
    IgemmArgs a;
    conv_args(a, g, d, tab);
    a.A = wt + d.wt_off; a.B = dy; a.B2 = x_out; a.C = dx; a.E = addend;
    a.b_pro = PRO_NONE;
    int mt, nt;
    pick_tile(a.M, a.Ntot, &mt, &nt);
    SLV_CHECK_ARG(dispatch<MODE_CONV>(a, mt, nt, 1, (hipStream_t)stream) == 0, "no kernel for tile");
    SLV_LAUNCH_CHECK();
  }
  return 0;
}

size_t slv_conv_wgrad_ws_bytes(const int32_t* geom) {
  Geom g;
  if (read_geom(geom, g) != 0) return 0;
  const int mt = pick_mt(g.Cout);
  const int s = wgrad_splits(g, mt, mt >= 15 ? 1 : 2);
  return s > 1 ? sizeof(float) * (size_t)s * g.Cout * g.Cin * g.kt * g.kh * g.kw : 0;
}

int slv_conv_wgrad(const int32_t* geom, const float* dy, const float* x_out, const float* bwd5, int a_relu,
                   const float* x_in, const float* in_scale_shift, int in_relu, const int32_t* tab,
                   float* dw, void* ws, size_t ws_bytes, slv_stream_t stream) {
  Geom g;
  SLV_CHECK_ARG(read_geom(geom, g) == 0, "invalid geometry");
  SLV_CHECK_ARG(dy && x_in && tab && dw, "null pointer");
  SLV_CHECK_ARG(!bwd5 && !x_out, "the on-load BN-backward prologue was removed: materialise dXout with slv_bn_bwd_apply");
  (void)a_relu;
  const Desc d = fwd_desc(g);
  IgemmArgs a;
  memset(&a, 0, sizeof(a));
  const int taps = g.kt * g.kh * g.kw;
  a.tab = (const int2*)tab;
  a.tapd = (const int*)(tab + 2 * kpad(d.Kd));
  a.Cin = g.Cin; a.Ti = g.Ti; a.Hi = g.Hi; a.Wi = g.Wi; a.Cout = g.Cout; a.To = g.To; a.Ho = g.Ho; a.Wo = g.Wo;
  a.st = g.st; a.sh = g.sh; a.sw = g.sw; a.pt = g.pt; a.ph = g.ph; a.pw = g.pw;
  a.A = dy; a.a_pro = PRO_NONE;
  a.B = x_in; a.pb = in_scale_shift; a.b_pro = in_scale_shift ? PRO_ACT : PRO_NONE; a.b_relu = in_relu;
  a.A_bytes = a.A2_bytes = (unsigned)((size_t)g.Bn * g.Cout * g.To * g.Ho * g.Wo * 4);
  a.B_bytes = (unsigned)((size_t)g.Bn * g.Cin * g.Ti * g.Hi * g.Wi * 4);
  a.M = g.Cout; a.Kd = 0; a.Ntot = (long long)g.Cin * taps; a.ldc = g.Cin * taps;
  a.Ptot = (long long)g.Bn * g.To * g.Ho * g.Wo;
  SLV_CHECK_ARG(a.Ptot < (1LL << 31), "more than 2^31 output positions");
  a.dPout = make_fastdiv((unsigned)(g.To * g.Ho * g.Wo));
  a.dHoWo = make_fastdiv((unsigned)(g.Ho * g.Wo));
  a.dWo = make_fastdiv((unsigned)g.Wo);
  const int mt = pick_mt(a.M);
  const int nt = mt >= 15 ? 1 : 2;
  const int splits = wgrad_splits(g, mt, nt);
  const long long chunks = (a.Ptot + 15) / 16;
  a.chunks_per_split = (int)((chunks + splits - 1) / splits);
  const size_t nel = (size_t)g.Cout * g.Cin * taps;
  if (splits > 1) {
    SLV_CHECK_ARG(ws && ws_bytes >= sizeof(float) * nel * splits, "workspace too small");
    a.C = (float*)ws;
  } else {
    a.C = dw;
  }
  SLV_CHECK_ARG(dispatch<MODE_WGRAD>(a, mt, nt, splits, (hipStream_t)stream) == 0, "no kernel for tile");
  SLV_LAUNCH_CHECK();
  if (splits > 1) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((nel + 255) / 256 < 2048 ? (nel + 255) / 256 : 2048)),
                       dim3(256), 0, (hipStream_t)stream, (const float*)ws, dw, nel, splits);
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

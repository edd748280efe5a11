// C-ABI entry points of the implicit-GEMM convolution family (see igemm.hpp for the kernel).
// Replaces cuDNN conv3d/conv2d forward, backward-data and backward-weight as reached from the
// torchvision nets instantiated by /root/reference/model.py:95,114 and their autograd backward
// (main.py:301).
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
  if (g.kt <= 0 || g.kh <= 0 || g.kw <= 0 || g.kt > 15 || g.kh > 15 || g.kw > 15) return -1;
  if ((g.st != 1 && g.st != 2) || (g.sh != 1 && g.sh != 2) || (g.sw != 1 && g.sw != 2)) return -1;
  const int To = (g.Ti + 2 * g.pt - g.kt) / g.st + 1, Ho = (g.Hi + 2 * g.ph - g.kh) / g.sh + 1,
            Wo = (g.Wi + 2 * g.pw - g.kw) / g.sw + 1;
  if (To != g.To || Ho != g.Ho || Wo != g.Wo || To <= 0 || Ho <= 0 || Wo <= 0) return -1;
  if ((long long)g.Cin * g.Ti * g.Hi * g.Wi >= (1LL << 31) || (long long)g.Cout * g.To * g.Ho * g.Wo >= (1LL << 31))
    return -1;  // per-sample offsets are 32-bit in the tables
  if (g.Cin >= (1 << 19) || g.Cout >= (1 << 19)) return -1;
  return 0;
}

static void fill_common(IgemmArgs& a, const Geom& g) {
  memset(&a, 0, sizeof(a));
  a.Bn = g.Bn; a.Cin = g.Cin; a.Ti = g.Ti; a.Hi = g.Hi; a.Wi = g.Wi;
  a.Cout = g.Cout; a.To = g.To; a.Ho = g.Ho; a.Wo = g.Wo;
  a.st = g.st; a.sh = g.sh; a.sw = g.sw; a.pt = g.pt; a.ph = g.ph; a.pw = g.pw;
}

// column tile: NT=2 (128 columns) unless that leaves most of the 256 CUs idle
static int pick_nt(long long ncols, int nblkM_for_nt2, int mt) {
  if (mt >= 15) return 1;  // 240-row tiles keep 60 accumulators per wave; NT=2 would spill
  const long long blocks2 = ((ncols + 127) / 128) * nblkM_for_nt2;
  return blocks2 >= 384 ? 2 : 1;
}

template <int MODE>
static int dispatch(const IgemmArgs& a0, int mt, int nt, int splits, hipStream_t st) {
  IgemmArgs a = a0;
  const int bm = mt * 16, bn = nt * 64;
  a.nblkM = (a.M + bm - 1) / bm;
  a.nblkN = (int)((a.Ntot + bn - 1) / bn);
#define SLV_CASE(MT_, NT_) \
  if (mt == MT_ && nt == NT_) { launch_igemm<MODE, MT_, NT_>(a, splits, st); return 0; }
  SLV_CASE(4, 1) SLV_CASE(4, 2) SLV_CASE(8, 1) SLV_CASE(8, 2)
  SLV_CASE(9, 1) SLV_CASE(9, 2) SLV_CASE(15, 1)
#undef SLV_CASE
  return -1;
}

__global__ void wt_transform_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin,
                                    int taps) {
  const size_t n = (size_t)Cout * Cin * taps;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const size_t r = i / taps;
    const int ci = (int)(r % Cin), co = (int)(r / Cin);
    wt[((size_t)ci * Cout + co) * taps + tap] = w[i];
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

// number of int2 entries of the padded table: C*taps rounded up to 16, plus 16 invalid entries
int32_t slv_conv_table_len(const int32_t* geom, int dgrad) {
  Geom g;
  if (read_geom(geom, g) != 0) return -1;
  const int n = (dgrad ? g.Cout : g.Cin) * g.kt * g.kh * g.kw;
  return ((n + 15) / 16) * 16 + 16;
}

// table entry k=(c,tap): {offset, dt | dh<<4 | dw<<8 | c<<12}; pad entries: {0, -1} (y < 0 = invalid)
int slv_conv_table(const int32_t* geom, int dgrad, int32_t* tab_host_out) {
  Geom g;
  SLV_CHECK_ARG(read_geom(geom, g) == 0 && tab_host_out, "invalid geometry");
  const int taps = g.kt * g.kh * g.kw;
  const int C = dgrad ? g.Cout : g.Cin;
  const int total = ((C * taps + 15) / 16) * 16 + 16;
  for (int k = C * taps; k < total; ++k) {
    tab_host_out[2 * k] = 0;
    tab_host_out[2 * k + 1] = INT32_MIN;  // sign bit = invalid; channel field stays 0 (in range)
  }
  const int THWi = g.Ti * g.Hi * g.Wi, HWi = g.Hi * g.Wi, Pout = g.To * g.Ho * g.Wo;
  for (int c = 0; c < C; ++c)
    for (int dt = 0; dt < g.kt; ++dt)
      for (int dh = 0; dh < g.kh; ++dh)
        for (int dw = 0; dw < g.kw; ++dw) {
          const int k = c * taps + (dt * g.kh + dh) * g.kw + dw;
          tab_host_out[2 * k] = dgrad ? c * Pout : c * THWi + dt * HWi + dh * g.Wi + dw;
          tab_host_out[2 * k + 1] = dt | (dh << 4) | (dw << 8) | (c << 12);
        }
  return 0;
}

int32_t slv_conv_fwd_nblk(const int32_t* geom) {
  Geom g;
  if (read_geom(geom, g) != 0) return -1;
  const int mt = pick_mt(g.Cout);
  const long long P = (long long)g.Bn * g.To * g.Ho * g.Wo;
  const int nt = pick_nt(P, (g.Cout + mt * 16 - 1) / (mt * 16), mt);
  return (int32_t)((P + nt * 64 - 1) / (nt * 64));
}

int slv_conv_fwd(const int32_t* geom, const float* x, const float* w, const int32_t* tab,
                 const float* in_scale_shift, int in_relu, float* y, float* stat_sum, float* stat_sq,
                 slv_stream_t stream) {
  Geom g;
  SLV_CHECK_ARG(read_geom(geom, g) == 0, "invalid geometry");
  SLV_CHECK_ARG(x && w && tab && y, "null pointer");
  IgemmArgs a;
  fill_common(a, g);
  a.A = w; a.B = x; a.tab = (const int2*)tab; a.C = y;
  a.pb = in_scale_shift; a.b_pro = in_scale_shift ? PRO_ACT : PRO_NONE; a.b_relu = in_relu;
  a.stat_sum = stat_sum; a.stat_sq = stat_sq;
  a.M = g.Cout; a.Kd = g.Cin * g.kt * g.kh * g.kw;
  a.Ntot = (long long)g.Bn * g.To * g.Ho * g.Wo;
  const int mt = pick_mt(a.M);
  const int nt = pick_nt(a.Ntot, (a.M + mt * 16 - 1) / (mt * 16), mt);
  SLV_CHECK_ARG(dispatch<MODE_FWD>(a, mt, nt, 1, (hipStream_t)stream) == 0, "no kernel for tile");
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_conv_wt_transform(const float* w, float* wt, int Cout, int Cin, int taps, slv_stream_t stream) {
  SLV_CHECK_ARG(w && wt && Cout > 0 && Cin > 0 && taps > 0, "null pointer or empty shape");
  const size_t n = (size_t)Cout * Cin * taps;
  hipLaunchKernelGGL(wt_transform_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)),
                     dim3(256), 0, (hipStream_t)stream, w, wt, Cout, Cin, taps);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_conv_dgrad(const int32_t* geom, const float* dy, const float* x_out, const float* wt,
                   const int32_t* tab, const float* bwd5, int relu, float* dx, const float* addend,
                   slv_stream_t stream) {
  Geom g;
  SLV_CHECK_ARG(read_geom(geom, g) == 0, "invalid geometry");
  SLV_CHECK_ARG(dy && wt && tab && dx && (!bwd5 || x_out), "null pointer");
  IgemmArgs a;
  fill_common(a, g);
  a.A = wt; a.B = dy; a.B2 = x_out; a.tab = (const int2*)tab; a.C = dx; a.E = addend;
  a.pb = bwd5; a.b_pro = bwd5 ? PRO_BWD : PRO_NONE; a.b_relu = relu;
  a.M = g.Cin; a.Kd = g.Cout * g.kt * g.kh * g.kw;
  a.Ntot = (long long)g.Bn * g.Ti * g.Hi * g.Wi;
  const int mt = pick_mt(a.M);
  const int nt = pick_nt(a.Ntot, (a.M + mt * 16 - 1) / (mt * 16), mt);
  SLV_CHECK_ARG(dispatch<MODE_DGRAD>(a, mt, nt, 1, (hipStream_t)stream) == 0, "no kernel for tile");
  SLV_LAUNCH_CHECK();
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
  SLV_CHECK_ARG(dy && x_in && tab && dw && (!bwd5 || x_out), "null pointer");
  IgemmArgs a;
  fill_common(a, g);
  const int taps = g.kt * g.kh * g.kw;
  a.A = dy; a.A2 = x_out; a.pa = bwd5; a.a_pro = bwd5 ? PRO_BWD : PRO_NONE; a.a_relu = a_relu;
  a.B = x_in; a.pb = in_scale_shift; a.b_pro = in_scale_shift ? PRO_ACT : PRO_NONE; a.b_relu = in_relu;
  a.tab = (const int2*)tab;
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
  IgemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.B = B; a.bias = bias; a.C = C; a.M = M; a.Kd = K; a.Ntot = N; a.ldc = ldc;
  a.Hi = a.Wi = a.Ti = a.Ho = a.Wo = a.To = 1;
  const int mt = pick_mt(M);
  const int nt = pick_nt(N, (M + mt * 16 - 1) / (mt * 16), mt);
  SLV_CHECK_ARG(dispatch<MODE_GEMM>(a, mt, nt, 1, (hipStream_t)stream) == 0, "no kernel for tile");
  SLV_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

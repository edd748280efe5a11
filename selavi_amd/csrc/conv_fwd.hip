// C-ABI entry points of the implicit-GEMM convolution family -- gather tables, launch configurations, weight re-layouts, FORWARD conv
// (see igemm.hpp for the kernel, conv_common.hpp for geometry / tables / launch configurations).
// Replaces cuDNN conv3d/conv2d forward, backward-data and backward-weight as reached from the torchvision nets
// instantiated by /root/reference/model.py:95,114 and their autograd backward (main.py:301).
#include "conv_common.hpp"

namespace slv {

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

// The x3 weight image of one launch (conv_common.hpp: x3_image_floats): every 16-byte slot of the image is produced by one
// thread -- 8 consecutive k of one row, gathered from w (fp32, [Cout][Cin][taps]), cut into the three bf16 pieces
// (igemm3.hpp: split3) and stored to the three planes; padding (channels beyond C, rows beyond M, the odd half chunk) is
// written as zeros, so no memset.  transposed = 0: rows = Cout, gathered channel = ci (forward); 1: rows = Cin,
// gathered channel = co (backward data, taps = this parity class' taps).
struct SplitDesc {
  unsigned long long off;      // dword offset of this launch's image inside the buffer
  int M, Cg, Kd, ntaps, transposed;
  int taps[27];
};
struct SplitArgs {
  SplitDesc d[9];              // the forward image + up to 8 stride-parity classes of the backward data: ONE launch per layer
};
__device__ __forceinline__ void w_split_body(const float* __restrict__ w, unsigned* __restrict__ img, const SplitDesc& d, int Cin,
                                             int taps_all, size_t first = 0, size_t count = ~(size_t)0) {
  const int M = d.M, Cg = d.Cg, Kd = d.Kd, ntaps = d.ntaps, transposed = d.transposed;
  const int Mp = (M + 15) / 16 * 16, nch = (Kd + 31) / 32;
  size_t total = (size_t)nch * Mp * 4;
  if (count < total - first) total = first + count;        // (a job of the batched launch covers a RANGE of the image's slots)
  for (size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int slot = (int)(i & 3);
    const size_t rr = i >> 2;
    const int row = (int)(rr % Mp), ch = (int)(rr / Mp);
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = ch * 32 + slot * 8 + e;
      float v = 0.f;
      if (k < Kd && row < M) {
        const int grp = k >> 4, j = grp % ntaps, c = (grp / ntaps) * 16 + (k & 15);
        if (c < Cg) v = transposed ? w[((size_t)c * Cin + row) * taps_all + d.taps[j]] : w[((size_t)row * Cin + c) * taps_all + d.taps[j]];
      }
      const unsigned u = __float_as_uint(v);
      h[e] = u & 0xffff0000u;
      const float r1 = v - __uint_as_float(h[e]);
      m[e] = __float_as_uint(r1) & 0xffff0000u;
      l[e] = __float_as_uint(r1 - __uint_as_float(m[e]));
    }
    const size_t base = ((size_t)ch * 3 * Mp + row) * 16 + (size_t)((slot ^ ((-(row >> 2)) & 3)) << 2);   // dwords
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      img[base + e] = (h[2 * e] >> 16) | h[2 * e + 1];
      img[base + (size_t)Mp * 16 + e] = (m[2 * e] >> 16) | m[2 * e + 1];
      img[base + (size_t)Mp * 32 + e] = (l[2 * e] >> 16) | (l[2 * e + 1] & 0xffff0000u);
    }
  }
}
__global__ __launch_bounds__(256) void w_split_kernel(const float* __restrict__ w, unsigned* __restrict__ wf_img,
                                                      unsigned* __restrict__ wt_img, const SplitArgs args, int Cin, int taps_all) {
  const SplitDesc& d = args.d[blockIdx.y];
  w_split_body(w, (d.transposed ? wt_img : wf_img) + d.off, d, Cin, taps_all);
}
// The weight images of MANY layers in one launch (slv_conv_w_transform_jobs): a job = one image (a layer's forward image or
// one stride-parity class of its backward data); the table lives in device memory, built once per model and input shape
// (pointers to the parameters and to the persistent image buffers are stable), blockIdx.y = job.  ~70 launches of a
// training step become 2 (one per trunk).
struct SplitJob {
  const float* w;
  unsigned* img;              // the image's first dword (offset inside the layer's buffer already added)
  int Cin, taps_all;
  unsigned long long first, count;      // the 16-byte slots of the image this job makes (W_JOB_SLOTS at most: the images of
  SplitDesc d;                          // layer 4 are 30 x a layer-1 image -- equal jobs keep every CU busy to the end)
};
constexpr unsigned W_JOB_SLOTS = 16384, W_JOB_BLOCKS = 16;
__global__ __launch_bounds__(256) void w_split_jobs_kernel(const SplitJob* __restrict__ jobs) {
  const SplitJob& j = jobs[blockIdx.y];
  w_split_body(j.w, j.img, j.d, j.Cin, j.taps_all, (size_t)j.first, (size_t)j.count);
}
static bool split_desc(SplitDesc& o, const Desc& d, size_t off_floats, int transposed) {
  if (d.ntaps == 0 || d.Kd == 0 || d.ntaps > 27) return false;      // a parity class no tap reaches: no weights, no K steps
  o.off = off_floats;
  o.M = d.M; o.Cg = d.C; o.Kd = d.Kd; o.ntaps = d.ntaps; o.transposed = transposed;
  for (int j = 0; j < 27; ++j) o.taps[j] = j < d.ntaps ? d.taps[j] : 0;
  return true;
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

int32_t slv_conv_configs(const int32_t* geom, int op, int32_t* out, int32_t max_out) {
  Geom g;
  if (read_geom(geom, g) != 0 || !out || max_out <= 0 || op < 0 || op > 2) return -1;
  static const int tiles[16][3] = {{9, 2, 0}, {8, 2, 0}, {15, 1, 0}, {4, 2, 0}, {9, 1, 0}, {8, 1, 0}, {4, 1, 0},
                                   {8, 2, 1}, {6, 2, 1}, {4, 2, 1}, {9, 4, 0}, {8, 4, 0}, {4, 4, 0},
                                   {9, 3, 0}, {8, 3, 0}, {4, 3, 0}};
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
    const bool x3 = op != 2 && x3_enabled() && want_tap_major(op == 0 ? g.Cin : g.Cout);
    if (t[2] && x3) continue;            // split-operand kernels: 16 x 16 tiles only
    if (t[1] == 4 && !x3) continue;      // 256-column (8-wave) tiles: split-operand conv kernels only
    if (t[1] == 3 && !(op == 2 && x3_enabled())) continue;   // 192-column tiles: split-operand weight gradient only
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
  SLV_CHECK_ARG((dispatch<MODE_CONV, SUB_FWD>(a, mt, nt, sp, (hipStream_t)stream, c.mf) == 0), "no kernel for tile");
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
  return d.kord == KORD_TAP ? d.a_floats : 0;
}

size_t slv_conv_wt_elems(const int32_t* geom) {
  Geom g;
  if (read_geom(geom, g) != 0) return 0;
  Desc ds[8];
  const int n = dgrad_descs(g, ds);
  size_t t = 0;
  for (int i = 0; i < n; ++i) t += ds[i].a_floats;
  return t > 0 ? t : 1;
}

int slv_conv_w_transform(const int32_t* geom, const float* w, float* wf, float* wt, slv_stream_t stream) {
  Geom g;
  SLV_CHECK_ARG(read_geom(geom, g) == 0 && w && (wf || wt), "invalid geometry or null pointer");
  const Desc df = fwd_desc(g);
  if (df.kord != KORD_TAP) wf = nullptr;   // the forward conv of this layer reads w directly
  Desc ds[8];
  const int n = dgrad_descs(g, ds);
  hipStream_t st = (hipStream_t)stream;
  // split-operand launches (igemm3.hpp) read a three-plane bf16 image instead of the fp32 re-layout: one launch makes the
  // forward image and the images of every stride-parity class of the backward data
  {
    SplitArgs sa;
    int nd = 0;
    size_t maxtot = 0;
    auto tot = [](const Desc& d) { return (size_t)((d.Kd + 31) / 32) * ((d.M + 15) / 16 * 16) * 4; };
    if (wf && df.x3) {
      SLV_CHECK_ARG(df.ntaps <= 27, "more than 27 taps");
      if (split_desc(sa.d[nd], df, 0, 0)) { maxtot = tot(df); ++nd; }
    }
    if (wt && n > 0 && ds[0].x3)
      for (int i = 0; i < n; ++i) {
        SLV_CHECK_ARG(ds[i].ntaps <= 27, "more than 27 taps in one stride-parity class");
        if (split_desc(sa.d[nd], ds[i], ds[i].wt_off, 1)) { if (tot(ds[i]) > maxtot) maxtot = tot(ds[i]); ++nd; }
      }
    if (nd > 0) {
      const unsigned bx = (unsigned)((maxtot + 255) / 256 < 1024 ? (maxtot + 255) / 256 : 1024);
      hipLaunchKernelGGL(w_split_kernel, dim3(bx, nd), dim3(256), 0, st, w, (unsigned*)wf, (unsigned*)wt, sa, g.Cin,
                         g.kt * g.kh * g.kw);
      SLV_LAUNCH_CHECK();
    }
    if (df.x3) wf = nullptr;
    if (n > 0 && ds[0].x3) wt = nullptr;
  }
  if (!wf && !wt) return 0;
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

// ---- batched weight images: see SplitJob.  slv_conv_w_jobs writes the jobs of ONE layer (host memory, slv_conv_w_job_words()
// int32 words each; returns their number, 0 when the layer does not use split-operand images: the caller keeps
// slv_conv_w_transform for it); the caller concatenates the jobs of its layers, copies the table to the device once and
// launches slv_conv_w_transform_jobs every step.
int32_t slv_conv_w_job_words(void) { return (int32_t)(sizeof(SplitJob) / 4); }

int32_t slv_conv_w_jobs(const int32_t* geom, const float* w, float* wf, float* wt, int32_t* out, int32_t max_jobs) {
  Geom g;
  if (read_geom(geom, g) != 0 || !w || !out || max_jobs < 9) return -1;
  const Desc df = fwd_desc(g);
  Desc ds[8];
  const int n = dgrad_descs(g, ds);
  if (!(wf && df.x3) && !(wt && n > 0 && ds[0].x3)) return 0;
  if ((wf && df.kord == KORD_TAP && !df.x3) || (wt && n > 0 && !ds[0].x3)) return 0;      // mixed layouts: the per-layer entry point
  SplitJob* jobs = (SplitJob*)out;
  int nj = 0;
  const int taps_all = g.kt * g.kh * g.kw;
  auto emit = [&](const Desc& d, unsigned* img, int transposed) -> bool {      // the image in jobs of W_JOB_SLOTS slots
    SplitJob j;
    memset(&j, 0, sizeof(j));
    if (!split_desc(j.d, d, 0, transposed)) return true;
    j.w = w; j.img = img; j.Cin = g.Cin; j.taps_all = taps_all; j.d.off = 0;
    const unsigned long long total = (unsigned long long)((d.Kd + 31) / 32) * ((d.M + 15) / 16 * 16) * 4;
    for (unsigned long long f = 0; f < total; f += W_JOB_SLOTS) {
      if (nj >= max_jobs) return false;
      j.first = f;
      j.count = total - f < W_JOB_SLOTS ? total - f : W_JOB_SLOTS;
      jobs[nj++] = j;
    }
    return true;
  };
  if (wf && df.x3) {
    if (df.ntaps > 27 || !emit(df, (unsigned*)wf, 0)) return -1;
  }
  if (wt && n > 0 && ds[0].x3)
    for (int i = 0; i < n; ++i)
      if (ds[i].ntaps > 27 || !emit(ds[i], (unsigned*)wt + ds[i].wt_off, 1)) return -1;
  return nj;
}

int slv_conv_w_transform_jobs(const int32_t* jobs_dev, int32_t njobs, int32_t blocks_per_job, slv_stream_t stream) {
  SLV_CHECK_ARG(jobs_dev && njobs > 0 && njobs <= 65535 && blocks_per_job >= 0 && blocks_per_job <= 1024, "bad argument");
  if (blocks_per_job == 0) blocks_per_job = W_JOB_BLOCKS;      // (16 x 256 threads for <= 16 384 slots)
  hipLaunchKernelGGL(w_split_jobs_kernel, dim3((unsigned)blocks_per_job, (unsigned)njobs), dim3(256), 0, (hipStream_t)stream,
                     (const SplitJob*)jobs_dev);
  SLV_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

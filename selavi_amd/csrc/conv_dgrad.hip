// C-ABI entry points of the implicit-GEMM convolution family -- BACKWARD-DATA conv
// (see igemm.hpp for the kernel, conv_common.hpp for geometry / tables / launch configurations).
// Replaces cuDNN conv3d/conv2d forward, backward-data and backward-weight as reached from the torchvision nets
// instantiated by /root/reference/model.py:95,114 and their autograd backward (main.py:301).
#include "conv_common.hpp"

namespace slv {

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

}  // namespace slv

using namespace slv;

extern "C" {

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
    SLV_CHECK_ARG((dispatch<MODE_CONV, SUB_DGRAD>(a, pc[i].mt, pc[i].nt, sp, (hipStream_t)stream, pc[i].mf) == 0), "no kernel for tile");
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

}  // extern "C"

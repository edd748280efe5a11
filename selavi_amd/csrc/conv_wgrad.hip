// C-ABI entry points of the implicit-GEMM convolution family -- BACKWARD-WEIGHT conv
// (see igemm.hpp for the kernel, conv_common.hpp for geometry / tables / launch configurations).
// Replaces cuDNN conv3d/conv2d forward, backward-data and backward-weight as reached from the torchvision nets
// instantiated by /root/reference/model.py:95,114 and their autograd backward (main.py:301).
#include "conv_common.hpp"

using namespace slv;

extern "C" {

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

}  // extern "C"

// Eval-mode forward conv with BatchNorm folded into the weights (the Sinkhorn-Knopp feature pass,
// /root/reference/src/sk_utils.py:137-254: the model in eval mode over the whole dataset): instantiations of the
// split-operand implicit-GEMM kernel (igemm3.hpp) with the EPI_EVAL epilogue -- y = relu?(conv(x, w') + bias (+ residual)) --
// with three (exact split, 6 partial products) or two (3 partial products, opt-in) bf16 pieces per operand.  No prologue, no
// statistics, unsplit K.  A translation unit of its own: compile parallelism.
#include "conv_common.hpp"
#include "igemm3.hpp"
#include "../../include/selavi_hip.h"

namespace slv {

template <int NP>
static int launch_x3_eval(const IgemmArgs& a, int mt, int nt, hipStream_t st) {
#define SLV_CASE3(MT_, NT_) \
  if (mt == MT_ && nt == NT_) { launch_igemm3_eval<MT_, NT_, NP>(a, st); return 0; }
  SLV_CASE3(4, 1) SLV_CASE3(4, 2) SLV_CASE3(8, 1) SLV_CASE3(8, 2) SLV_CASE3(9, 1) SLV_CASE3(9, 2) SLV_CASE3(15, 1)
  SLV_CASE3(4, 4) SLV_CASE3(8, 4) SLV_CASE3(9, 4)
#undef SLV_CASE3
  return -1;
}

}  // namespace slv

extern "C" {

int32_t slv_conv_fwd_eval_ok(const int32_t* geom) {
  using namespace slv;
  Geom g;
  if (read_geom(geom, g) != 0) return 0;
  return (x3_enabled() && fwd_desc(g).kord == KORD_TAP) ? 1 : 0;
}

int slv_conv_fwd_eval(const int32_t* geom, const float* x, const float* wf, const int32_t* tab, const float* bias,
                      const float* res, int relu, int pieces, float* y, int32_t cfg, slv_stream_t stream) {
  using namespace slv;
  Geom g;
  SLV_CHECK_ARG(read_geom(geom, g) == 0, "invalid geometry");
  SLV_CHECK_ARG(x && wf && tab && y, "null pointer");
  SLV_CHECK_ARG(pieces == 2 || pieces == 3, "pieces: 3 (exact split) or 2");
  SLV_CHECK_ARG(x3_enabled(), "the eval-mode conv reads split-operand weight images: slv_conv_set_arithmetic(1)");
  Cfg c;
  SLV_CHECK_ARG(fwd_cfg(g, cfg, c) == 0, "invalid launch configuration");
  const Desc d = fwd_desc(g);
  SLV_CHECK_ARG(d.kord == KORD_TAP, "this layer has no split-operand image (the 3 / 1-channel stems): slv_conv_fwd");
  IgemmArgs a;
  conv_args(a, g, d, tab);
  a.A = wf; a.B = x; a.C = y;
  a.b_pro = PRO_NONE;
  a.bias = bias; a.E = res; a.epi_relu = relu;
  const int bm = c.mt * 16, bn = c.nt * 64;
  a.nblkM = (a.M + bm - 1) / bm;
  a.nblkN = (int)((a.Ntot + bn - 1) / bn);
  if (a.nblkN == 0) return 0;
  const int rc = pieces == 3 ? launch_x3_eval<3>(a, c.mt, c.nt, (hipStream_t)stream) : launch_x3_eval<2>(a, c.mt, c.nt, (hipStream_t)stream);
  SLV_CHECK_ARG(rc == 0, "no kernel for tile");
  SLV_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

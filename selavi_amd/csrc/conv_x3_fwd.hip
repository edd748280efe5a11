// Instantiations of the split-operand implicit-GEMM kernel (igemm3.hpp) for the forward conv (a translation unit of its own:
// compile parallelism).  Reference semantics: torchvision Conv3d/Conv2d forward conv as reached from
// /root/reference/model.py:95,114 and main.py:284-301.
#include "conv_common.hpp"
#include "igemm3.hpp"

namespace slv {

// Conv arithmetic of the fp32 path: 1 = split operands on the bf16 matrix cores (igemm3.hpp), 0 = the native fp32 MFMA
// kernels (igemm.hpp).  Default from SELAVI_CONV_X3 (on); slv_conv_set_arithmetic switches it at run time -- weight images
// and tables made under one setting must not be used under the other (the host side drops its plans).
static int g_x3 = -1;
int x3_enabled() {
  if (g_x3 < 0) {
    const char* e = getenv("SELAVI_CONV_X3");
    g_x3 = e ? (atoi(e) != 0) : 1;
  }
  return g_x3;
}
void x3_set(int on) { g_x3 = on ? 1 : 0; }

int launch_x3_fwd(const IgemmArgs& a, int mt, int nt, int splits, hipStream_t st) {
#define SLV_CASE3(MT_, NT_) \
  if (mt == MT_ && nt == NT_) { launch_igemm3<MT_, NT_, SUB_FWD>(a, splits, st); return 0; }
  SLV_CASE3(4, 1) SLV_CASE3(4, 2) SLV_CASE3(8, 1) SLV_CASE3(8, 2) SLV_CASE3(9, 1) SLV_CASE3(9, 2) SLV_CASE3(15, 1)
  SLV_CASE3(4, 4) SLV_CASE3(8, 4) SLV_CASE3(9, 4)
#undef SLV_CASE3
  return -1;
}

}  // namespace slv

extern "C" {
int slv_conv_set_arithmetic(int split_bf16x3) {
  slv::x3_set(split_bf16x3);
  return 0;
}
int32_t slv_conv_get_arithmetic(void) { return slv::x3_enabled(); }
}

#ifdef SLV_X3_TRACE
extern "C" int slv_debug_x3_trace(void* out_host, size_t bytes) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(slv::slv_x3_trace_buf), bytes < sizeof(slv::slv_x3_trace_buf) ? bytes : sizeof(slv::slv_x3_trace_buf)) == hipSuccess ? 0 : -1;
}
#endif

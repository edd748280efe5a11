// Instantiations of the split-operand weight-gradient kernel (igemm3.hpp: igemm3_wgrad_kernel), a translation unit of its
// own (compile parallelism).  Reference semantics: torchvision Conv3d/Conv2d backward-weight as reached from
// /root/reference/model.py:95,114 and main.py:301.
#include "conv_common.hpp"
#include "igemm3.hpp"

namespace slv {

int launch_x3_wgrad(const IgemmArgs& a, int mt, int nt, int splits, bool vec_a, hipStream_t st) {
#define SLV_CASE3(MT_, NT_) \
  if (mt == MT_ && nt == NT_) { launch_igemm3_wgrad<MT_, NT_>(a, splits, vec_a, st); return 0; }
  SLV_CASE3(4, 1) SLV_CASE3(4, 2) SLV_CASE3(8, 1) SLV_CASE3(8, 2) SLV_CASE3(9, 1) SLV_CASE3(9, 2) SLV_CASE3(15, 1)
  SLV_CASE3(4, 3) SLV_CASE3(8, 3) SLV_CASE3(9, 3)
#undef SLV_CASE3
  return -1;
}

}  // namespace slv

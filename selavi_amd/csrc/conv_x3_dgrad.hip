// Instantiations of the split-operand implicit-GEMM kernel (igemm3.hpp) for the backward-data conv (a translation unit of its own:
// compile parallelism).  Reference semantics: torchvision Conv3d/Conv2d backward-data conv as reached from
// /root/reference/model.py:95,114 and main.py:284-301.
#include "conv_common.hpp"
#include "igemm3.hpp"

namespace slv {

int launch_x3_dgrad(const IgemmArgs& a, int mt, int nt, int splits, hipStream_t st) {
#define SLV_CASE3(MT_, NT_) \
  if (mt == MT_ && nt == NT_) { launch_igemm3<MT_, NT_, SUB_DGRAD>(a, splits, st); return 0; }
  SLV_CASE3(4, 1) SLV_CASE3(4, 2) SLV_CASE3(8, 1) SLV_CASE3(8, 2) SLV_CASE3(9, 1) SLV_CASE3(9, 2) SLV_CASE3(15, 1)
  SLV_CASE3(4, 4) SLV_CASE3(8, 4) SLV_CASE3(9, 4)
#undef SLV_CASE3
  return -1;
}

}  // namespace slv

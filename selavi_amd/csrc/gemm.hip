// Dense NT GEMM entry point (heads applied to the SK feature bank, sk_utils.py:309-312) on the same MFMA core.
#include "conv_common.hpp"

using namespace slv;

extern "C" {

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

// Weight gradient of the stride-1 temporal (3,1,1) convs of the 16-bit path (csrc/wgrad_cl16_t.hip) that ALSO yields the
// BatchNorm-backward sums of the layer the conv reads -- the spatial half of the same Conv2Plus1D (/root/reference/
// model.py:147-176: Conv3d (1,3,3) - BatchNorm3d - ReLU - Conv3d (3,1,1); main.py:296-299 runs the backward) -- so that
// layer's reduce pass over the gradient and the activation (2 x 4.1 GB per conv at 128 clips x 32 frames) disappears.
//
// With y the spatial conv's raw output, a = relu(y s + h) the temporal conv's input, m = [y s + h > 0], dY the temporal
// conv's output gradient and W its weights:
//     g[p][c]  = sum_{co,dt} W[co][c][dt] dY[p - dt][co]                 (backward data: the gradient w.r.t. a)
//     S1[c]    = sum_p g m          = sum_{co,dt} W[co][c][dt] G1[co][c][dt],     G1 = sum_p dY[p - dt][co] m[p][c]
//     S2[c]    = sum_p g m y        = sum_{co,dt} W[co][c][dt] G2[co][c][dt],     G2 = sum_p dY[p - dt][co] (m y)[p][c]
//     dW[co][c][dt] = sum_p dY[p - dt][co] a[p][c] = s[c] G2 + h[c] G1
// i.e. both BatchNorm sums are contractions of the temporal weights with two weight-gradient-shaped tensors, and the
// weight gradient itself is a combination of the same two.  G2 and G1 come from the column-order kernel itself
// (csrc/wgrad_cl16_t.hip, PRO 2): every (tile, K slice) is taken by TWO workgroups, consecutive units = the same XCD's L2,
// whose staging writes (m y) resp. m as bf16 (both exact) instead of relu(y s + h); they walk the same rows at the same
// time, so dY and y come from memory once and from L2 the second time.  (A single 8-wave workgroup forming both kinds from
// one staged tile was tried first: its two tiles per frame take 107 KB of LDS = one workgroup per CU, and without a second
// workgroup to fill a step's load -> store -> barrier chain it ran 3.70 ms against 1.24 ms of the plain kernel.)
// cl16_wgrad_t2_finish_kernel then sums the K slices' partials, writes dW and the two sums per channel:
//     part[c] = { S1, (S2 - mean S1) invstd }   = { sum g', sum g' xhat }: what slv_cl16_bn_bwd_reduce would have produced
// (from the unrounded g: the reduce pass sums the bf16-rounded values the backward-data conv stored).
#include "cl16.hpp"

namespace slv {

// sum of the K slices' partials, both kinds: gsum[kind][row][col], row < Cout, col < 3 * Cin_p
__global__ __launch_bounds__(256) void cl16_wgrad_t2_sum_kernel(const float* __restrict__ part, float* __restrict__ gsum,
                                                               int Cout, unsigned ncols, int slices, size_t slice_stride,
                                                               size_t kind_stride) {
  const unsigned col = blockIdx.x * 256 + threadIdx.x, co = blockIdx.y, kind = blockIdx.z;
  if (col >= ncols) return;
  const float* p = part + (size_t)kind * kind_stride + (size_t)co * ncols + col;
  float s4[4] = {0.f, 0.f, 0.f, 0.f};
  int i = 0;
  for (; i + 4 <= slices; i += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) s4[u] += p[(size_t)(i + u) * slice_stride];
  }
  for (; i < slices; ++i) s4[i & 3] += p[(size_t)i * slice_stride];
  gsum[((size_t)kind * Cout + co) * ncols + col] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

// one workgroup per input channel c: dW[co][c][dt] = s G2 + h G1; part[c] = { S1, (S2 - mean S1) invstd } with
// S1 = sum W G1, S2 = sum W G2 over (co, dt), W = the bf16-rounded master weights (what the backward-data conv multiplied by)
__global__ __launch_bounds__(256) void cl16_wgrad_t2_finish_kernel(const float* __restrict__ gsum, const float* __restrict__ w,
                                                                  const float* __restrict__ in_ss, const float* __restrict__ mi,
                                                                  float* __restrict__ dw, float* __restrict__ bn_part, int Cout,
                                                                  int Cin, int Cin_p) {
  __shared__ float red[2][256];
  const int c = blockIdx.x, tid = threadIdx.x;
  const unsigned ncols = 3u * Cin_p;
  const float s = in_ss[c], h = in_ss[Cin + c];
  float s1 = 0.f, s2 = 0.f;
  for (int j = tid; j < Cout * 3; j += 256) {
    const int co = j / 3, dt = j - co * 3;
    const float g2 = gsum[(size_t)co * ncols + (size_t)dt * Cin_p + c];
    const float g1 = gsum[((size_t)Cout + co) * ncols + (size_t)dt * Cin_p + c];
    const size_t o = ((size_t)co * Cin + c) * 3 + dt;
    dw[o] = __builtin_fmaf(s, g2, h * g1);
    const float wb = bf2f(f2bf(w[o]));
    s1 = __builtin_fmaf(wb, g1, s1);
    s2 = __builtin_fmaf(wb, g2, s2);
  }
  red[0][tid] = s1;
  red[1][tid] = s2;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {                             // fixed tree: deterministic
    if (tid < st) {
      red[0][tid] += red[0][tid + st];
      red[1][tid] += red[1][tid + st];
    }
    __syncthreads();
  }
  if (tid == 0) {
    const float S1 = red[0][0], S2 = red[1][0];
    bn_part[2 * c] = S1;
    bn_part[2 * c + 1] = (S2 - mi[c] * S1) * mi[Cin + c];
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
static bool t2_enabled() {            // (read per plan, not cached: tests switch it per case)
  const char* e = getenv("SELAVI_CL16_WGT2");
  return !(e && e[0] == '0');
}

// the shapes wgrad_t_plan takes, one workgroup of 8 waves per CU
bool wgrad_t2_plan(int N, int T, int H, int W, int Cin_p, int Cin, int Cout_p, int kt, int kh, int kw, int st, int sh, int sw,
                   int pt, int ph, int pw, int To, int Ho, int Wo, int* wm, int* nc, ClWgradT* out) {
  if (!t2_enabled()) return false;
  if (!wgrad_t_plan(N, T, H, W, Cin_p, Cin, Cout_p, kt, kh, kw, st, sh, sw, pt, ph, pw, To, Ho, Wo, wm, nc, out)) return false;
  // SELAVI_CL16_WGT2=all: also the layers only the two-workgroup column kernel takes (slower than the reduce pass it saves)
  const char* ea = getenv("SELAVI_CL16_WGT2");
  const bool all = ea && ea[0] == 'a';
  if (!all && !wgrad_tacc_applies(*out)) return false;
  ClWgradT& g = *out;                       // one round of resident workgroups with two per unit: half the K slices
  const long long steps = (long long)N * g.PB * (T + 1);
  long long ksl = (g.kslices + 1) / 2;
  if (ksl < 1) ksl = 1;
  g.sper = (int)((steps + ksl - 1) / ksl);
  g.kslices = (int)((steps + g.sper - 1) / g.sper);
  return true;
}

// workspace: [2 kinds][slices][rows][3 Cin_p] partials, then [2][Cout_p rows][3 Cin_p] sums (slices: the K slices of the
// column-order kernel, or the workgroups of the accumulator-resident kernel of the layer-1 shape: csrc/wgrad_cl16_tacc.hip)
size_t wgrad_t2_ws_bytes(const ClWgradT& g, int wm) {
  const size_t rows = (size_t)g.mtiles * wm * 32;
  const size_t slices = wgrad_tacc_applies(g) ? (size_t)wgrad_tacc_blocks() : (size_t)g.kslices;
  return (2 * slices * rows + 2 * rows) * 3 * g.Cin_p * sizeof(float);
}

int wgrad_t2_launch(const ClWgradT& g, int wm, int nc, const void* dy, const void* x, const float* in_ss, const float* mi,
                    const float* w, float* dw, float* bn_part, int Cout, float* ws, hipStream_t st) {
  const size_t rows = (size_t)g.mtiles * wm * 32, ncols = (size_t)3 * g.Cin_p;
  const bool tacc = wgrad_tacc_applies(g);
  const int slices = tacc ? wgrad_tacc_blocks() : g.kslices;
  const size_t slice_stride = rows * ncols, kind_stride = (size_t)slices * slice_stride;
  float* gsum = ws + 2 * kind_stride;
  int rc;
  if (tacc) {
    rc = wgrad_tacc_launch(g, dy, x, in_ss, ws, kind_stride, st);
  } else {
    wgrad_t_launch_dual(g, wm, nc, dy, x, in_ss, ws, kind_stride, st);
    rc = launch_check("slv_cl16_wgrad_bnr");
  }
  if (rc) return rc;
  hipLaunchKernelGGL(cl16_wgrad_t2_sum_kernel, dim3((unsigned)((ncols + 255) / 256), Cout, 2), dim3(256), 0, st, ws, gsum, Cout,
                     (unsigned)ncols, slices, slice_stride, kind_stride);
  rc = launch_check("slv_cl16_wgrad_bnr");
  if (rc) return rc;
  hipLaunchKernelGGL(cl16_wgrad_t2_finish_kernel, dim3(g.Cin), dim3(256), 0, st, gsum, w, in_ss, mi, dw, bn_part, Cout, g.Cin,
                     g.Cin_p);
  return launch_check("slv_cl16_wgrad_bnr");
}

}  // namespace slv

// First kernel of the 16-bit MFMA path (BASELINE configs[4], main.py:151 --use_fp16): a forward convolution on bf16
// CHANNELS-LAST activations with fp32 accumulation on v_mfma_f32_16x16x32_bf16 and a fused per-channel affine
// (eval-mode BatchNorm) + residual + ReLU epilogue.
//
// Why another layout: a 16-bit MFMA operand is 8 consecutive k per lane.  With N,C,T,H,W activations k (= channel)
// has stride T*H*W, so every fragment would be 8 scalar gathers; with N,T,H,W,C it is one 16-byte load and the LDS
// image needs no transpose.  On the hot layer (Conv3d 64->144 (1,3,3), 16 clips) this kernel takes 0.22 ms against
// 1.24 ms for the fp32 kernel (tools/proto/bf16_conv133.hip measured 601 TFLOP/s with the first, untuned version).
//
// Implicit GEMM  D[cout][pos] = sum_{tap, c} W[cout][tap][c] * X[pos*stride + tap - pad][c]:
//   block = 4 waves = (MT*16) couts x 128 output positions, wave = (MT*16) x 32, K-step = 32 channels of one tap;
//   weights pre-laid-out [tap][Cin_p/32][Mrows][32] (zero padded), activations [P][Cin_p] with Cin_p % 32 == 0;
//   register-staged double buffer, one barrier per K-step, LDS rows of 32 bf16 padded to 80 bytes (conflict-free
//   ds_read_b128 fragments); padding positions are buffer loads with an out-of-range offset (return 0).
//   Output [P_out][Cout_p] bf16: channels >= Cout are written as zeros so the next layer can read whole 32-channel
//   K-steps.
#include "common.hpp"
#include "../../include/selavi_hip.h"

namespace slv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ClGeom {                       // int32 x 20, mirrored by selavi_amd/ops16.py
  int N, Ti, Hi, Wi, Cin_p;           // input  [N][Ti][Hi][Wi][Cin_p]
  int Cout, Cout_p, To, Ho, Wo;       // output [N][To][Ho][Wo][Cout_p]
  int kt, kh, kw, st, sh, sw, pt, ph, pw;
  int Mrows;                          // rows of the weight layout: Cout rounded up to the block's M tile
};

constexpr int CL_BN = 128, CL_ROWB = 64;
// LDS image: rows of 32 bf16 = 64 bytes, unpadded; the 16-byte slot of k-group q in row r is q ^ swz(r).  ds_read_b128
// is serviced in lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md): with fragment lanes
// (row = lane & 15, k-group = lane >> 4) a group holds rows 0-3 and 12-15 of one k-group and rows 4-11 of its
// neighbour, and NO row padding separates them (the 80-byte rows of the first version measured 49 % conflict
// cycles).  swz(r) = (-(r >> 2)) & 3 makes the 16 slots of every group distinct.
__device__ __forceinline__ int cl_swz(int row) { return (-(row >> 2)) & 3; }

__device__ __forceinline__ unsigned short f2bf(float f) {             // round to nearest even (finite inputs)
  unsigned int u = __float_as_uint(f);
  u += 0x7FFF + ((u >> 16) & 1);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }

template <int MT>
__global__ __launch_bounds__(256, MT >= 8 ? 3 : 4) void conv_cl16_kernel(const unsigned short* __restrict__ x,
                                                           const unsigned short* __restrict__ wl,
                                                           unsigned short* __restrict__ y,
                                                           const float* __restrict__ scale_shift,   // [2][Cout] or null
                                                           const unsigned short* __restrict__ res,  // [P_out][Cout_p] or null
                                                           int relu, ClGeom g) {
  constexpr int BM = MT * 16;
  constexpr int APIECES = BM * 4, AITER = (APIECES + 255) / 256;
  constexpr int OROW = BM * 2 + 16;                   // bytes per position row of the transposed output tile (+16: banks)
  constexpr int STAGE = (BM + CL_BN) * CL_ROWB;
  constexpr int LDSB = 2 * STAGE > CL_BN * OROW + 2 * BM * 4 ? 2 * STAGE : CL_BN * OROW + 2 * BM * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDSB];
  unsigned char(*lds)[STAGE] = (unsigned char(*)[STAGE])lds_raw;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned P = (unsigned)g.N * g.To * g.Ho * g.Wo;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)x, 0, (int)((unsigned)g.N * g.Ti * g.Hi * g.Wi * g.Cin_p * 2u), 0x00020000);
  const int m0 = blockIdx.y * BM;
  // ---- activation rows of this thread: 128 rows x 4 pieces of 16 B, 2 per thread
  int bt[2], bh[2], bw[2];
  unsigned bbase[2];
  const int piece = tid & 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned p = blockIdx.x * CL_BN + (tid >> 2) + 64 * i;
    unsigned q = p;
    const int wo = q % g.Wo; q /= g.Wo;
    const int ho = q % g.Ho; q /= g.Ho;
    const int to = q % g.To; q /= g.To;      // q = clip
    bt[i] = to * g.st - g.pt;
    bh[i] = ho * g.sh - g.ph;
    bw[i] = wo * g.sw - g.pw;
    bbase[i] = p < P ? (((q * g.Ti + bt[i]) * g.Hi + bh[i]) * g.Wi + bw[i]) * (unsigned)(g.Cin_p * 2) + piece * 16u
                     : 0xFFFFFFF0u;          // (wraps for negative coordinates; only used when the tap is valid)
    if (p >= P) bt[i] = -(1 << 20);
  }
  const int kcs = g.Cin_p >> 5, ksteps = g.kt * g.kh * g.kw * kcs;
  u32x4 ra[AITER], rb[2];
  int tap = 0, kc = 0, dt = 0, dh = 0, dw = 0;                     // K-step being LOADED
  auto gload = [&]() __attribute__((always_inline)) {
    const unsigned toff = (unsigned)(((dt * g.Hi + dh) * g.Wi + dw) * g.Cin_p * 2 + kc * 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = (unsigned)(bt[i] + dt) < (unsigned)g.Ti && (unsigned)(bh[i] + dh) < (unsigned)g.Hi &&
                      (unsigned)(bw[i] + dw) < (unsigned)g.Wi;
      rb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? bbase[i] + toff : 0xFFFFFFF0u, 0, 0));
    }
    const unsigned short* ws = wl + ((size_t)(tap * kcs + kc) * g.Mrows + m0) * 32;
#pragma unroll
    for (int i = 0; i < AITER; ++i) {
      const int pc = tid + 256 * i;
      if (pc < APIECES) ra[i] = *(const u32x4*)(ws + pc * 8);
    }
    if (++kc == kcs) {                                              // advance to the next K-step
      kc = 0;
      ++tap;
      if (++dw == g.kw) {
        dw = 0;
        if (++dh == g.kh) {
          dh = 0;
          ++dt;
        }
      }
    }
  };
  auto lstore = [&](int buf) __attribute__((always_inline)) {
    unsigned char* A = lds[buf];
    unsigned char* B = A + BM * CL_ROWB;
#pragma unroll
    for (int i = 0; i < AITER; ++i) {
      const int pc = tid + 256 * i;
      if (pc < APIECES) *(u32x4*)(A + (pc >> 2) * CL_ROWB + (((pc & 3) ^ cl_swz(pc >> 2)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *(u32x4*)(B + ((tid >> 2) + 64 * i) * CL_ROWB + ((piece ^ cl_swz(tid >> 2)) << 4)) = rb[i];
  };
  f32x4 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fk = lane >> 4, fsw = (fk ^ cl_swz(fr)) << 4;   // rows i*16+fr, wave*32+j*16+fr: same swz
  gload();
  lstore(0);
  __syncthreads();
  for (int s = 0; s < ksteps; ++s) {
    if (s + 1 < ksteps) gload();
    const unsigned char* A = lds[s & 1];
    const unsigned char* B = A + BM * CL_ROWB;
    bf16x8 b[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = *(const bf16x8*)(B + (wave * 32 + j * 16 + fr) * CL_ROWB + fsw);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const bf16x8 a = *(const bf16x8*)(A + (i * 16 + fr) * CL_ROWB + fsw);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[j], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < ksteps) lstore((s + 1) & 1);
    __syncthreads();
  }
  // ---- epilogue.  C/D: col = lane & 15 (position), rows (lane >> 4) * 4 + r (cout): 4 consecutive couts per lane, i.e.
  // 8-byte pieces 320 bytes apart across lanes -- stored like that they cost a quarter of the forward (ablation).  The
  // tile is transposed through LDS instead: [position][cout] rows, then 16-byte stores that run along a position's
  // channels (288 contiguous bytes per row for the 144-channel tile).  scale / shift also come through LDS.
  __syncthreads();                                      // the K loop's last fragment reads are done
  unsigned char* ot = lds_raw;                          // [CL_BN][OROW]
  float* ssl = (float*)(lds_raw + CL_BN * OROW);        // [2][BM]
  if (scale_shift) {
    for (int i = tid; i < 2 * BM; i += 256) {
      const int c = m0 + (i % BM);
      ssl[i] = c < g.Cout ? scale_shift[(i / BM) * g.Cout + c] : 0.f;
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int co = m0 + i * 16 + fk * 4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (scale_shift) {
      sc = *(const f32x4*)(ssl + i * 16 + fk * 4);
      sh = *(const f32x4*)(ssl + BM + i * 16 + fk * 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pl = wave * 32 + j * 16 + fr;             // position inside the block
      const unsigned p = blockIdx.x * CL_BN + pl;
      uint2 rr = make_uint2(0u, 0u);
      if (res && p < P && co < g.Cout_p) rr = *(const uint2*)(res + (size_t)p * g.Cout_p + co);
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = acc[i][j][r] * sc[r] + sh[r];
        if (res) t += bf2f((unsigned short)((r < 2 ? rr.x : rr.y) >> ((r & 1) * 16)));
        if (relu) t = fmaxf(t, 0.f);
        v[r] = (co + r < g.Cout) ? t : 0.f;                         // padding channels stay zero
      }
      *(uint2*)(ot + pl * OROW + (i * 16 + fk * 4) * 2) =
          make_uint2(f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16), f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16));
    }
  }
  __syncthreads();
  // rows of this block in the output: channels [m0, m0 + BM) clipped to Cout_p; the last M block also zero-fills the
  // padding channels no M tile covers (Mrows < Cout_p)
  const int c_lo = m0, c_hi = min(m0 + BM, g.Cout_p);
  const int c_end = (blockIdx.y == gridDim.y - 1) ? g.Cout_p : c_hi;
  const int pieces = (c_end - c_lo) >> 3;                 // 16-byte pieces per position (channel counts are multiples of 8)
  for (int idx = tid; idx < CL_BN * pieces; idx += 256) {
    const int pl = idx / pieces, pc = idx - pl * pieces;
    const unsigned p = blockIdx.x * CL_BN + pl;
    if (p >= P) continue;
    u32x4 val = {0u, 0u, 0u, 0u};
    if (c_lo + pc * 8 < c_hi) val = *(const u32x4*)(ot + pl * OROW + pc * 16);
    *(u32x4*)(y + (size_t)p * g.Cout_p + c_lo + pc * 8) = val;
  }
}

// fp32 N,C,T,H,W -> bf16 N,T,H,W,Cp (channels >= C zero): one thread per (position, 8-channel piece)
__global__ __launch_bounds__(256) void to_cl16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int C,
                                                      int Cp, unsigned S /* T*H*W */, unsigned total /* N*S*Cp/8 */) {
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const unsigned pieces = Cp >> 3, pc = idx % pieces, pos = idx / pieces, n = pos / S, s = pos - n * S;
  unsigned short v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned c = pc * 8 + i;
    v[i] = c < (unsigned)C ? f2bf(x[((size_t)n * C + c) * S + s]) : (unsigned short)0;
  }
  u32x4 o = {v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16),
             v[6] | ((unsigned)v[7] << 16)};
  *(u32x4*)(y + (size_t)idx * 8) = o;
}


// Stem input: fp32 N,C,T,H,W -> bf16 [N][T][H][Wo][32] where the 32 "channels" of an output column wo are the
// kw x C input values of its receptive row (index dw*C + c, zero beyond kw*C and outside the image).  A (1,kh,kw)
// stem conv over C = 3 (or 1) channels then runs as a (1,kh,1) conv over 32 channels: kh K-steps instead of kh*kw with
// 3 of 32 channels used, and the converted input is half the size of a 32-channel padded copy.
__global__ __launch_bounds__(256) void to_cl16_wpatch_kernel(const float* __restrict__ x, unsigned short* __restrict__ y,
                                                             int C, unsigned TH, int W, int Wo, int kw, int sw, int pw,
                                                             unsigned total /* N*TH*Wo*4 pieces */) {
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const unsigned pc = idx & 3, q = idx >> 2, wo = q % Wo, row = q / Wo, n = row / TH, th = row - n * TH;
  unsigned short v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = pc * 8 + i, dw = k / C, c = k - dw * C, w = (int)wo * sw - pw + dw;
    float f = 0.f;
    if (dw < kw && (unsigned)w < (unsigned)W) f = x[(((size_t)n * C + c) * TH + th) * W + w];
    v[i] = f2bf(f);
  }
  u32x4 o = {v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16),
             v[6] | ((unsigned)v[7] << 16)};
  *(u32x4*)(y + (size_t)idx * 8) = o;
}

// MaxPool2d(3, stride 2, pad 1) on bf16 channels-last [N][H][W][Cp] (audio trunk, model.py:114 -> torchvision ResNet):
// one thread per (output position, 8-channel piece)
__global__ __launch_bounds__(256) void maxpool_cl16_kernel(const unsigned short* __restrict__ x,
                                                           unsigned short* __restrict__ y, int H, int W, int Ho, int Wo,
                                                           int Cp, unsigned total) {
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const unsigned pieces = Cp >> 3, pc = idx % pieces;
  unsigned q = idx / pieces;
  const int wo = q % Wo; q /= Wo;
  const int ho = q % Ho; q /= Ho;                 // q = image
  float m[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) m[i] = -INFINITY;
  for (int dh = 0; dh < 3; ++dh)
    for (int dw = 0; dw < 3; ++dw) {
      const int h = ho * 2 - 1 + dh, w = wo * 2 - 1 + dw;
      if ((unsigned)h >= (unsigned)H || (unsigned)w >= (unsigned)W) continue;
      const u32x4 v = *(const u32x4*)(x + (((size_t)q * H + h) * W + w) * Cp + pc * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        m[2 * i] = fmaxf(m[2 * i], bf2f((unsigned short)(v[i] & 0xFFFF)));
        m[2 * i + 1] = fmaxf(m[2 * i + 1], bf2f((unsigned short)(v[i] >> 16)));
      }
    }
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = f2bf(m[2 * i]) | ((unsigned)f2bf(m[2 * i + 1]) << 16);
  *(u32x4*)(y + (size_t)idx * 8) = o;
}

// AdaptiveAvgPool(1) + flatten: bf16 [N][S][Cp] -> fp32 [N][C]; one wave per (clip, 64 channels), fixed order
__global__ __launch_bounds__(64) void avgpool_cl16_kernel(const unsigned short* __restrict__ x, float* __restrict__ y,
                                                          int S, int C, int Cp) {
  const int n = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  float acc = 0.f;
  for (int s = 0; s < S; ++s) acc += bf2f(x[((size_t)n * S + s) * Cp + c]);
  y[(size_t)n * C + c] = acc / (float)S;
}

}  // namespace slv

extern "C" {

// geom: 20 int32 (ClGeom).  MT (16-row tiles per block) follows from geom.Mrows: the Python side picks it.
int slv_conv_cl16_fwd(const int32_t* geom, int mt, const void* x_bf16, const void* w_layout_bf16, void* y_bf16,
                      const float* scale_shift, const void* res_bf16, int relu, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(geom && x_bf16 && w_layout_bf16 && y_bf16, "null pointer");
  ClGeom g;
  memcpy(&g, geom, sizeof(g));
  SLV_CHECK_ARG(g.N > 0 && g.Cin_p > 0 && (g.Cin_p & 31) == 0 && g.Cout > 0 && g.Cout_p >= g.Cout && (g.Cout_p & 7) == 0,
                "channel counts (Cin_p % 32, Cout_p % 8)");
  SLV_CHECK_ARG(g.kt > 0 && g.kh > 0 && g.kw > 0 && g.st > 0 && g.sh > 0 && g.sw > 0, "kernel / stride");
  SLV_CHECK_ARG(g.To == (g.Ti + 2 * g.pt - g.kt) / g.st + 1 && g.Ho == (g.Hi + 2 * g.ph - g.kh) / g.sh + 1 &&
                    g.Wo == (g.Wi + 2 * g.pw - g.kw) / g.sw + 1 && g.To > 0 && g.Ho > 0 && g.Wo > 0,
                "output extent does not match the geometry");
  SLV_CHECK_ARG((long long)g.N * g.Ti * g.Hi * g.Wi * g.Cin_p * 2 < 0xFFFFFFF0LL &&
                    (long long)g.N * g.To * g.Ho * g.Wo * g.Cout_p * 2 < 0xFFFFFFF0LL,
                "tensor beyond the 32-bit buffer range");
  SLV_CHECK_ARG(mt == 4 || mt == 8 || mt == 9, "M tile (4, 8 or 9 x 16 rows)");
  SLV_CHECK_ARG(g.Mrows % (16 * mt) == 0 && g.Mrows >= g.Cout, "weight layout rows");
  const unsigned P = (unsigned)g.N * g.To * g.Ho * g.Wo;
  dim3 grid((P + CL_BN - 1) / CL_BN, g.Mrows / (16 * mt));
#define SLV_CL16(MT_)                                                                                              \
  hipLaunchKernelGGL((conv_cl16_kernel<MT_>), grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x_bf16, \
                     (const unsigned short*)w_layout_bf16, (unsigned short*)y_bf16, scale_shift,                    \
                     (const unsigned short*)res_bf16, relu, g)
  if (mt == 4) SLV_CL16(4);
  else if (mt == 8) SLV_CL16(8);
  else SLV_CL16(9);
#undef SLV_CL16
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_to_cl16(const float* x, void* y_bf16, int64_t N, int C, int Cp, int64_t S, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(x && y_bf16 && N > 0 && C > 0 && Cp >= C && (Cp & 7) == 0 && S > 0, "bad argument");
  const long long total = N * S * (Cp / 8);
  SLV_CHECK_ARG(total < 0xFFFFFFFFLL, "tensor too large");
  hipLaunchKernelGGL(to_cl16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (unsigned short*)y_bf16, C, Cp, (unsigned)S, (unsigned)total);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_maxpool_cl16(const void* x_bf16, void* y_bf16, int64_t N, int H, int W, int Cp, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(x_bf16 && y_bf16 && N > 0 && H > 0 && W > 0 && Cp > 0 && (Cp & 7) == 0, "bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long total = N * Ho * Wo * (Cp / 8);
  SLV_CHECK_ARG(total < 0xFFFFFFFFLL, "tensor too large");
  hipLaunchKernelGGL(maxpool_cl16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)x_bf16, (unsigned short*)y_bf16, H, W, Ho, Wo, Cp, (unsigned)total);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_avgpool_cl16(const void* x_bf16, float* y, int64_t N, int64_t S, int C, int Cp, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(x_bf16 && y && N > 0 && N <= 65535 && S > 0 && C > 0 && Cp >= C, "bad argument");
  hipLaunchKernelGGL(avgpool_cl16_kernel, dim3((C + 63) / 64, (unsigned)N), dim3(64), 0, (hipStream_t)stream,
                     (const unsigned short*)x_bf16, y, (int)S, C, Cp);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_to_cl16_wpatch(const float* x, void* y_bf16, int64_t N, int C, int64_t TH, int W, int kw, int sw, int pw,
                       slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(x && y_bf16 && N > 0 && C > 0 && TH > 0 && W > 0 && kw > 0 && sw > 0 && kw * C <= 32,
                "bad argument (kw * C must fit 32)");
  const int Wo = (W + 2 * pw - kw) / sw + 1;
  const long long total = N * TH * Wo * 4;
  SLV_CHECK_ARG(Wo > 0 && total < 0xFFFFFFFFLL && TH < 0x7FFFFFFFLL, "tensor too large");
  hipLaunchKernelGGL(to_cl16_wpatch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (unsigned short*)y_bf16, C, (unsigned)TH, W, Wo, kw, sw, pw, (unsigned)total);
  SLV_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

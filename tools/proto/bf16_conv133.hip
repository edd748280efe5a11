// PROTOTYPE (not part of libselavi_hip.so): what a 16-bit MFMA path buys on the hot layer.
// Conv3d(64 -> 144, (1,3,3), pad (0,1,1)) of R(2+1)D-18 layer 1 with bf16 CHANNELS-LAST activations
// [N][T][H][W][C], bf16 weights pre-laid-out [tap][k-half][Cout][32], fp32 accumulation on
// v_mfma_f32_16x16x32_bf16 (gfx950), bf16 channels-last output.  Implicit GEMM: D[cout][pos] = sum_k W[cout][k] X[k][pos],
// k = tap*64 + c.  Block = 4 waves = 144 cout x 128 positions; wave = 144 x 32 (9 x 2 MFMA tiles); K-step 32 =
// half a tap; register-staged double buffer, one barrier per K-step; LDS rows of 32 bf16 padded to 80 bytes
// (conflict-free ds_read_b128 fragments).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/proto/bf16_conv133.hip -o /tmp/bf16_conv133 && /tmp/bf16_conv133
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int CIN = 64, COUT = 144, TAPS = 9, KSTEPS = TAPS * 2, BN_POS = 128, ROWB = 80;   // bytes per LDS row

__device__ __forceinline__ unsigned short f2bf(float f) {           // round to nearest even
  unsigned int u = __float_as_uint(f);
  u += 0x7FFF + ((u >> 16) & 1);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }

__device__ __forceinline__ u32x4 bload16(__amdgpu_buffer_rsrc_t r, unsigned off) {
  // raw buffer load with range check: offsets beyond the resource's size return 0 (the conv's zero padding)
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

__global__ __launch_bounds__(256, 2) void conv133_bf16(const unsigned short* __restrict__ x,      // [P][64]
                                                       const unsigned short* __restrict__ wl,     // [18][144][32]
                                                       unsigned short* __restrict__ y,            // [P][144]
                                                       int T, int H, int W, unsigned P) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][(COUT + BN_POS) * ROWB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(P * CIN * 2u), 0x00020000);
  // ---- loader assignment.  B (activations): 128 rows x 4 pieces of 16 B = 512 pieces, 2 per thread.
  int brow[2], bh[2], bw[2];
  unsigned bbase[2];
  const int piece = tid & 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    brow[i] = (tid >> 2) + 64 * i;
    const unsigned p = blockIdx.x * BN_POS + brow[i];
    const unsigned hw = p % (unsigned)(H * W);
    bh[i] = hw / W;
    bw[i] = hw % W;
    bbase[i] = p < P ? p * (CIN * 2u) + piece * 16u : 0xFFFFFFF0u;
  }
  // A (weights): 144 rows x 4 pieces = 576 pieces: 2 per thread + 64 extra (threads 0..63 take a third)
  u32x4 ra[3], rb[2];
  auto gload = [&](int s) {
    const int tap = s >> 1, half = s & 1, dh = tap / 3 - 1, dw = tap % 3 - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = (unsigned)(bh[i] + dh) < (unsigned)H && (unsigned)(bw[i] + dw) < (unsigned)W;
      const unsigned off = bbase[i] + (unsigned)((dh * W + dw) * CIN * 2 + half * 64);
      rb[i] = bload16(rx, ok ? off : 0xFFFFFFF0u);
    }
    const unsigned short* ws = wl + (size_t)s * COUT * 32;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int pc = tid + 256 * i;
      if (i < 2 || tid < 64) ra[i] = *(const u32x4*)(ws + pc * 8);
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* A = lds[buf];
    unsigned char* B = A + COUT * ROWB;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int pc = tid + 256 * i;
      if (i < 2 || tid < 64) *(u32x4*)(A + (pc >> 2) * ROWB + (pc & 3) * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) *(u32x4*)(B + brow[i] * ROWB + piece * 16) = rb[i];
  };
  f32x4 acc[9][2];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fk = lane >> 4;          // fragment row/col and k-group (8 bf16 = 16 B)
  gload(0);
  lstore(0);
  __syncthreads();
  for (int s = 0; s < KSTEPS; ++s) {
    if (s + 1 < KSTEPS) gload(s + 1);
    const unsigned char* A = lds[s & 1];
    const unsigned char* B = A + COUT * ROWB;
    bf16x8 b[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = *(const bf16x8*)(B + (wave * 32 + j * 16 + fr) * ROWB + fk * 16);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const bf16x8 a = *(const bf16x8*)(A + (i * 16 + fr) * ROWB + fk * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[j], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < KSTEPS) lstore((s + 1) & 1);
    __syncthreads();
  }
  // ---- epilogue: C/D col = lane & 15 (position), rows (lane >> 4) * 4 + r (cout): 4 consecutive couts -> one 8-byte store
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const unsigned p = blockIdx.x * BN_POS + wave * 32 + j * 16 + fr;
    if (p >= P) continue;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const unsigned lo = f2bf(acc[i][j][0]) | ((unsigned)f2bf(acc[i][j][1]) << 16);
      const unsigned hi = f2bf(acc[i][j][2]) | ((unsigned)f2bf(acc[i][j][3]) << 16);
      *(uint2*)(y + (size_t)p * COUT + i * 16 + fk * 4) = make_uint2(lo, hi);
    }
  }
}

// naive reference on the same bf16 data (fp32 accumulate), one thread per (pos, cout)
__global__ void conv133_ref(const unsigned short* x, const unsigned short* w /* [144][9][64] */, float* y, int T, int H,
                            int W, unsigned P) {
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P * COUT) return;
  const unsigned p = idx / COUT, co = idx % COUT;
  const int hw = p % (H * W), h = hw / W, ww = hw % W;
  float s = 0.f;
  for (int tap = 0; tap < 9; ++tap) {
    const int dh = tap / 3 - 1, dw = tap % 3 - 1;
    if ((unsigned)(h + dh) >= (unsigned)H || (unsigned)(ww + dw) >= (unsigned)W) continue;
    const unsigned short* xs = x + ((size_t)p + dh * W + dw) * CIN;
    const unsigned short* ws = w + ((size_t)co * 9 + tap) * CIN;
    for (int c = 0; c < CIN; ++c) s += bf2f(xs[c]) * bf2f(ws[c]);
  }
  y[idx] = s;
}

static unsigned short h_f2bf(float f) {
  unsigned int u;
  std::memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (unsigned short)(u >> 16);
}
static float h_bf2f(unsigned short h) {
  unsigned int u = ((unsigned int)h) << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

static void run(int N, int T, int H, int W, bool check) {
  const unsigned P = (unsigned)N * T * H * W;
  std::vector<unsigned short> hx((size_t)P * CIN), hw((size_t)COUT * 9 * CIN), hwl((size_t)KSTEPS * COUT * 32);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.f - 0.5f; };
  for (auto& v : hx) v = h_f2bf(rnd() * 2.f);
  for (auto& v : hw) v = h_f2bf(rnd() * 0.1f);
  for (int st = 0; st < KSTEPS; ++st)
    for (int co = 0; co < COUT; ++co)
      for (int k = 0; k < 32; ++k) hwl[((size_t)st * COUT + co) * 32 + k] = hw[((size_t)co * 9 + st / 2) * CIN + (st & 1) * 32 + k];
  unsigned short *dx, *dw, *dwl, *dy;
  float* dr;
  hipMalloc(&dx, hx.size() * 2);
  hipMalloc(&dw, hw.size() * 2);
  hipMalloc(&dwl, hwl.size() * 2);
  hipMalloc(&dy, (size_t)P * COUT * 2);
  hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dwl, hwl.data(), hwl.size() * 2, hipMemcpyHostToDevice);
  const int grid = (P + BN_POS - 1) / BN_POS;
  hipLaunchKernelGGL(conv133_bf16, dim3(grid), dim3(256), 0, 0, dx, dwl, dy, T, H, W, P);
  hipDeviceSynchronize();
  if (check) {
    hipMalloc(&dr, (size_t)P * COUT * 4);
    hipLaunchKernelGGL(conv133_ref, dim3((P * COUT + 255) / 256), dim3(256), 0, 0, dx, dw, dr, T, H, W, P);
    std::vector<unsigned short> hy((size_t)P * COUT);
    std::vector<float> hr((size_t)P * COUT);
    hipMemcpy(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), dr, hr.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (size_t i = 0; i < hy.size(); ++i) {
      maxerr = fmax(maxerr, fabs(h_bf2f(hy[i]) - hr[i]));
      maxref = fmax(maxref, fabs(hr[i]));
    }
    printf("check N=%d T=%d %dx%d: max |err| %.4g of max |ref| %.4g (bf16 output rounding ~ %.4g)  %s\n", N, T, H, W, maxerr,
           maxref, maxref / 256, maxerr <= maxref / 128 ? "OK" : "MISMATCH");
    hipFree(dr);
  } else {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(conv133_bf16, dim3(grid), dim3(256), 0, 0, dx, dwl, dy, T, H, W, P);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flop = 2.0 * P * COUT * CIN * 9, bytes = (double)P * (CIN + COUT) * 2;
    printf("N=%d T=%d %dx%d: %.3f ms  %.1f TFLOP/s (%.3f of 2500 dense bf16)  %.2f TB/s algorithmic (%.3f of 8 TB/s)   [fp32 path: 1.24 ms]\n",
           N, T, H, W, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500, bytes / ms / 1e9, bytes / ms / 1e9 / 8);
  }
  hipFree(dx); hipFree(dw); hipFree(dwl); hipFree(dy);
}

int main() {
  run(1, 2, 12, 20, true);
  run(2, 3, 56, 56, true);
  run(16, 16, 56, 56, false);
  run(64, 16, 56, 56, false);
  return 0;
}

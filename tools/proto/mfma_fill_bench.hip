// How many independent VALU instructions does ONE wave hide behind each v_mfma_f32_16x16x32_bf16 (one wave per SIMD)?
// Build: hipcc --offload-arch=gfx950 -O3 tools/proto/mfma_fill_bench.hip -o /tmp/mfma_fill && /tmp/mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

template <int FILL, int AREG, int KIND, int LDSR = 0>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, float* sink, int iters) {
  f32x4 acc[12];
  for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b[4];
  for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)(threadIdx.x & 3);
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) b[j][i] = (__bf16)(float)((threadIdx.x + j) & 3);
  float f[6] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f};
  unsigned u[4] = {threadIdx.x, threadIdx.x * 3u, 7u, 9u};
  float big[24];
  for (int i = 0; i < 24; ++i) big[i] = (float)i;
  extern __shared__ unsigned char lds[];
  bf16x8 lb[4];
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      if (LDSR && (m & 1) == 0 && m < 8) lb[m >> 1] = *(const bf16x8*)(lds + threadIdx.x * 16 + (m >> 1) * 4096 + (it & 3) * 16384);
      if (LDSR == 2 && m < 4) { for (int i = 0; i < 8; ++i) b[m][i] = lb[m][i]; }
      if (AREG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "a"(a), "v"(b[m & 3]));
      else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b[m & 3]));
#pragma unroll
      for (int q = 0; q < FILL; ++q) {
        if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(m + q) % 6]) : "v"(f[(m + q + 3) % 6]));
        else if (KIND == 1) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[(m + q) & 3]));
        else if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[(m + q) & 3]) : "v"(f[q % 6]), "v"(f[(q + 1) % 6]));
        else if (KIND == 3) {       // the statistics chain of conv_cl16_sr: unpack -> add -> fma, one instruction per slot, dependent
          const int w = (m + q) % 3;
          if (w == 0) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(f[5]) : "v"(u[0]));
          else if (w == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[m % 4]) : "v"(f[5]));
          else asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(f[(m + 1) % 4]) : "v"(f[5]));
        } else if (KIND == 4) {     // the same with 24 distinct accumulators (register pressure / banks as in the kernel)
          const int w = (m + q) % 3;
          if (w == 0) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(f[5]) : "v"(u[0]));
          else if (w == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(big[(m * 2) % 24]) : "v"(f[5]));
          else asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(big[(m * 2 + 1) % 24]) : "v"(f[5]));
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_nop 15\n\ts_nop 7");
  float s = 0.f;
  for (int i = 0; i < 12; ++i) s += acc[i][0];
  for (int i = 0; i < 6; ++i) s += f[i];
  for (int i = 0; i < 24; ++i) s += big[i];
  if (LDSR) for (int j = 0; j < 4; ++j) s += (float)lb[j][0];
  for (int i = 0; i < 4; ++i) s += (float)u[i];
  if (s == 123.456f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int FILL, int AREG, int KIND, int LDSR = 0>
void run(const char* name) {
  unsigned long long* d; float* s;
  hipMalloc(&d, 1024 * 8); hipMalloc(&s, 4);
  const int iters = 2000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<FILL, AREG, KIND, LDSR>), dim3(256), dim3(256), 65536, 0, d, s, iters);
  hipDeviceSynchronize();
  unsigned long long h[1024]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < 1024; ++i) m += h[i];
  printf("%-44s fill %d: %.2f cycles per MFMA\n", name, FILL, m / 1024 / iters / 12);
  hipFree(d); hipFree(s);
}
int main() {
  run<0, 1, 0>("A in AGPR, v_add_f32"); run<1, 1, 0>("A in AGPR, v_add_f32"); run<2, 1, 0>("A in AGPR, v_add_f32"); run<3, 1, 0>("A in AGPR, v_add_f32"); run<4, 1, 0>("A in AGPR, v_add_f32");
  run<0, 0, 0>("A in VGPR, v_add_f32"); run<1, 0, 0>("A in VGPR, v_add_f32"); run<2, 0, 0>("A in VGPR, v_add_f32"); run<3, 0, 0>("A in VGPR, v_add_f32");
  run<1, 1, 1>("A in AGPR, v_lshlrev_b32"); run<2, 1, 1>("A in AGPR, v_lshlrev_b32"); run<3, 1, 1>("A in AGPR, v_lshlrev_b32");
  run<1, 1, 2>("A in AGPR, v_cvt_pk_bf16"); run<2, 1, 2>("A in AGPR, v_cvt_pk_bf16");
  run<1, 1, 3>("stats chain (6 regs)"); run<2, 1, 3>("stats chain (6 regs)");
  run<1, 1, 4>("stats chain (24 accumulators)"); run<2, 1, 4>("stats chain (24 accumulators)");
  run<0, 1, 0, 1>("4 ds_read_b128 per 12 MFMAs, unused"); run<1, 1, 4, 1>("4 ds_read_b128 per 12 + stats chain");
  run<0, 1, 0, 2>("4 ds_read_b128 per 12 feeding the MFMAs"); run<1, 1, 4, 2>("4 ds_read feeding MFMAs + stats chain");
  run<1, 1, 2, 2>("4 ds_read feeding MFMAs + cvt_pk");
  return 0;
}

// Micro-benchmark: LDS read throughput per CU of ds_read_b64_tr_b16 against ds_read_b64 / ds_read_b128 with the address
// patterns of csrc/wgrad_cl16_acc.hip (pixel pitch 160 / 352 bytes), 1..4 waves of one workgroup reading back to back.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tr_read_bench.hip -o /tmp/tr_read_bench && /tmp/tr_read_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>   // 0 tr_b64, 1 b64, 2 b128
__global__ __launch_bounds__(256, 1) void bench(unsigned* out, long long* cyc, int pitch, int iters, int nwaves) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 32768 / 4; i += 256) ((unsigned*)lds)[i] = i * 2654435761u;
  __syncthreads();
  const int fr = lane & 15, fk = lane >> 4, p0 = 4 * fk + (fr >> 2);
  unsigned addr;
  if (KIND == 2) addr = (unsigned)(((fr >> 3) * 10 + (fr & 7)) * pitch + fk * 16);     // b128 fragment of 2 rows x 8 pixels
  else addr = (unsigned)(((p0 >> 3) * 10 + (p0 & 7)) * pitch + wave * 32 + 8 * (fr & 3));
  addr += (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)lds;
  unsigned acc = 0;
  long long t0 = 0, t1 = 0;
  if (wave < nwaves) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (KIND == 0) {
          u32x2 v;
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(u * 160));
          asm volatile("" ::"v"(v));
        } else if (KIND == 1) {
          u32x2 v;
          asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(u * 160));
          asm volatile("" ::"v"(v));
        } else {
          u32x4 v;
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(u * 160));
          asm volatile("" ::"v"(v));
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime();
  }
  out[blockIdx.x * 256 + tid] = acc;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

int main() {
  unsigned* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
  const int iters = 2000;
  std::vector<long long> h(1024);
  const char* names[3] = {"ds_read_b64_tr_b16", "ds_read_b64", "ds_read_b128"};
  for (int kind = 0; kind < 3; ++kind)
    for (int pitch : {160, 352})
      for (int nw : {1, 2, 4}) {
        for (int rep = 0; rep < 2; ++rep) {
          if (kind == 0) hipLaunchKernelGGL(bench<0>, dim3(256), dim3(256), 65536, 0, out, cyc, pitch, iters, nw);
          if (kind == 1) hipLaunchKernelGGL(bench<1>, dim3(256), dim3(256), 65536, 0, out, cyc, pitch, iters, nw);
          if (kind == 2) hipLaunchKernelGGL(bench<2>, dim3(256), dim3(256), 65536, 0, out, cyc, pitch, iters, nw);
          hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int b = 0; b < 256; ++b) s += (double)h[b * 4];
        const double per = s / 256 / (iters * 16.0);                     // ticks per read instruction of one wave
        printf("%-20s pitch %3d  %d waves: %.2f ticks per wave-instruction, %.2f per CU-instruction (%d B each)\n", names[kind], pitch, nw,
               per, per / nw, kind == 2 ? 1024 : 512);
      }
  return 0;
}

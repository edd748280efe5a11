// Micro-benchmark: LDS fragment reads + fp32 MFMA only (no global traffic in the loop), to find the ceiling of
// the igemm compute loop per tiling scheme on gfx950.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_lds_bench.hip -o /tmp/mfma_lds_bench && /tmp/mfma_lds_bench
// Schemes (256 threads = 4 waves per workgroup, 16-deep K chunk, one barrier per chunk, like igemm.hpp):
//   A  16x16x4 MFMA, wave tile (MT*16) x (NT*16), fragments by ds_read_b32 (the production loop)
//   B  32x32x2 MFMA, wave tile (MT*32) x (NT*32)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MT, int NT, int OCC>
__global__ __launch_bounds__(256, OCC) void bench16(const float* __restrict__ src, float* __restrict__ dst, int chunks) {
  constexpr int BM = MT * 16, BN = NT * 64, AS = 18, BS = BN + 16;
  __shared__ float smem[2 * (BM * AS + 16 * BS)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * (BM * AS + 16 * BS); i += 256) smem[i] = src[i & 1023];
  __syncthreads();
  const int fi = lane & 15, fk = lane >> 4;
  f32x4 acc[MT][NT];
  for (int i = 0; i < MT; ++i)
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < chunks; ++c) {
    const float* As = smem + (c & 1) * (BM * AS + 16 * BS);
    const float* Bs = As + BM * AS;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = As[(i * 16 + fi) * AS + kk * 4 + fk];
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = Bs[(kk * 4 + fk) * BS + (wave * NT + j) * 16 + fi];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < MT; ++i)
    for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  dst[(size_t)blockIdx.x * 256 + tid] = s;
}

// 16x16x4 with 8-byte A fragment reads: k assignment k = 8h + 2fk + s, row stride 20 (64-bank conflict-free)
template <int MT, int NT, int OCC>
__global__ __launch_bounds__(256, OCC) void bench16b64(const float* __restrict__ src, float* __restrict__ dst, int chunks) {
  constexpr int BM = MT * 16, BN = NT * 64, AS = 20, BS = BN + 8;
  __shared__ float smem[2 * (BM * AS + 16 * BS)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * (BM * AS + 16 * BS); i += 256) smem[i] = src[i & 1023];
  __syncthreads();
  const int fi = lane & 15, fk = lane >> 4;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x4 acc[MT][NT];
  for (int i = 0; i < MT; ++i)
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < chunks; ++c) {
    const float* As = smem + (c & 1) * (BM * AS + 16 * BS);
    const float* Bs = As + BM * AS;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x2 a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = *(const f32x2*)&As[(i * 16 + fi) * AS + 8 * h + 2 * fk];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        b[j][0] = Bs[(8 * h + 2 * fk) * BS + (wave * NT + j) * 16 + fi];
        b[j][1] = Bs[(8 * h + 2 * fk + 1) * BS + (wave * NT + j) * 16 + fi];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s2], b[j][s2], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < MT; ++i)
    for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  dst[(size_t)blockIdx.x * 256 + tid] = s;
}

// 32x32x2: lane (fi = lane & 31, fk = lane >> 5) holds A[row fi][k fk] / B[k fk][col fi]; 16 accumulator regs
template <int MT, int NT, int OCC>
__global__ __launch_bounds__(256, OCC) void bench32(const float* __restrict__ src, float* __restrict__ dst, int chunks) {
  constexpr int BM = MT * 32, BN = NT * 128, AS = 18, BS = BN + 16;
  __shared__ float smem[2 * (BM * AS + 16 * BS)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * (BM * AS + 16 * BS); i += 256) smem[i] = src[i & 1023];
  __syncthreads();
  const int fi = lane & 31, fk = lane >> 5;
  f32x16 acc[MT][NT];
  for (int i = 0; i < MT; ++i)
    for (int j = 0; j < NT; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int c = 0; c < chunks; ++c) {
    const float* As = smem + (c & 1) * (BM * AS + 16 * BS);
    const float* Bs = As + BM * AS;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = As[(i * 32 + fi) * AS + kk * 2 + fk];
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = Bs[(kk * 2 + fk) * BS + (wave * NT + j) * 32 + fi];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < MT; ++i)
    for (int j = 0; j < NT; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  dst[(size_t)blockIdx.x * 256 + tid] = s;
}

template <typename F>
static void run(const char* name, F launch, double flop_per_block_chunk, int blocks, int chunks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  printf("%-44s %8.3f ms  %7.1f TF\n", name, ms, flop_per_block_chunk * blocks * chunks / ms / 1e9);
}

int main() {
  float *src, *dst;
  hipMalloc(&src, 4096);
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.f - 0.5f;
  hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
  const int blocks = 6144, chunks = 36;
  hipMalloc(&dst, (size_t)blocks * 256 * 4);
#define RUN16(MT, NT, OCC) run("16x16x4  tile " #MT "x16 x " #NT "x64  occ " #OCC, [&] { hipLaunchKernelGGL((bench16<MT, NT, OCC>), dim3(blocks), dim3(256), 0, 0, src, dst, chunks); }, 2.0 * MT * 16 * NT * 64 * 16, blocks, chunks)
#define RUN32(MT, NT, OCC) run("32x32x2  tile " #MT "x32 x " #NT "x128 occ " #OCC, [&] { hipLaunchKernelGGL((bench32<MT, NT, OCC>), dim3(blocks), dim3(256), 0, 0, src, dst, chunks); }, 2.0 * MT * 32 * NT * 128 * 16, blocks, chunks)
  RUN16(9, 2, 3);
  RUN16(9, 1, 4);
  RUN16(8, 2, 4);
  RUN16(4, 2, 5);
  RUN16(4, 1, 8);
  RUN16(15, 1, 3);
#define RUN16B(MT, NT, OCC) run("16x16x4 b64-A tile " #MT "x16 x " #NT "x64 occ " #OCC, [&] { hipLaunchKernelGGL((bench16b64<MT, NT, OCC>), dim3(blocks), dim3(256), 0, 0, src, dst, chunks); }, 2.0 * MT * 16 * NT * 64 * 16, blocks, chunks)
  RUN16B(9, 2, 3);
  RUN16B(8, 2, 4);
  RUN16B(4, 1, 8);
  RUN32(4, 1, 3);
  RUN32(2, 1, 5);
  RUN32(5, 1, 2);
  RUN32(1, 1, 8);
  return 0;
}

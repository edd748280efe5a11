// Probe of ds_read_b64_tr_b16 (__builtin_amdgcn_ds_read_tr16_b64_v4i16) on gfx950: which LDS element lands in which
// (lane, element) slot.  LDS holds its own halfword index; lane l passes the byte address addr[l].
//   hipcc --offload-arch=gfx950 -O2 tools/proto/tr16_probe.hip -o tools/proto/tr16_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const int* addr, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  int h[64]; short o[256];
  int *d; short* od;
  hipMalloc(&d, sizeof(h)); hipMalloc(&od, sizeof(o));
  // pattern A: lane i of a 16-lane group g points at row (i>>2) + 4g (row stride 64 halfwords), column chunk 4*(i&3)
  for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; h[l] = ((i >> 2) + 4 * g) * 64 + 4 * (i & 3); }
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d, od);
  hipMemcpy(o, od, sizeof(o), hipMemcpyDeviceToHost);
  printf("pattern A (row = (i>>2)+4g, stride 64, col chunk = i&3): lane: 4 values as (row,col)\n");
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf(" (%d,%d)", o[l * 4 + j] / 64, o[l * 4 + j] % 64);
    printf("\n");
  }
  return 0;
}

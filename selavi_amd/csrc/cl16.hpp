// Shared pieces of the 16-bit MFMA path (bf16 CHANNELS-LAST activations [N][T][H][W][Cp], Cp % 32 == 0, padding
// channels zero; fp32 accumulation, fp32 master weights and BatchNorm parameters).  BASELINE configs[4]: the reference
// trains its convs in half precision through apex O1 (main.py:151-153,296-299); here the half type is bf16 (same
// exponent range as fp32: no loss scaling).
#pragma once
#include "common.hpp"

namespace slv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned short f2bf(float f) {             // round to nearest even (finite inputs)
  unsigned int u = __float_as_uint(f);
  u += 0x7FFF + ((u >> 16) & 1);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }
__device__ __forceinline__ float bf_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
// two floats -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  const f32x2 r = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}
// ReLU on a packed bf16 pair: a negative bf16 is a negative int16 (v_pk_max_i16); -0 -> +0
__device__ __forceinline__ unsigned relu_bf2(unsigned v) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), (s16x2){0, 0}));
}
// The BatchNorm affine as EVERY kernel of this path evaluates it (one fused multiply-add in fp32): the ReLU mask of the
// backward kernels must reproduce the forward's sign decision bit for bit.
__device__ __forceinline__ float bn_affine(float x, float s, float h) { return __builtin_fmaf(x, s, h); }

// relu(x*s + h) on the 8 bf16 of a 16-byte piece; s, h: the piece's 8 channels
__device__ __forceinline__ u32x4 affine_relu8(u32x4 v, const float* s, const float* h) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    o[i] = relu_bf2(pack_bf2(bn_affine(bf_lo(v[i]), s[2 * i], h[2 * i]), bn_affine(bf_hi(v[i]), s[2 * i + 1], h[2 * i + 1])));
  return o;
}

}  // namespace slv

// Scaled two-way fp16 operand split shared by the fp32-accurate contractions on the f16 matrix cores (conv_split.hip:
// tokenizer convolutions / SigLIP projections; siglip_attn.hip: SigLIP attention).  See the header of conv_split.hip.
#pragma once
#include "common.h"

typedef _Float16 h16x8_t __attribute__((ext_vector_type(8)));     // f16 MFMA operand
typedef _Float16 h16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr float F16_MAX = 65504.f;

// power-of-two scale exponent from an upper bound of max|x| (device float, or null = unscaled): the bound's binade
// [2^k, 2^(k+1)) maps to [2^14, 2^15).  Zero / denormal / non-finite bounds clamp; so does anything outside 2^+-60.
__device__ __forceinline__ int scale_exp(const float* amax) {
  if (!amax) return 0;
  const int k = (int)((__float_as_uint(*amax) >> 23) & 0xffu) - 127;
  return max(-60, min(60, 14 - k));
}
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

// two fp32 values (already scaled) -> two packed fp16 pairs; saturating, so a caller's too-small bound cannot make infinities
__device__ __forceinline__ void split2_pair(float a, float b, uint32_t& p1, uint32_t& p2) {
  a = __builtin_fminf(__builtin_fmaxf(a, -F16_MAX), F16_MAX);
  b = __builtin_fminf(__builtin_fmaxf(b, -F16_MAX), F16_MAX);
  const h16x2_t h1 = __builtin_convertvector(f32x2_t{a, b}, h16x2_t);              // round to nearest even
  const f32x2_t back = __builtin_convertvector(h1, f32x2_t);
  const h16x2_t h2 = __builtin_convertvector(f32x2_t{a - back[0], b - back[1]}, h16x2_t);
  p1 = __builtin_bit_cast(uint32_t, h1);
  p2 = __builtin_bit_cast(uint32_t, h2);
}

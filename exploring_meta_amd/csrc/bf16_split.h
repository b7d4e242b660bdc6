// Split-bf16 operand helpers shared by the convolution kernels (conv_mfma.hip, wgrad_bf16.hip).
// An fp32 value is the exact sum of three bf16 pieces, x = h + m + l (round-to-nearest pieces of x, x - h, x - h - m; 8 mantissa bits
// each), so an fp32 product is  ah bh + ah bm + am bh + am bm + ah bl + al bh  up to the three dropped terms (am bl, al bm, al bl:
// <= 2^-24 |a b| together -- the size of ONE fp32 rounding), each partial product exact in the fp32 accumulator of
// v_mfma_f32_32x32x16_bf16.  Six 8-pass bf16 MFMAs cover K = 16 where the fp32 pipe needs eight 16-pass ones: 2.7x the matrix rate,
// provided the 4.5 VALU instructions per split value issue in the MFMAs' shadow (tools/mfma_valu_overlap_probe.hip: they do).
#pragma once
#include "mi_common.h"
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
// (the subtractions as two scalar v_sub_f32 each, pinned in assembly: the vectoriser would make them one v_pk_add_f32, which takes 2.4x the
// issue time of a plain VALU instruction and -- alone among the instructions used here -- does not run in the shadow of an MFMA,
// tools/mfma_shadow_probe.hip)
__device__ __forceinline__ float bf16_sub(float a, float b) {
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void bf16_split2(floatx2 v, unsigned& h, unsigned& m, unsigned& l) {
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));                    // v_cvt_pk_bf16_f32 (RNE)
  const floatx2 r = {bf16_sub(v[0], __uint_as_float(h << 16)), bf16_sub(v[1], __uint_as_float(h & 0xffff0000u))};   // exact
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  const floatx2 q = {bf16_sub(r[0], __uint_as_float(m << 16)), bf16_sub(r[1], __uint_as_float(m & 0xffff0000u))};   // exact
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2));
}
#define MI_BF8(q) __builtin_bit_cast(bf16x8, (mi_u32x4{(q)[0], (q)[1], (q)[2], (q)[3]}))
#define MI_BF_MFMA(x, y, acc) __builtin_amdgcn_mfma_f32_32x32x16_bf16(MI_BF8(x), MI_BF8(y), acc, 0, 0, 0)

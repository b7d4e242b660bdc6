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

// ---- Two-plane fp16 operand form.  x * s = h + l with h = fp16(x * s) and l = fp16(x * s - h) (round to nearest both; the residual is
// exact in fp32): 22 significand bits, |x s - h - l| <= 2^-22 |x s| while l is a normal fp16 (|x s| >= 2^-3) and <= 2^-25 absolutely
// below (fp16 denormals: v_mfma_f32_32x32x16_f16 keeps them, tools/f16_split_probe.hip).  s is a power of two chosen per (task, tensor)
// from the tensor's largest magnitude so that it lands in [2^14, 2^15): fp16's narrow exponent range then covers 2^-17.5 of the maximum
// at full precision and 2^-39.5 of it absolutely -- finer than the fp32 rounding of any sum the maximum takes part in.  A product needs
// THREE MFMAs (l b_h, h b_l, h b_h; the dropped l b_l is 2^-22 of the product) where the bf16 form needs six, a value 2.5 vector
// instructions where it needs 4.5: v_fma_mixlo/mixhi_f16 scale and round in one step, v_fma_mix_f32 forms the residual from the packed
// half, v_cvt_pk_f16_f32 rounds a pair of residuals.  The accumulators carry the factor s_a s_b; the epilogues divide it out (exact).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// exponent k of the scale 2^k for a tensor whose largest magnitude has the fp32 bits `amax_bits` (0, denormal, inf and NaN included):
// amax 2^k in [2^14, 2^15), clamped to +-60 so that products and quotients of two scales stay normal fp32 numbers
__device__ __forceinline__ int f16_scale_exp(unsigned amax_bits) {
  const int k = 141 - (int)((amax_bits >> 23) & 0xffu);
  return k < -60 ? -60 : (k > 60 ? 60 : k);
}
__device__ __forceinline__ float f16_pow2(int k) { return __uint_as_float((unsigned)(127 + k) << 23); }   // -126 <= k <= 127
// Two terms that accumulate into the same registers need ONE factor: the term with the larger scale product gives way (its operands sit
// lower in fp16's range; what falls below the absolute floor there is below the other term's fp32 rounding).  k[term][operand].
__device__ __forceinline__ void f16_common_scale(int (&k)[2][2]) {
  const int e0 = k[0][0] + k[0][1], e1 = k[1][0] + k[1][1];
  const int t = e0 > e1 ? 0 : 1;
  int d = e0 > e1 ? e0 - e1 : e1 - e0;                       // <= 240
  const int d1 = d < k[t][1] + 126 ? d : k[t][1] + 126;      // the second operand first, down to 2^-126 ...
  k[t][1] -= d1; d -= d1;
  k[t][0] -= d < k[t][0] + 126 ? d : k[t][0] + 126;          // ... then the first (both exhausted: the term is 2^-250 of the other)
}
#define MI_F16_SPLIT_BODY(SC)                                                                                              \
  unsigned hh;                                                                                                             \
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hh) : "v"(v[0]), SC(s));                                                      \
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hh) : "v"(v[1]), SC(s));                                                      \
  float r0, r1;                                                                                                            \
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(v[0]), SC(s), "v"(hh));                           \
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(v[1]), SC(s), "v"(hh));            \
  h = hh;                                                                                                                  \
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(floatx2{r0, r1}, f16x2));      /* v_cvt_pk_f16_f32 (RNE) */
#define MI_F16_SGPR "s"
#define MI_F16_VGPR "v"
// s wave-uniform (a scalar register) / per lane
__device__ __forceinline__ void f16_split2(floatx2 v, float s, unsigned& h, unsigned& l) { MI_F16_SPLIT_BODY(MI_F16_SGPR) }
__device__ __forceinline__ void f16_split2_v(floatx2 v, float s, unsigned& h, unsigned& l) { MI_F16_SPLIT_BODY(MI_F16_VGPR) }
#define MI_F16X8(q) __builtin_bit_cast(f16x8, (mi_u32x4{(q)[0], (q)[1], (q)[2], (q)[3]}))
#define MI_F16_MFMA(x, y, acc) __builtin_amdgcn_mfma_f32_32x32x16_f16(MI_F16X8(x), MI_F16X8(y), acc, 0, 0, 0)

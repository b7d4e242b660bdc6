// Canonical fold of the per-workgroup partials of a weight-gradient element (conv_mfma.hip reduce_partials_kernel, gram.hip).
// The nch partials of an element are taken in rounds of 16 consecutive ones; the rounds are dealt round-robin to S = fold_slices(nch)
// slices; a slice adds its rounds in increasing order (the 16 values of a round in order), and the slices' sums are added in slice order.
// S = 1 up to 16 partials -- the plain sequential sum, i.e. what every launch at 32 tasks per call has always computed -- and grows to 8
// for the few-task calls whose tasks spread over 64 .. 256 workgroups: the slices of an element then run on different threads (one trip
// to memory instead of up to sixteen) and every kernel that folds uses this one order, so fused and separate launches stay bit-identical.
#pragma once
#include "mi_common.h"

__host__ __device__ inline int fold_slices(int nch) {
  const int rounds = (nch + 15) / 16;
  int s = 1;
  while (s < rounds && s < 8) s *= 2;
  return s;
}

// sum of slice s (of S) of the partials p[c * stride], c < nch, in ACC precision
template <typename ACC>
__device__ __forceinline__ ACC fold_slice(const float* __restrict__ p, size_t stride, int nch, int S, int s) {
  ACC acc = (ACC)0;
  for (int r0 = s * 16; r0 < nch; r0 += S * 16) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = p[(size_t)(r0 + k < nch ? r0 + k : r0) * stride];
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (r0 + k < nch) acc += (ACC)v[k];
  }
  return acc;
}
// all slices by one thread, in the canonical order (kernels without a spare thread per slice)
template <typename ACC>
__device__ __forceinline__ ACC fold_all(const float* __restrict__ p, size_t stride, int nch) {
  const int S = fold_slices(nch);
  ACC t = fold_slice<ACC>(p, stride, nch, S, 0);
  for (int s = 1; s < S; ++s) t += fold_slice<ACC>(p, stride, nch, S, s);
  return t;
}

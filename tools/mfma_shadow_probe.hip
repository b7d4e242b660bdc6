// Which VALU instructions issue in the shadow of a running bf16 MFMA (v_mfma_f32_32x32x16_bf16, 8 passes), same wave, one wave per SIMD?
// Per iteration: 12 dependent MFMAs, and after each of them NPER independent instances of one VALU instruction.  If the instruction
// overlaps, the time stays at the MFMA-only time; if not, it adds NPER * 12 * 4 cycles.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_shadow_probe.hip -o /tmp/sh && /tmp/sh
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
#define OPS4(ASM) asm volatile(ASM "\n" ASM "\n" ASM "\n" ASM : "+v"(r0), "+v"(r1) : "v"(c0), "v"(c1));
#define KERNEL(NAME, ASM)                                                                               \
  __global__ __launch_bounds__(64) void NAME(float* out, int iters, int with_mfma) {                    \
    floatx16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;                                            \
    uintx4 au = {threadIdx.x * 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};                     \
    bf16x8 a = __builtin_bit_cast(bf16x8, au);                                                          \
    unsigned r0 = threadIdx.x, r1 = threadIdx.x + 7, c0 = 0x3f800000u + threadIdx.x, c1 = 0x05040100u;  \
    if (with_mfma) {                                                                                    \
      for (int it = 0; it < iters; ++it) {                                                              \
        _Pragma("unroll") for (int m = 0; m < 12; ++m) {                                                \
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);                            \
          __builtin_amdgcn_sched_barrier(0);                                                            \
          OPS4(ASM)                                                                                     \
          __builtin_amdgcn_sched_barrier(0);                                                            \
        }                                                                                               \
      }                                                                                                 \
    } else {                                                                                            \
      for (int it = 0; it < iters; ++it) {                                                              \
        _Pragma("unroll") for (int m = 0; m < 12; ++m) { OPS4(ASM) }                                    \
      }                                                                                                 \
    }                                                                                                   \
    float s = 0.f; for (int r = 0; r < 16; ++r) s += acc[r];                                            \
    out[blockIdx.x * 64 + threadIdx.x] = s + __uint_as_float(r0 ^ r1);                                  \
  }
KERNEL(k_none, "s_nop 0")
KERNEL(k_fma, "v_fma_f32 %0, %0, %2, %2")
KERNEL(k_and, "v_and_b32 %0, 0xffff0000, %0")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 16, %0")
KERNEL(k_cvt, "v_cvt_pk_bf16_f32 %0, %0, %2")
KERNEL(k_sub, "v_sub_f32 %0, %0, %2")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %3")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 16")
KERNEL(k_dppand, "v_and_b32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_andor, "v_and_or_b32 %0, %0, %2, %1")
KERNEL(k_addu, "v_add_u32 %0, %0, %2")
__global__ __launch_bounds__(64) void k_pkadd(float* out, int iters, int with_mfma) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  floatx16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  uintx4 au = {threadIdx.x * 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  bf16x8 a = __builtin_bit_cast(bf16x8, au);
  f2 r0 = {1.f * threadIdx.x, 2.f}, c0 = {1.0001f, 0.5f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      if (with_mfma) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("v_pk_add_f32 %0, %0, %1\nv_pk_add_f32 %0, %0, %1\nv_pk_add_f32 %0, %0, %1\nv_pk_add_f32 %0, %0, %1" : "+v"(r0) : "v"(c0));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f; for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 64 + threadIdx.x] = s + r0[0] + r0[1];
}
template <class K> static float timeit(K k, float* out, int with) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, out, 4000, with);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, out, 4000, with);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* out; hipMalloc(&out, 1024 * 64 * 4);
#define RUN(K) printf("%-10s with MFMA %.3f ms | alone %.3f ms\n", #K, timeit(K, out, 1), timeit(K, out, 0));
  RUN(k_none) RUN(k_fma) RUN(k_and) RUN(k_lshl) RUN(k_cvt) RUN(k_sub) RUN(k_perm) RUN(k_alignbit) RUN(k_dppand) RUN(k_andor) RUN(k_addu) RUN(k_pkadd)
  return 0;
}

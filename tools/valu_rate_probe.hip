// Issue cost (cycles per wave64 instruction, one wave per SIMD) of the VALU instructions the split-bf16 convolution spends its time on.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate_probe.hip -o /tmp/vr && /tmp/vr
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define BODY(NAME, ASM)                                                                                     \
  __global__ __launch_bounds__(64) void NAME(unsigned* out, unsigned long long* cyc, int iters) {           \
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    unsigned b0 = 0x3f800000u + threadIdx.x, b1 = b0 + 1;                                                   \
    unsigned long long t0 = __builtin_readcyclecounter();                                                   \
    for (int i = 0; i < iters; ++i) {                                                                       \
      REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));) \
    }                                                                                                       \
    unsigned long long t1 = __builtin_readcyclecounter();                                                   \
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                             \
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;                                              \
  }
// 8 independent instructions per asm block, 8 blocks per iteration = 64 instructions
BODY(k_and, "v_and_b32 %0, 0xffff0000, %0\n v_and_b32 %1, 0xffff0000, %1\n v_and_b32 %2, 0xffff0000, %2\n v_and_b32 %3, 0xffff0000, %3\n v_and_b32 %4, 0xffff0000, %4\n v_and_b32 %5, 0xffff0000, %5\n v_and_b32 %6, 0xffff0000, %6\n v_and_b32 %7, 0xffff0000, %7")
BODY(k_sub, "v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8")
BODY(k_cvt, "v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %1, %1, %8\n v_cvt_pk_bf16_f32 %2, %2, %8\n v_cvt_pk_bf16_f32 %3, %3, %8\n v_cvt_pk_bf16_f32 %4, %4, %8\n v_cvt_pk_bf16_f32 %5, %5, %8\n v_cvt_pk_bf16_f32 %6, %6, %8\n v_cvt_pk_bf16_f32 %7, %7, %8")
BODY(k_perm, "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9")
BODY(k_dpp, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")
BODY(k_rowdpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
BODY(k_cnd, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc")
BODY(k_cnddpp, "v_cndmask_b32_dpp %0, %1, %8, vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %1, %2, %8, vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %2, %3, %8, vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %3, %4, %8, vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %4, %5, %8, vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %5, %6, %8, vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %6, %7, %8, vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %7, %0, %8, vcc wave_shr:1 row_mask:0xf bank_mask:0xf")
BODY(k_lshl, "v_lshlrev_b32 %0, 16, %0\n v_lshlrev_b32 %1, 16, %1\n v_lshlrev_b32 %2, 16, %2\n v_lshlrev_b32 %3, 16, %3\n v_lshlrev_b32 %4, 16, %4\n v_lshlrev_b32 %5, 16, %5\n v_lshlrev_b32 %6, 16, %6\n v_lshlrev_b32 %7, 16, %7")
// the fp16 split's instructions (bf16_split.h): scale-and-round, residual from the packed half, pair conversion
BODY(k_mixlo, "v_fma_mixlo_f16 %0, %0, %8, 0\n v_fma_mixlo_f16 %1, %1, %8, 0\n v_fma_mixlo_f16 %2, %2, %8, 0\n v_fma_mixlo_f16 %3, %3, %8, 0\n v_fma_mixlo_f16 %4, %4, %8, 0\n v_fma_mixlo_f16 %5, %5, %8, 0\n v_fma_mixlo_f16 %6, %6, %8, 0\n v_fma_mixlo_f16 %7, %7, %8, 0")
BODY(k_mixhi, "v_fma_mixhi_f16 %0, %0, %8, 0\n v_fma_mixhi_f16 %1, %1, %8, 0\n v_fma_mixhi_f16 %2, %2, %8, 0\n v_fma_mixhi_f16 %3, %3, %8, 0\n v_fma_mixhi_f16 %4, %4, %8, 0\n v_fma_mixhi_f16 %5, %5, %8, 0\n v_fma_mixhi_f16 %6, %6, %8, 0\n v_fma_mixhi_f16 %7, %7, %8, 0")
BODY(k_mix32, "v_fma_mix_f32 %0, %0, %8, -%9 op_sel_hi:[0,0,1]\n v_fma_mix_f32 %1, %1, %8, -%9 op_sel_hi:[0,0,1]\n v_fma_mix_f32 %2, %2, %8, -%9 op_sel_hi:[0,0,1]\n v_fma_mix_f32 %3, %3, %8, -%9 op_sel_hi:[0,0,1]\n v_fma_mix_f32 %4, %4, %8, -%9 op_sel_hi:[0,0,1]\n v_fma_mix_f32 %5, %5, %8, -%9 op_sel_hi:[0,0,1]\n v_fma_mix_f32 %6, %6, %8, -%9 op_sel_hi:[0,0,1]\n v_fma_mix_f32 %7, %7, %8, -%9 op_sel_hi:[0,0,1]")
BODY(k_cvtf16, "v_cvt_pk_f16_f32 %0, %0, %8\n v_cvt_pk_f16_f32 %1, %1, %8\n v_cvt_pk_f16_f32 %2, %2, %8\n v_cvt_pk_f16_f32 %3, %3, %8\n v_cvt_pk_f16_f32 %4, %4, %8\n v_cvt_pk_f16_f32 %5, %5, %8\n v_cvt_pk_f16_f32 %6, %6, %8\n v_cvt_pk_f16_f32 %7, %7, %8")
BODY(k_fma, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9")
BODY(k_cvt1f16, "v_cvt_f16_f32 %0, %0\n v_cvt_f16_f32 %1, %1\n v_cvt_f16_f32 %2, %2\n v_cvt_f16_f32 %3, %3\n v_cvt_f16_f32 %4, %4\n v_cvt_f16_f32 %5, %5\n v_cvt_f16_f32 %6, %6\n v_cvt_f16_f32 %7, %7")
BODY(k_cvt32f16, "v_cvt_f32_f16 %0, %0\n v_cvt_f32_f16 %1, %1\n v_cvt_f32_f16 %2, %2\n v_cvt_f32_f16 %3, %3\n v_cvt_f32_f16 %4, %4\n v_cvt_f32_f16 %5, %5\n v_cvt_f32_f16 %6, %6\n v_cvt_f32_f16 %7, %7")
BODY(k_mul, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8")
__global__ __launch_bounds__(64) void k_pkadd(unsigned* out, unsigned long long* cyc, int iters) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a0 = {1.f * threadIdx.x, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b = {1.0001f, 0.5f};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + threadIdx.x] = (unsigned)(a0[0] + a1[1] + a2[0] + a3[1]);
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  unsigned* out; unsigned long long* cyc; hipMalloc(&out, 1024 * 64 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
#define RUN(K) { hipLaunchKernelGGL(K, dim3(1024), dim3(64), 0, 0, out, cyc, iters); hipDeviceSynchronize(); hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0); hipLaunchKernelGGL(K, dim3(1024), dim3(64), 0, 0, out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-10s %.3f ms  -> %.2f ns per instruction per wave (counter %.2f per instr)\n", #K, ms, ms * 1e6 / (iters * 64.0), (double)c / (iters * 64.0)); }
  RUN(k_and) RUN(k_sub) RUN(k_lshl) RUN(k_cvt) RUN(k_perm) RUN(k_pkadd) RUN(k_cnd) RUN(k_dpp) RUN(k_rowdpp) RUN(k_cnddpp) RUN(k_mul) RUN(k_fma) RUN(k_mixlo) RUN(k_mixhi) RUN(k_mix32) RUN(k_cvtf16) RUN(k_cvt1f16) RUN(k_cvt32f16)
  return 0;
}

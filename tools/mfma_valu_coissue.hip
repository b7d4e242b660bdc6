// Micro-benchmark: do fp32 MFMAs and vector ALU instructions of DIFFERENT waves overlap on one SIMD of an MI355X?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_coissue.hip -o coissue && ./coissue
// Measured (round 1, profiles/README.md): a dependent chain of v_mfma_f32_32x32x2_f32 in ONE wave already saturates the pipe
// (0.601 ms for 20000 MFMAs vs 0.582 ms ideal), and a SIMD shared by one MFMA wave and one VALU wave takes the SUM of the two
// (1.10 ms = 0.60 + 0.50; 2.63 ms = 0.60 + 2.03): the fp32-input MFMA executes on the vector FMA lanes (its peak equals the fp32
// vector peak, 157.3 TFLOP/s), so every VALU instruction in an fp32-MFMA kernel costs four cycles of matrix time.  The conv
// kernels' "MFMA busy + 4 x VALU instructions" therefore adds up to ~90 % of their cycles, and the lever left is the VALU count.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
// NACC independent accumulator chains per wave; waves per SIMD set by the grid (blocks of 64 threads)
template <int NACC>
__global__ __launch_bounds__(64) void mfma_only(float* out, int iters) {
  floatx16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
// half of the waves run an MFMA loop, the other half a VALU loop (no per-iteration branch)
template <int VPM>
__global__ __launch_bounds__(64) void split(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, s = 0.f;
  if ((blockIdx.x & 1) == 0) {
    floatx16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) s += acc[r];
  } else {
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < VPM; u += 4) { v0 = __builtin_fmaf(v0, b, a); v1 = __builtin_fmaf(v1, b, a); v2 = __builtin_fmaf(v2, b, a); v3 = __builtin_fmaf(v3, b, a); }
    }
    s = v0 + v1 + v2 + v3;
  }
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* out; hipMalloc(&out, 1 << 24);
  const int it = 20000;
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 1024 * wps;
    printf("MFMA only, %d waves/SIMD: 1 chain %.3f ms", wps, timeit([&] { hipLaunchKernelGGL((mfma_only<1>), dim3(blocks), dim3(64), 0, 0, out, it); }));
    printf(" | 2 chains (2x MFMAs) %.3f ms", timeit([&] { hipLaunchKernelGGL((mfma_only<2>), dim3(blocks), dim3(64), 0, 0, out, it); }));
    printf(" | 4 chains (4x MFMAs) %.3f ms\n", timeit([&] { hipLaunchKernelGGL((mfma_only<4>), dim3(blocks), dim3(64), 0, 0, out, it); }));
  }
  printf("ideal: 20000 MFMAs x 64 cycles per wave-chain at ~2.2 GHz = %.3f ms per (wave x chain) on one SIMD\n", 20000 * 64 / 2.2e6);
  for (int wps : {2, 4, 8}) {
    const int blocks = 1024 * wps;
    printf("split, %d waves/SIMD (half MFMA, half VALU): 16 VALU/iter %.3f ms", wps, timeit([&] { hipLaunchKernelGGL((split<16>), dim3(blocks), dim3(64), 0, 0, out, it); }));
    printf(" | 64 VALU/iter %.3f ms\n", timeit([&] { hipLaunchKernelGGL((split<64>), dim3(blocks), dim3(64), 0, 0, out, it); }));
  }
  return 0;
}

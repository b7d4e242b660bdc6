// Rate of v_mfma_f32_32x32x16_bf16 as ONE dependent accumulator chain vs two / three interleaved chains (one wave per SIMD, and two).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_chain_probe.hip -o /tmp/ch && /tmp/ch
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
template <int NCH>
__global__ __launch_bounds__(64) void chains(float* out, int iters) {
  floatx16 acc[NCH];
  for (int c = 0; c < NCH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  uintx4 au = {threadIdx.x * 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  bf16x8 a = __builtin_bit_cast(bf16x8, au);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 12; ++m) acc[m % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc[m % NCH], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < NCH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NCH> static float run(float* out, int wps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(chains<NCH>, dim3(1024 * wps), dim3(64), 0, 0, out, 4000);
  hipEventRecord(e0);
  hipLaunchKernelGGL(chains<NCH>, dim3(1024 * wps), dim3(64), 0, 0, out, 4000);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 64 * 4);
  for (int w = 1; w <= 2; ++w)
    printf("waves/SIMD %d: 1 chain %.3f ms | 2 chains %.3f | 3 chains %.3f | 4 chains %.3f   (48000 MFMAs per wave)\n", w, run<1>(out, w), run<2>(out, w),
           run<3>(out, w), run<4>(out, w));
  return 0;
}

// Probe: bf16 MFMAs (v_mfma_f32_32x32x16_bf16, 8 passes) whose B operand is PRODUCED by vector instructions right in front of them (NV per MFMA:
// v_and of loop-carried registers with a mask into the operand tuple), two accumulator chains -- the shape of the sparse weight gradient's
// split-bf16 loop (csrc/gram.hip).  Does the vector work hide under the matrix pipe, at one and two waves per SIMD?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_dep_probe.hip -o /tmp/dep && /tmp/dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

template <int NV, int MODE, int FRESH>   // MODE 0: MFMA only (constant B), 1: VALU only, 2: both (B from the ANDs);  FRESH: 1 = a new tuple per MFMA (no WAR on the operand registers)
__global__ __launch_bounds__(64) void probe(float* out, int iters, unsigned seed) {
  floatx16 acc, acc2;
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
  uintx4 au = {threadIdx.x * 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  const bf16x8 a = __builtin_bit_cast(bf16x8, au);
  unsigned v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed * (threadIdx.x + i + 1);
  unsigned m = seed | 0x3f803f80u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      uintx4 b = au;
      if (MODE != 0) {
#pragma unroll
        for (int u = 0; u < NV; ++u) { v[u % 8] = (v[u % 8] & m) + (FRESH ? 0u : 0u); asm volatile("" : "+v"(v[u % 8])); }
        b = uintx4{v[0] & 0x3f803f80u, v[1] & 0x3f803f80u, v[2] & 0x3f803f80u, v[3] & 0x3f803f80u};
      }
      if (MODE != 1) {
        if (k & 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b), acc2, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
      } else {
        asm volatile("" : "+v"(b));
        v[0] ^= b[0];
      }
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
  for (int i = 0; i < 8; ++i) s += __uint_as_float(v[i]);
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NV, int MODE>
static float run(int waves_per_simd, float* out) {
  const int iters = 4000;
  dim3 grid(256 * 4 * waves_per_simd);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NV, MODE, 0>), grid, dim3(64), 0, 0, out, iters, 3u);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<NV, MODE, 0>), grid, dim3(64), 0, 0, out, iters, 3u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
#define ROW(NV) printf("waves/SIMD %d  %d vector instructions (+4 operand ANDs) per MFMA: mfma only %.3f  vector only %.3f  both %.3f ms\n", w, NV, run<NV, 0>(w, out), run<NV, 1>(w, out), run<NV, 2>(w, out));
int main() {
  float* out; hipMalloc(&out, 256 * 4 * 4 * 64 * 4 * 2);
  for (int w = 1; w <= 4; w *= 2) { ROW(0) ROW(2) ROW(4) ROW(6) ROW(8) }
  return 0;
}

// Probe: can VALU instructions of the SAME wave issue in the shadow of bf16 MFMAs (v_mfma_f32_32x32x16_bf16, 8 passes)?
// One wave per SIMD (and two), per iteration 12 dependent MFMAs and NV independent VALU fmas, either back to back (MFMAs then VALU)
// or interleaved with sched_group_barrier (1 MFMA : NV/12 VALU).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_overlap_probe.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

template <int NV, int MODE>   // MODE 0: MFMA only, 1: VALU only, 2: serial, 3: interleaved
__global__ __launch_bounds__(64) void probe(float* out, int iters) {
  floatx16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  uintx4 au = {threadIdx.x * 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  bf16x8 a = __builtin_bit_cast(bf16x8, au);
  float v[6]; for (int i = 0; i < 6; ++i) v[i] = threadIdx.x * 1e-3f + i;
  const float b = 1.0001f, c = 0.5f;
  for (int it = 0; it < iters; ++it) {
    if (MODE != 1) {
#pragma unroll
      for (int m = 0; m < 12; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
    }
    if (MODE != 0) {
#pragma unroll
      for (int u = 0; u < NV; ++u) v[u % 6] = __builtin_fmaf(v[u % 6], b, c);
    }
    if (MODE == 3) {
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, NV / 12, 0);  // NV/12 VALU
      }
    }
    if (MODE == 2) {
      __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f; for (int r = 0; r < 16; ++r) s += acc[r];
  for (int i = 0; i < 6; ++i) s += v[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NV, int MODE>
static float run(int waves_per_simd, float* out) {
  const int iters = 4000;
  dim3 grid(256 * 4 * waves_per_simd);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NV, MODE>), grid, dim3(64), 0, 0, out, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<NV, MODE>), grid, dim3(64), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float* out; hipMalloc(&out, 256 * 4 * 4 * 64 * 4 * 2);
  for (int w = 1; w <= 2; ++w) {
    printf("waves/SIMD %d  NV=72 : mfma %.3f  valu %.3f  serial %.3f  interleaved %.3f ms\n", w, run<72, 0>(w, out), run<72, 1>(w, out),
           run<72, 2>(w, out), run<72, 3>(w, out));
    printf("waves/SIMD %d  NV=48 : mfma %.3f  valu %.3f  serial %.3f  interleaved %.3f ms\n", w, run<48, 0>(w, out), run<48, 1>(w, out),
           run<48, 2>(w, out), run<48, 3>(w, out));
    printf("waves/SIMD %d  NV=96 : mfma %.3f  valu %.3f  serial %.3f  interleaved %.3f ms\n", w, run<96, 0>(w, out), run<96, 1>(w, out),
           run<96, 2>(w, out), run<96, 3>(w, out));
  }
  return 0;
}

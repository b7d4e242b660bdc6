// Probe for a split-bf16 convolution (each fp32 operand as three bf16 pieces, six v_mfma_f32_32x32x16_bf16 products accumulated in
// fp32): (1) rate of the bf16 MFMA alone, (2) does VALU work of another wave overlap with it, (3) one K=32 slab of a 32x32 tile as
// 16 fp32 MFMAs vs 12 bf16 MFMAs with the A operand split in registers every slab vs with both operands pre-split.
//   hipcc -O3 --offload-arch=gfx950 tools/bf16_split_probe.hip -o bf16_probe && ./bf16_probe
// Measured on MI355X (round 2; 4000 slabs per wave, ms): 1 wave/SIMD  fp32 1.726 | split in registers 1.514 | pre-split 0.691
//                                                        2 waves/SIMD fp32 3.441 | 2.437 | 1.320     4 waves/SIMD  fp32 10.3 | 4.887 | 2.727
// and a bf16-MFMA wave plus a VALU wave on one SIMD take close to the SUM of their times (0.469 + 1.0 -> 1.473 ms): the 88 VALU
// instructions that split 16 values cost as much issue time as the 12 MFMAs they feed.  So the 2.5x of the six-product form is
// only there with operands split ONCE (by the producing kernel, or while staging a tile with its halo into LDS) -- not per tap.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void bf16_only(float* out, int iters) {
  floatx16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  uintx4 au = {threadIdx.x * 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  bf16x8 a = __builtin_bit_cast(bf16x8, au), b = a;
  for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  float s = 0.f; for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int VPM>
__global__ __launch_bounds__(64) void split_waves(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, s = 0.f;
  if ((blockIdx.x & 1) == 0) {
    floatx16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    uintx4 au = {threadIdx.x * 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    bf16x8 av = __builtin_bit_cast(bf16x8, au);
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, av, acc, 0, 0, 0);
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
// three-way truncation split of 8 floats -> 3 bf16x8 planes (hi, mid, lo): x = hi + mid + lo + O(2^-24 x)
__device__ inline void split8(const float* x, bf16x8& h, bf16x8& m, bf16x8& l) {
  unsigned hu[4], mu[4], lu[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = x[2 * j], x1 = x[2 * j + 1];
    const unsigned h0 = __float_as_uint(x0) & 0xffff0000u, h1 = __float_as_uint(x1) & 0xffff0000u;
    const float r0 = x0 - __uint_as_float(h0), r1 = x1 - __uint_as_float(h1);
    const unsigned m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
    const float q0 = r0 - __uint_as_float(m0), q1 = r1 - __uint_as_float(m1);
    hu[j] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
    mu[j] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
    lu[j] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
  }
  uintx4 hv = {hu[0], hu[1], hu[2], hu[3]}, mv = {mu[0], mu[1], mu[2], mu[3]}, lv = {lu[0], lu[1], lu[2], lu[3]};
  h = __builtin_bit_cast(bf16x8, hv); m = __builtin_bit_cast(bf16x8, mv); l = __builtin_bit_cast(bf16x8, lv);
}
// per iteration: one K=32 slab of a 32x32 tile.  A split in registers every iteration (values change), B pre-split (constant).
__global__ __launch_bounds__(64) void emu_slab(float* out, const float* in, int iters) {
  floatx16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float x[16];
  for (int j = 0; j < 16; ++j) x[j] = in[threadIdx.x * 16 + j];
  bf16x8 bh[2], bm[2], bl[2];
  split8(x, bh[0], bm[0], bl[0]); split8(x + 8, bh[1], bm[1], bl[1]);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 h, m, l;
      split8(x + 8 * s, h, m, l);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l, bh[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h, bl[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m, bm[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m, bh[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h, bm[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h, bh[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] += acc[j & 15] * 1e-30f + 1e-3f;      // new operand values every iteration (cheap dependency)
  }
  float s = 0.f; for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ __launch_bounds__(64) void emu_slab_presplit(float* out, const float* in, int iters) {   // A pre-split too (e.g. read from LDS planes)
  floatx16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float x[16];
  for (int j = 0; j < 16; ++j) x[j] = in[threadIdx.x * 16 + j];
  bf16x8 bh[2], bm[2], bl[2];
  split8(x, bh[0], bm[0], bl[0]); split8(x + 8, bh[1], bm[1], bl[1]);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[s], bh[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[s], bl[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm[s], bm[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm[s], bh[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[s], bm[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[s], bh[s], acc, 0, 0, 0);
    }
  }
  float s = 0.f; for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ __launch_bounds__(64) void f32_slab(float* out, const float* in, int iters) {
  floatx16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float x[16];
  for (int j = 0; j < 16; ++j) x[j] = in[threadIdx.x * 16 + j];
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[j], x[15 - j], acc, 0, 0, 0);
  }
  float s = 0.f; for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float *out, *in; hipMalloc(&out, 1 << 24); hipMalloc(&in, 1 << 20); hipMemset(in, 0x3c, 1 << 20);
  const int it = 20000;
  for (int wps : {1, 2, 4}) {
    const int blocks = 1024 * wps;
    printf("%d waves/SIMD: bf16 32x32x16 MFMA only (20000) %.3f ms", wps, timeit([&] { hipLaunchKernelGGL(bf16_only, dim3(blocks), dim3(64), 0, 0, out, it); }));
    if (wps > 1) {
      printf(" | half bf16-MFMA waves + half VALU waves: 8 VALU/iter %.3f ms", timeit([&] { hipLaunchKernelGGL((split_waves<8>), dim3(blocks), dim3(64), 0, 0, out, it); }));
      printf(" | 32 VALU/iter %.3f ms", timeit([&] { hipLaunchKernelGGL((split_waves<32>), dim3(blocks), dim3(64), 0, 0, out, it); }));
    }
    printf("\n");
  }
  const int it2 = 4000;
  for (int wps : {1, 2, 4}) {
    const int blocks = 1024 * wps;
    printf("%d waves/SIMD, %d K=32 slabs per wave: fp32 MFMA %.3f ms", wps, it2, timeit([&] { hipLaunchKernelGGL(f32_slab, dim3(blocks), dim3(64), 0, 0, out, in, it2); }));
    printf(" | 6 bf16 products, A split in registers %.3f ms", timeit([&] { hipLaunchKernelGGL(emu_slab, dim3(blocks), dim3(64), 0, 0, out, in, it2); }));
    printf(" | 6 bf16 products, operands pre-split %.3f ms\n", timeit([&] { hipLaunchKernelGGL(emu_slab_presplit, dim3(blocks), dim3(64), 0, 0, out, in, it2); }));
  }
  return 0;
}

// What an LDS-DMA buffer load (buffer_load_dwordx4 ... lds) writes for lanes whose offset is out of the descriptor's range: zeros, or nothing?
// (The strip weight gradient's rows-through-LDS form relies on the answer for its image padding.)  Also: a wave's own counted vmcnt orders its own
// ds_read behind its own DMA.   hipcc -O3 --offload-arch=gfx950 tools/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __amdgpu_buffer_rsrc_t mi_rsrc;
__global__ __launch_bounds__(64) void probe(const float* __restrict__ src, float* __restrict__ out, int valid_bytes) {
  __shared__ __attribute__((aligned(16))) float buf[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) buf[i] = -7.f;           // sentinel
  __syncthreads();
  const mi_rsrc r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, valid_bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)buf, 16, lane * 16, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int i = lane; i < 256; i += 64) out[i] = buf[i];
}
int main() {
  std::vector<float> h(256);
  for (int i = 0; i < 256; ++i) h[i] = 1.f + i;
  float *d, *o;
  hipMalloc(&d, 1024); hipMalloc(&o, 1024);
  hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
  for (int valid : {1024, 400, 0}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, valid);
    std::vector<float> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int ok = 0, zero = 0, sentinel = 0, other = 0;
    for (int i = 0; i < 256; ++i) {
      if (i * 4 < valid) ok += r[i] == h[i];
      else if (r[i] == 0.f) ++zero; else if (r[i] == -7.f) ++sentinel; else ++other;
    }
    printf("valid bytes %4d: in-range words correct %d / %d; out-of-range words: zero %d, untouched (sentinel) %d, other %d\n", valid, ok, valid / 4, zero, sentinel, other);
  }
  return 0;
}

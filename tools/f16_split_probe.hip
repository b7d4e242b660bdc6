// Probe for the two-plane fp16 operand form (f16_split.h): (1) the split x*s = h + l built from v_fma_mixlo/mixhi_f16, v_fma_mix_f32 and
// v_cvt_pk_f16_f32 -- is h the round-to-nearest fp16 of x*s, is the residual exact; (2) does v_mfma_f32_32x32x16_f16 keep fp16 DENORMAL
// inputs (the low plane of small values lives there) or flush them to zero.  Build: hipcc -O3 --offload-arch=gfx950 tools/f16_split_probe.hip
// -o build/f16_split_probe; prints one line per check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void f16_split2(float v0, float v1, float s, unsigned& h, unsigned& l) {
  unsigned hh;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hh) : "v"(v0), "s"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hh) : "v"(v1), "s"(s));
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(v0), "s"(s), "v"(hh));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(v1), "s"(s), "v"(hh));
  h = hh;
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(floatx2{r0, r1}, halfx2));
}
__global__ void split_k(const float* x, unsigned* o, float s, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h, l;
  f16_split2(x[2 * i], x[2 * i + 1], s, h, l);
  o[2 * i] = h; o[2 * i + 1] = l;
}
// C = A B with A[m][k] = a for k == 0 (else 0), B[k][n] = b for k == 0: every C entry = a * b
__global__ void mfma_k(unsigned short abits, unsigned short bbits, float* out) {
  const int lane = threadIdx.x;
  _Float16 a = __builtin_bit_cast(_Float16, abits), b = __builtin_bit_cast(_Float16, bbits);
  halfx8 av = {}, bv = {};
  if (lane < 32) { av[0] = a; bv[0] = b; }          // k = 0 lives in element 0 of lanes 0..31
  floatx16 acc = {};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
  if (lane == 0) out[0] = acc[0];
}
static float h2f(unsigned short b) { _Float16 h; memcpy(&h, &b, 2); return (float)h; }
int main() {
  const int n = 1 << 16;
  std::vector<float> x(2 * n);
  unsigned seed = 12345u;
  for (auto& v : x) { seed = seed * 1664525u + 1013904223u; const float u = (float)(seed >> 8) / 16777216.f; seed = seed * 1664525u + 1013904223u;
    const int e = (int)(seed >> 27) - 24; v = ldexpf(u * 2.f - 1.f, e); }
  float *dx; unsigned* dout; hipMalloc(&dx, 8 * n); hipMalloc(&dout, 8 * n);
  hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice);
  const float s = 256.f;
  split_k<<<n / 256, 256>>>(dx, dout, s, n);
  std::vector<unsigned> o(2 * n);
  hipMemcpy(o.data(), dout, 8 * n, hipMemcpyDeviceToHost);
  int bad_h = 0, bad_l = 0; double worst = 0;
  for (int i = 0; i < n; ++i)
    for (int p = 0; p < 2; ++p) {
      const float xs = x[2 * i + p] * s;
      const unsigned short hb = (unsigned short)(o[2 * i] >> (16 * p)), lb = (unsigned short)(o[2 * i + 1] >> (16 * p));
      const _Float16 hr = (_Float16)xs;                      // host RNE conversion
      unsigned short hrb; memcpy(&hrb, &hr, 2);
      if (hrb != hb) ++bad_h;
      const float r = xs - h2f(hb);
      const _Float16 lr = (_Float16)r; unsigned short lrb; memcpy(&lrb, &lr, 2);
      if (lrb != lb) ++bad_l;
      const double err = fabs((double)xs - (double)h2f(hb) - (double)h2f(lb));
      if (xs != 0.f && err / fabs(xs) > worst && fabsf(xs) >= 0.125f) worst = err / fabs(xs);
    }
  printf("split: %d values, h != RNE fp16(x*s): %d, l != RNE fp16(x*s - h): %d, worst |x*s - h - l| / |x*s| for |x*s| >= 2^-3: %.3e (2^-22 = %.3e)\n",
         2 * n, bad_h, bad_l, worst, ldexp(1.0, -22));
  float* dres; hipMalloc(&dres, 4);
  const unsigned short cases[][2] = {{0x3c00, 0x3c00}, {0x0001, 0x7800}, {0x03ff, 0x7800}, {0x0001, 0x0001}, {0x0400, 0x0400}};
  const char* names[] = {"1 * 1", "min denormal (2^-24) * 2^15", "max denormal * 2^15", "min denormal squared (2^-48)", "min normal squared (2^-28)"};
  for (int c = 0; c < 5; ++c) {
    mfma_k<<<1, 64>>>(cases[c][0], cases[c][1], dres);
    float r; hipMemcpy(&r, dres, 4, hipMemcpyDeviceToHost);
    printf("mfma f16: %-32s = %.9e  (exact %.9e)\n", names[c], r, (double)h2f(cases[c][0]) * (double)h2f(cases[c][1]));
  }
  return 0;
}

// One-kernel sweeps of the MAML-TRPO Fisher-vector product (reference core_functions/rl.py:417-418: trpo.hessian_vector_product of
// the mean KL inside conjugate gradient, 11 products per meta-iteration).  Round 2 ran one product as ~34 launches of per-layer
// dense kernels (every [rows, 100] intermediate a round trip through HBM, 0.86 ms, 10 % of the fp32 matrix peak); here a product
// is three sweeps over the stored passes plus three small folds:
//     A   hv_t = H_t v          over the support pass at theta          -> u_t = v - lr hv_t
//     B   w_t  = F_t u_t        over the query pass at theta'_t         (Gaussian Fisher: tangent forward, cotangent, backward)
//     C   hv_t = H_t w_t        over the support pass                   -> out = mean_t (w_t - lr hv_t) + damping v
// A sweep takes a slab of 32 rows through the WHOLE chain inside one workgroup: tangent forward of the hidden layer, the head and
// the Gaussian tangent on the vector unit, tangent backward and the weight-gradient products -- both 100 x 100 weight matrices
// (the pass's and the direction's) stay in LDS for the workgroup's lifetime (80 KB), the slab's activations live in LDS / registers
// and nothing but the per-workgroup gradient partial is written.  Matrix work on v_mfma_f32_32x32x2_f32 (exact fp32):
//   * wave w of the 4 owns output columns [32w, 32w+32) of every product; the reduction index is dealt to the lane halves as
//     [0, KH0) / [KH0, H) so a lane's A operands (and the forward products' B operands) are 16-byte LDS reads;
//   * the weight gradient dW2 (100 x 100, the sum over rows) accumulates in registers across all slabs of a task (4 x 16
//     accumulators per lane), one deterministic partial per (workgroup, task), folded in a fixed order.
// The 2-wide layers (states -> hidden, hidden -> actions) and the per-row Gaussian formulas are vector work.
// ReLU policies with equal hidden widths H % 8 == 4 or 0, H <= 128, S <= 4, A <= 6 (DiagNormalPolicy defaults: 2-100-100-2); other
// shapes keep the per-layer path.
#pragma once
#include "mi_common.h"

#define SW_MAX_S 4
#define SW_MAX_A 6

struct SweepArgs {
  const float* x;        // states of the pass        [T][B][S]
  const float* act;      // actions                    [T][B][A]   (HVP)
  const float* h1;       // stored activations         [T][B][H]
  const float* h2;
  const float* mu;       // [T][B][A]                               (HVP)
  const float* coef;     // dL/dlogp per row  [T][B]                (HVP)
  const float* dmu;      // primal cotangents [T][B][A], [T][B][H]  (HVP)
  const float* d2;
  const int32_t* count;  // [T] valid rows (null: B)
  const float* theta; size_t tstride;   // parameters of the pass (stride 0 = shared by all tasks)
  const float* dir; size_t dstride;     // direction of the product
  float* partial;        // [T][slots][P]
  int T, B, S, A, spt, spw, slots;      // spt = slabs per task, spw = slabs per workgroup
  int o_sigma, o_w1, o_b1, o_w2, o_b2, o_w3, o_b3, P;
};

__device__ __forceinline__ floatx4 lds4(const float* p) { return *reinterpret_cast<const floatx4*>(p); }

template <int H, bool HVP>
__global__ __launch_bounds__(256) void policy_sweep_kernel(SweepArgs a) {
  static_assert(H % 4 == 0 && H <= 128 && (H % 8 == 0 || H % 8 == 4), "hidden width");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HH = H * H, SL = 32 * H, KH0 = ((H + 7) / 8) * 4, NG = KH0 / 4, MT = (H + 31) / 32;
  const int S = a.S, A = a.A;
  float* W2s = lds;                     // the pass's W2 [o][k]
  float* W2d = W2s + HH;                // the direction's W2
  float* h1s = W2d + HH;                // slab arrays [32][H]
  float* h1d = h1s + SL;
  float* h2s = h1d + SL;
  float* h2d = h2s + SL;                // tangent of h2; later r2 (the tangent-backward cotangent of z2)
  float* d2s = h2d + SL;                // primal dz2 (HVP)
  float* sm = d2s + SL + 32;            // 32 floats of slack: the last M / N tile of the dW2 product reads past a slab array
  float* xs = sm;            sm += 32 * SW_MAX_S;
  float* acts = sm;          sm += 32 * SW_MAX_A;
  float* mus = sm;           sm += 32 * SW_MAX_A;
  float* dmus = sm;          sm += 32 * SW_MAX_A;
  float* rdmus = sm;         sm += 32 * SW_MAX_A;
  float* muds = sm;          sm += 32 * SW_MAX_A;
  float* coefs = sm;         sm += 32;
  float* red = sm;           sm += SW_MAX_A * 8 * 32;
  float* W1d = sm;           sm += H * SW_MAX_S;
  float* b1d = sm;           sm += H;
  float* b2d = sm;           sm += H;
  float* W3s = sm;           sm += SW_MAX_A * H;
  float* W3d = sm;           sm += SW_MAX_A * H;
  float* b3d = sm;           sm += 8;
  float* rho = sm;           sm += 8;
  float* rhod = sm;          sm += 8;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hh = lane >> 5;
  const int B = a.B, spt = a.spt;
  const int slab0 = blockIdx.x * a.spw, slab1 = min(slab0 + a.spw, a.T * spt);

  // ---- accumulators that live across the slabs of one task
  floatx16 accW2[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) accW2[m][r] = 0.f;
  constexpr int NW3 = (SW_MAX_A * H + 255) / 256;
  float accW3[NW3], accb3 = 0.f, accb2 = 0.f, accb1 = 0.f, accW1[SW_MAX_S], accrho[SW_MAX_A];
#pragma unroll
  for (int u = 0; u < NW3; ++u) accW3[u] = 0.f;
#pragma unroll
  for (int s = 0; s < SW_MAX_S; ++s) accW1[s] = 0.f;
#pragma unroll
  for (int d = 0; d < SW_MAX_A; ++d) accrho[d] = 0.f;

  auto load_weights = [&](int t) {
    const float* th = a.theta + (size_t)t * a.tstride;
    const float* dv = a.dir + (size_t)t * a.dstride;
#pragma unroll 4
    for (int e = tid; e < HH; e += 256) { W2s[e] = th[a.o_w2 + e]; W2d[e] = dv[a.o_w2 + e]; }
    for (int e = tid; e < H * S; e += 256) W1d[e] = dv[a.o_w1 + e];
    for (int e = tid; e < H; e += 256) { b1d[e] = dv[a.o_b1 + e]; b2d[e] = dv[a.o_b2 + e]; }
    for (int e = tid; e < A * H; e += 256) { W3s[e] = th[a.o_w3 + e]; W3d[e] = dv[a.o_w3 + e]; }
    if (tid < A) { b3d[tid] = dv[a.o_b3 + tid]; rho[tid] = th[a.o_sigma + tid]; rhod[tid] = dv[a.o_sigma + tid]; }
  };

  // one partial [P] per (workgroup, task): slot = this workgroup's position among the workgroups that touch the task
  auto flush = [&](int t) {
    const int slot = blockIdx.x - (t * spt) / a.spw;
    float* pv = a.partial + ((size_t)t * a.slots + slot) * a.P;
    const int icol = 32 * wave + n;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (o < H && icol < H) pv[a.o_w2 + o * H + icol] = accW2[m][r];
        accW2[m][r] = 0.f;
      }
#pragma unroll
    for (int u = 0; u < NW3; ++u) {
      const int idx = tid + 256 * u;
      if (idx < A * H) pv[a.o_w3 + idx] = accW3[u];
      accW3[u] = 0.f;
    }
    if (tid < A) pv[a.o_b3 + tid] = accb3;
    if (tid < H) pv[a.o_b2 + tid] = accb2;
    accb3 = 0.f; accb2 = 0.f;
    // W1 / b1 partials sit per lane half (16 rows each), rho partials per row-thread: fold through LDS (the slab arrays are free)
    float* t1 = h1s;                                   // [2][H][S + 1]
    if (icol < H) {
#pragma unroll
      for (int s = 0; s < SW_MAX_S; ++s) if (s < S) t1[(hh * H + icol) * (S + 1) + s] = accW1[s];
      t1[(hh * H + icol) * (S + 1) + S] = accb1;
    }
    float* t2 = h1d;                                   // [32][A]
    if (tid < 32) {
#pragma unroll
      for (int d = 0; d < SW_MAX_A; ++d) if (d < A) t2[tid * A + d] = accrho[d];
    }
    __syncthreads();
    if (tid < H) {
      for (int s = 0; s < S; ++s) pv[a.o_w1 + tid * S + s] = t1[tid * (S + 1) + s] + t1[(H + tid) * (S + 1) + s];
      pv[a.o_b1 + tid] = t1[tid * (S + 1) + S] + t1[(H + tid) * (S + 1) + S];
    }
    if (tid < A) {
      float s = 0.f;
      for (int r = 0; r < 32; ++r) s += t2[r * A + tid];
      pv[a.o_sigma + tid] = s;
    }
    __syncthreads();
    accb1 = 0.f;
#pragma unroll
    for (int s = 0; s < SW_MAX_S; ++s) accW1[s] = 0.f;
#pragma unroll
    for (int d = 0; d < SW_MAX_A; ++d) accrho[d] = 0.f;
  };

  int cur = -1;
  for (int slab = slab0; slab < slab1; ++slab) {
    const int t = slab / spt, row0 = (slab - t * spt) * 32;
    if (t != cur) {
      if (cur >= 0) flush(cur);
      if (cur < 0 || a.tstride != 0 || a.dstride != 0) load_weights(t);
      cur = t;
    }
    const int cnt = a.count ? a.count[t] : B;
    const int nv = min(32, B - row0);                 // rows of this slab inside the padded batch
    const size_t rbase = (size_t)t * B + row0;
    // ---- stage the slab: stored activations (16-byte coalesced copies; rows past the batch are zero) and the per-row scalars
    {
      const floatx4 z4 = {0.f, 0.f, 0.f, 0.f};
      const float* g1 = a.h1 + rbase * H;
      const float* g2 = a.h2 + rbase * H;
      const float* g3 = HVP ? a.d2 + rbase * H : nullptr;
#pragma unroll 1
      for (int e = tid * 4; e < SL; e += 1024) {
        const bool ok = e < nv * H;
        *reinterpret_cast<floatx4*>(h1s + e) = ok ? *reinterpret_cast<const floatx4*>(g1 + e) : z4;
        *reinterpret_cast<floatx4*>(h2s + e) = ok ? *reinterpret_cast<const floatx4*>(g2 + e) : z4;
        if (HVP) *reinterpret_cast<floatx4*>(d2s + e) = ok ? *reinterpret_cast<const floatx4*>(g3 + e) : z4;
      }
      if (tid < 32 * S) xs[tid] = (tid < nv * S) ? a.x[rbase * S + tid] : 0.f;
      if (HVP) {
        if (tid < 32 * A) {
          const bool ok = tid < nv * A;
          acts[tid] = ok ? a.act[rbase * A + tid] : 0.f;
          mus[tid] = ok ? a.mu[rbase * A + tid] : 0.f;
          dmus[tid] = ok ? a.dmu[rbase * A + tid] : 0.f;
        }
        if (tid < 32) coefs[tid] = (tid < nv && row0 + tid < cnt) ? a.coef[rbase + tid] : 0.f;
      }
    }
    __syncthreads();
    // ---- tangent of the first hidden layer (S-wide: vector work): h1d = [h1 > 0] (x W1d^T + b1d)
#pragma unroll 1
    for (int e = tid; e < SL; e += 256) {
      const int r = e / H, o = e - r * H;
      float z = b1d[o];
      for (int s = 0; s < S; ++s) z = fmaf(xs[r * S + s], W1d[o * S + s], z);
      h1d[e] = h1s[e] > 0.f ? z : 0.f;
    }
    __syncthreads();
    // ---- tangent of the second hidden layer on the matrix pipe: z2d = h1 W2d^T + h1d W2^T + b2d, h2d = [h2 > 0] z2d
    {
      floatx16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int ocol = min(32 * wave + n, H - 1);
#pragma unroll 1
      for (int term = 0; term < 2; ++term) {
        const float* arow = (term == 0 ? h1s : h1d) + n * H + hh * KH0;
        const float* brow = (term == 0 ? W2d : W2s) + ocol * H + hh * KH0;
        floatx4 av = lds4(arow), bv = lds4(brow);          // operands one group of 4 k ahead of the MFMAs that use them
#pragma unroll 1
        for (int j = 0; j < NG; ++j) {
          floatx4 an = av, bn = bv;
          if (j + 1 < NG) { an = lds4(arow + 4 * j + 4); bn = lds4(brow + 4 * j + 4); }
          if (KH0 + 4 * j >= H && hh) { av = floatx4{0.f, 0.f, 0.f, 0.f}; bv = av; }     // upper half's k past the end
#pragma unroll
          for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bv[c], acc, 0, 0, 0);
          av = an; bv = bn;
        }
      }
      const int o = 32 * wave + n;
      if (o < H) {
        const float bb = b2d[o];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
          h2d[row * H + o] = h2s[row * H + o] > 0.f ? acc[r] + bb : 0.f;
        }
      }
    }
    __syncthreads();
    // ---- head tangent (A-wide): mud = h2 W3d^T + h2d W3^T + b3d; partial dot products per (row, eighth of the columns)
    {
      const int r = tid & 31, q = tid >> 5;
#pragma unroll 1
      for (int d = 0; d < A; ++d) {
        float s = 0.f;
#pragma unroll 4
        for (int k = q; k < H; k += 8) s = fmaf(h2s[r * H + k], W3d[d * H + k], fmaf(h2d[r * H + k], W3s[d * H + k], s));
        red[(d * 8 + q) * 32 + r] = s;
      }
    }
    __syncthreads();
    if (tid < 32 * A) {
      const int r = tid & 31, d = tid >> 5;
      float m = b3d[d];
      for (int q = 0; q < 8; ++q) m += red[(d * 8 + q) * 32 + r];
      muds[r * A + d] = m;
    }
    __syncthreads();
    // ---- Gaussian part, one thread per row: the cotangent of mu that the tangent backward starts from (and the sigma slots)
    if (tid < 32) {
      const bool valid = tid < nv && row0 + tid < cnt;
      const float invD = 1.f / (float)A;
      if (HVP) {
        const float c = coefs[tid];                    // 0 on padding rows
#pragma unroll
        for (int d = 0; d < SW_MAX_A; ++d) {
          if (d >= A) break;
          const float rp = rho[d];
          const bool live = rp > LOG_EPS;
          const float rr = fmaxf(rp, LOG_EPS), sg = expf(rr), iv = 1.f / (sg * sg);
          const float rd = live ? rhod[d] : 0.f;
          const float df = acts[tid * A + d] - mus[tid * A + d];
          const float md = muds[tid * A + d];
          rdmus[tid * A + d] = c * invD * (-md * iv - 2.f * df * rd * iv);
          if (live) accrho[d] += c * invD * (-2.f * df * md * iv - 2.f * df * df * rd * iv);
        }
      } else {
        const float invB = 1.f / (float)cnt;
        for (int d = 0; d < A; ++d) {
          const float rr = fmaxf(rho[d], LOG_EPS), sg = expf(rr);
          rdmus[tid * A + d] = valid ? muds[tid * A + d] * invB * invD / (sg * sg) : 0.f;
        }
      }
    }
    __syncthreads();
    // ---- head weight gradient (A x H outputs, 32 rows each) and its bias
#pragma unroll
    for (int u = 0; u < NW3; ++u) {
      const int idx = tid + 256 * u;
      if (idx < A * H) {
        const int d = idx / H, k = idx - d * H;
        float s = 0.f;
#pragma unroll 4
        for (int r = 0; r < 32; ++r) {
          s = fmaf(rdmus[r * A + d], h2s[r * H + k], s);
          if (HVP) s = fmaf(dmus[r * A + d], h2d[r * H + k], s);
        }
        accW3[u] += s;
      }
    }
    if (tid < A) {
      float s = 0.f;
      for (int r = 0; r < 32; ++r) s += rdmus[r * A + tid];
      accb3 += s;
    }
    __syncthreads();                                   // h2d is about to become r2
    // ---- r2 = [h2 > 0] (rdmu W3 + dmu W3d)   (A-wide reduction: vector work)
#pragma unroll 1
    for (int e = tid; e < SL; e += 256) {
      const int r = e / H, o = e - r * H;
      float v = 0.f;
      for (int d = 0; d < A; ++d) {
        v = fmaf(rdmus[r * A + d], W3s[d * H + o], v);
        if (HVP) v = fmaf(dmus[r * A + d], W3d[d * H + o], v);
      }
      h2d[e] = h2s[e] > 0.f ? v : 0.f;
    }
    __syncthreads();
    const float* r2 = h2d;
    if (tid < H) {
      float s = 0.f;
#pragma unroll 4
      for (int r = 0; r < 32; ++r) s += r2[r * H + tid];
      accb2 += s;
    }
    // ---- r1 = [h1 > 0] (r2 W2 + d2 W2d) on the matrix pipe; its products with the states (W1, b1 gradients) from the accumulators
    {
      floatx16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int icol = min(32 * wave + n, H - 1);
#pragma unroll 1
      for (int term = 0; term < (HVP ? 2 : 1); ++term) {
        const float* arow = (term == 0 ? r2 : d2s) + n * H + hh * KH0;
        const float* bcol = (term == 0 ? W2s : W2d) + icol + hh * KH0 * H;
        floatx4 av = lds4(arow), bv;
#pragma unroll
        for (int c = 0; c < 4; ++c) bv[c] = bcol[c * H];
#pragma unroll 1
        for (int j = 0; j < NG; ++j) {
          floatx4 an = av, bn = bv;
          if (j + 1 < NG) {
            an = lds4(arow + 4 * j + 4);
            const bool pastn = KH0 + 4 * j + 4 >= H && hh;           // the upper half's last group does not exist: stay in bounds
            const float* bp = pastn ? bcol : bcol + (4 * j + 4) * H;
#pragma unroll
            for (int c = 0; c < 4; ++c) bn[c] = bp[c * H];
          }
          if (KH0 + 4 * j >= H && hh) av = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bv[c], acc, 0, 0, 0);
          av = an; bv = bn;
        }
      }
      const int i = 32 * wave + n;
      if (i < H) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
          const float v = h1s[row * H + i] > 0.f ? acc[r] : 0.f;
          accb1 += v;
#pragma unroll
          for (int s = 0; s < SW_MAX_S; ++s) if (s < S) accW1[s] = fmaf(v, xs[row * S + s], accW1[s]);
        }
      }
    }
    // ---- dW2[o][i] += sum_rows r2[row][o] h1[row][i] + d2[row][o] h1d[row][i]: M = o (MT tiles), N = this wave's columns, K = rows
    {
      const int icol = 32 * wave + n;                  // columns past H read the next row: finite, never stored
#pragma unroll 1
      for (int term = 0; term < (HVP ? 2 : 1); ++term) {
        const float* am = (term == 0 ? r2 : d2s) + (16 * hh) * H + n;
        const float* bm = (term == 0 ? h1s : h1d) + (16 * hh) * H + icol;
#pragma unroll 2
        for (int s = 0; s < 16; ++s) {
          const float bv = bm[s * H];
          float avv[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) avv[m] = am[s * H + 32 * m];
#pragma unroll
          for (int m = 0; m < MT; ++m) accW2[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(avv[m], bv, accW2[m], 0, 0, 0);
        }
      }
    }
    __syncthreads();                                   // the next slab's staging overwrites the slab arrays
  }
  if (cur >= 0) flush(cur);
}

// Fold the per-workgroup partials of a sweep in slot order (deterministic) and finish the phase:
//   mode 0 (after A): out[t][p] = v[p] - lr sum                          (u_t)
//   mode 1 (after B): out[t][p] = sum; sigma slots: the Gaussian Fisher's 2 u / D where sigma is not clamped   (w_t)
//   mode 2 (after C): out[p] = mean_t (w[t][p] - lr sum_t) + damping v[p]
struct FoldArgs {
  const float* partial; int slots, spt, spw, T, P;
  const float* v;          // [P]
  const float* w;          // [T][P]   (mode 2)
  const float* thetap;     // [T][P]   (mode 1: rho of theta')
  const float* u;          // [T][P]   (mode 1: direction)
  float lr, damping;
  int o_sigma, A;
  float* out;
  int mode;
};
__device__ __forceinline__ float sweep_fold_sum(const FoldArgs& f, int t, int p) {
  const int first = (t * f.spt) / f.spw, last = (t * f.spt + f.spt - 1) / f.spw;
  const float* pp = f.partial + (size_t)t * f.slots * f.P + p;
  float s = 0.f;
  for (int k = 0; k <= last - first; ++k) s += pp[(size_t)k * f.P];
  return s;
}
__global__ __launch_bounds__(256) void policy_sweep_fold_kernel(FoldArgs f) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= f.P) return;
  if (f.mode == 2) {
    float acc = 0.f;
    for (int t = 0; t < f.T; ++t) acc += f.w[(size_t)t * f.P + p] - f.lr * sweep_fold_sum(f, t, p);
    f.out[p] = acc * (1.f / (float)f.T) + f.damping * f.v[p];
    return;
  }
  const int t = blockIdx.y;
  const float s = sweep_fold_sum(f, t, p);
  if (f.mode == 0) {
    f.out[(size_t)t * f.P + p] = f.v[p] - f.lr * s;
  } else {
    float o = s;
    if (p >= f.o_sigma && p < f.o_sigma + f.A)
      o = f.thetap[(size_t)t * f.P + p] > LOG_EPS ? 2.f * f.u[(size_t)t * f.P + p] / (float)f.A : 0.f;
    f.out[(size_t)t * f.P + p] = o;
  }
}

template <int H>
static size_t policy_sweep_lds_bytes() {
  return ((size_t)2 * H * H + 5 * 32 * H + 32 + 32 * SW_MAX_S + 5 * 32 * SW_MAX_A + 32 + SW_MAX_A * 8 * 32 + H * SW_MAX_S + 2 * H +
          2 * SW_MAX_A * H + 24) * sizeof(float);
}

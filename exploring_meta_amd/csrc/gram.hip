// Input Gram matrix of the first ConvBlock: BatchNorm statistics of conv1 WITHOUT running conv1.
//
// conv1's output is linear in its weights: z[pix][co] = sum_a P[pix][a] w[a][co], P = the 3x3xCi0 zero-padded input patches
// (a = tap*Ci0 + c, the weight-row order).  The per-channel sums BatchNorm needs (reference BatchNorm2d in train mode inside
// ConvBlock.forward, core_functions/vision_models.py:188-193) are therefore quadratic forms of two small tables that depend
// on the images only:
//     s[a] = sum_pix P[pix][a]                G[a][b] = sum_pix P[pix][a] P[pix][b]
//     sum z  = w_c . s      sum z^2 = w_c^T G w_c      sum zd = wd_c . s      sum z zd = w_c^T G wd_c   (wd = tangent weights)
// A MAML task applies K inner steps plus K Hessian-vector products to the SAME support images with different weights, so G is
// computed once per meta-iteration and replaces 2K full conv-recompute passes (block1_kernel<STATS/TSTATS>) by 2K launches of
// a 28x28x32 quadratic form.  Everything is fp64: G on v_mfma_f64_16x16x4_f64 (P^T P, exact products of the fp32 pixels),
// the quadratic forms on the vector unit -- more accurate than summing fp32 conv outputs.
#include "mi_common.h"
#include "kernels.h"

typedef double doublex4 __attribute__((ext_vector_type(4)));

// NP = 9*Ci0 + 1 patch entries + the constant 1 (which yields s as the last row/column of G), padded to NT tiles of 16.
template <int CI0> struct GramDims {
  static constexpr int KP = 9 * CI0, NP = KP + 1, NT = (NP + 15) / 16, NG = NT * 16;
};

// One wave walks whole image rows, 4 pixels per MFMA step.  Lane l holds, for pixel (x0 + l/16) of the row, patch entries
// i = l%16 (+16 per tile): the same register is the A operand (P^T tile: row i, k = l/16) and the B operand (P tile: k, column i).
// Accumulator tile (ti, tj), register r of lane l = G[16 ti + 4 r + l/16][16 tj + l%16]  (the f64 16x16x4 C/D layout: rows
// interleave over the four 16-lane groups, unlike the f32 tiles).
template <int CI0>
__global__ __launch_bounds__(256) void input_gram_kernel(const float* __restrict__ x, int n, int H, int W, int rows_per_wave,
                                                         double* __restrict__ partial) {
  using D = GramDims<CI0>;
  constexpr int NT = D::NT, KP = D::KP;
  __shared__ double red[4 * NT * NT * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kpix = lane >> 4;
  const int task = blockIdx.y;
  const float* x_t = x + (size_t)task * n * H * W * CI0;
  // this lane's patch entries: (dy, dx, c) or the constant / padding
  int dy[NT], dx[NT], cc[NT], kind[NT];     // kind 0 = pixel value, 1 = constant one, 2 = zero padding
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int a = 16 * t + i;
    kind[t] = a < KP ? 0 : (a == KP ? 1 : 2);
    const int tap = a < KP ? a / CI0 : 0;
    dy[t] = tap / 3 - 1; dx[t] = tap % 3 - 1; cc[t] = a < KP ? a % CI0 : 0;
  }
  doublex4 acc[NT][NT];
#pragma unroll
  for (int ti = 0; ti < NT; ++ti)
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) acc[ti][tj] = doublex4{0.0, 0.0, 0.0, 0.0};

  const int nrows = n * H;
  const int row0 = (blockIdx.x * 4 + wave) * rows_per_wave;
  for (int row = row0; row < row0 + rows_per_wave && row < nrows; ++row) {
    const int img = row / H, y = row - img * H;
    const float* rowp[NT];
    bool rok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      rok[t] = kind[t] == 0 && (unsigned)(y + dy[t]) < (unsigned)H;
      rowp[t] = x_t + ((size_t)(img * H + y + dy[t]) * W + dx[t]) * CI0 + cc[t];
    }
    // patch values of one 4-pixel step; the next step's loads are issued before this step's MFMAs
    auto load_step = [&](int x0, double* v) {
      const int xx = x0 + kpix;
      const bool pv = xx < W;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bool inb = pv && rok[t] && (unsigned)(xx + dx[t]) < (unsigned)W;
        const float f = *(inb ? rowp[t] + (size_t)xx * CI0 : mi_zero_word);
        v[t] = kind[t] == 1 ? (pv ? 1.0 : 0.0) : (double)f;
      }
    };
    double v[NT], vn[NT];
    load_step(0, v);
    for (int x0 = 0; x0 < W; x0 += 4) {
      if (x0 + 4 < W) load_step(x0 + 4, vn);
      // G is symmetric: tiles below the diagonal are filled in by gram_reduce_kernel
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = ti; tj < NT; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[ti], v[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) v[t] = vn[t];
    }
  }
  // 4 waves -> one partial per workgroup, fixed order
#pragma unroll
  for (int ti = 0; ti < NT; ++ti)
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * NT + ti) * NT + tj) * 256 + r * 64 + lane] = acc[ti][tj][r];
  __syncthreads();
  double* out = partial + ((size_t)task * gridDim.x + blockIdx.x) * D::NG * D::NG;
  for (int e = tid; e < NT * NT * 256; e += 256) {
    const double s = red[e] + red[NT * NT * 256 + e] + red[2 * NT * NT * 256 + e] + red[3 * NT * NT * 256 + e];
    const int tile = e >> 8, r = (e >> 6) & 3, l = e & 63;
    const int ti = tile / NT, tj = tile - ti * NT;
    out[(16 * ti + 4 * r + (l >> 4)) * D::NG + 16 * tj + (l & 15)] = s;
  }
}

// G[task] = sum over workgroup partials in block order; entries in tiles below the diagonal come from their mirror image.
__global__ void gram_reduce_kernel(const double* __restrict__ partial, int nblk, int ng, double* __restrict__ g) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int ng2 = ng * ng;
  if (e >= ng2) return;
  const int row = e / ng, col = e - row * ng;
  const int src = (row >> 4) > (col >> 4) ? col * ng + row : e;
  const double* p = partial + (size_t)blockIdx.y * nblk * ng2 + src;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += p[(size_t)b * ng2];
  g[(size_t)blockIdx.y * ng2 + e] = s;
}

// One workgroup per task.  G (with the patch sums in row kp), this task's weights w and the direction v (= w when !tangent) go
// to LDS as fp64; work item (a, c) forms rowdot = sum_b w[b][c] G[b][a]; channel c then folds v[a][c] * rowdot and
// v[a][c] * s[a] over a in a fixed order.
//  tangent == 0: mu = sum z / M, rstd = 1/sqrt(E[z^2] - mu^2 + eps)                       (bn_finalize FIN_STATS)
//  tangent == 1: m1 = sum zd / M, m2 = sum zh zd / M with zh = (z - mu) rstd               (bn_finalize FIN_TSTATS)
__global__ __launch_bounds__(256) void gram_stats_kernel(const double* __restrict__ g, int ng, int kp, const float* __restrict__ w,
                                                          size_t wstride, const float* __restrict__ wd, size_t vstride, int co,
                                                          double inv_m, int tangent, float* __restrict__ out0,
                                                          float* __restrict__ out1, const float* __restrict__ mu_in,
                                                          const float* __restrict__ rstd_in) {
  extern __shared__ double sm[];
  double* gs = sm;                       // [ng][ng]
  double* ws = gs + ng * ng;             // [kp][co]
  double* vs = ws + kp * co;             // [kp][co]   (aliases ws when !tangent)
  double* qd = vs + (tangent ? kp * co : 0);   // [kp][co] partial products v[a][c] * rowdot(a, c)
  const int task = blockIdx.x, tid = threadIdx.x;
  for (int e = tid; e < ng * ng; e += 256) gs[e] = g[(size_t)task * ng * ng + e];
  for (int e = tid; e < kp * co; e += 256) {
    ws[e] = (double)w[(size_t)task * wstride + e];
    if (tangent) vs[e] = (double)wd[(size_t)task * vstride + e];
  }
  __syncthreads();
  if (!tangent) vs = ws;
  for (int e = tid; e < kp * co; e += 256) {
    const int a = e / co, c = e - a * co;
    double rowdot = 0.0;
    for (int b = 0; b < kp; ++b) rowdot = fma(ws[b * co + c], gs[b * ng + a], rowdot);
    qd[e] = vs[e] * rowdot;
  }
  __syncthreads();
  for (int c = tid; c < co; c += 256) {
    double lin = 0.0, quad = 0.0;
    for (int a = 0; a < kp; ++a) {
      lin = fma(vs[a * co + c], gs[a * ng + kp], lin);
      quad += qd[a * co + c];
    }
    if (!tangent) {
      const double mean = lin * inv_m;
      double var = quad * inv_m - mean * mean;
      var = var > 0.0 ? var : 0.0;
      out0[(size_t)task * co + c] = (float)mean;
      out1[(size_t)task * co + c] = (float)(1.0 / sqrt(var + MI_BN_EPS));
    } else {
      const double mu = (double)mu_in[(size_t)task * co + c], rs = (double)rstd_in[(size_t)task * co + c];
      out0[(size_t)task * co + c] = (float)(lin * inv_m);
      out1[(size_t)task * co + c] = (float)(rs * (quad - mu * lin) * inv_m);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
static int gram_ng(int ci) { return ci == 3 ? GramDims<3>::NG : GramDims<1>::NG; }
static const int kGramRowsPerWave = 8;

int gram_blocks_per_task(int n, int h) { return ceil_div(n * h, 4 * kGramRowsPerWave); }
size_t gram_partial_doubles(int tasks, int n, int h, int ci) {
  return (size_t)tasks * gram_blocks_per_task(n, h) * gram_ng(ci) * gram_ng(ci);
}
size_t gram_doubles(int tasks, int ci) { return (size_t)tasks * gram_ng(ci) * gram_ng(ci); }

// x [T][n][H][W][ci] -> g [T][NG][NG] (row/column 9*ci = the constant-one entry: G[a][9ci] = s[a], G[9ci][9ci] = pixel count)
hipError_t launch_input_gram(hipStream_t st, const float* x, int tasks, int n, int h, int w, int ci, double* partial, double* g) {
  const int nblk = gram_blocks_per_task(n, h), ng = gram_ng(ci);
  if (ci == 3)
    hipLaunchKernelGGL(input_gram_kernel<3>, dim3(nblk, tasks), dim3(256), 0, st, x, n, h, w, kGramRowsPerWave, partial);
  else if (ci == 1)
    hipLaunchKernelGGL(input_gram_kernel<1>, dim3(nblk, tasks), dim3(256), 0, st, x, n, h, w, kGramRowsPerWave, partial);
  else
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(ceil_div(ng * ng, 256), tasks), dim3(256), 0, st, partial, nblk, ng, g);
  return hipGetLastError();
}

hipError_t launch_gram_stats(hipStream_t st, const double* g, int tasks, int ci, int co, const float* w, size_t wstride,
                             const float* wd, size_t vstride, double inv_m, int tangent, float* out0, float* out1,
                             const float* mu, const float* rstd) {
  const int ng = gram_ng(ci);
  const int kp = 9 * ci;
  const size_t smem = ((size_t)ng * ng + (size_t)(tangent ? 3 : 2) * kp * co) * sizeof(double);   // <= 49 KB (ci 3, co 64, tangent)
  hipLaunchKernelGGL(gram_stats_kernel, dim3(tasks), dim3(256), smem, st, g, ng, kp, w, wstride, wd, vstride, co, inv_m, tangent,
                     out0, out1, mu, rstd);
  return hipGetLastError();
}

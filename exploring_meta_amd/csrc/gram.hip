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
#include "fold.h"
#include "bf16_split.h"

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
                                                         int row_pitch, double* __restrict__ partial) {
  using D = GramDims<CI0>;
  constexpr int NT = D::NT, KP = D::KP;
  // LDS: per wave the three input rows an output row touches, as fp64, [3][RPD] with a zero halo pixel on the left and zero
  // padding on the right (so steps that run past the row read zeros); the cross-wave reduction re-uses the buffer from 0.
  extern __shared__ __attribute__((aligned(16))) double ldsd[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kpix = lane >> 4;
  const int task = blockIdx.y;
  const int RPD = row_pitch, ROWF = W * CI0;
  double* rows = ldsd + (size_t)wave * 3 * RPD;
  const float* x_t = x + (size_t)task * n * H * W * CI0;
  // this lane's patch entries: LDS offset of (row dy+1, column dx+1 pixels incl. the halo, channel c); the constant-one entry
  // and the padding entries read the left halo (zero) and are patched / left at zero
  int off[NT];
  bool one[NT], pad[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int a = 16 * t + i;
    one[t] = a == KP;
    pad[t] = a > KP;
    const int tap = a < KP ? a / CI0 : 0, c = a < KP ? a % CI0 : 0;
    off[t] = a < KP ? (tap / 3) * RPD + (tap % 3) * CI0 + c + kpix * CI0 : 0;
  }
  doublex4 acc[NT][NT];
#pragma unroll
  for (int ti = 0; ti < NT; ++ti)
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) acc[ti][tj] = doublex4{0.0, 0.0, 0.0, 0.0};
  for (int e = lane; e < 3 * RPD; e += 64) rows[e] = 0.0;     // halos and padding stay zero

  const int nrows = n * H;
  const int row0 = (blockIdx.x * 4 + wave) * rows_per_wave;
  const int row1 = min(row0 + rows_per_wave, nrows);
  // Rows of at most 256 floats (84 x 3, 28 x 1): the three input rows of an output row are REQUESTED as 12 unconditional loads (clamped
  // addresses, zero selected afterwards) one output row ahead, under the matrix work of the current row.  As a loop of predicated
  // load / convert / store triples the staging was twelve memory round trips per output row against one microsecond of MFMAs
  // (the kernel ran at a quarter of the fp64 matrix rate).
  const bool narrow = ROWF <= 256;
  float pre[3][4];
  auto fetch_rows = [&](int row) {
    const int img = row / H, y = row - img * H;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int yy = y - 1 + r;
      const bool rv = (unsigned)yy < (unsigned)H;
      const float* src = x_t + (size_t)(img * H + (rv ? yy : 0)) * ROWF;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = lane + 64 * k;
        const float f = src[e < ROWF ? e : ROWF - 1];
        pre[r][k] = rv ? f : 0.f;
      }
    }
  };
  if (narrow && row0 < row1) fetch_rows(row0);
  for (int row = row0; row < row1; ++row) {
    const int img = row / H, y = row - img * H;
    // stage input rows y-1, y, y+1 (coalesced), converted to fp64 once; rows outside the image are zeros
    if (narrow) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int e = lane + 64 * k;
          if (e < ROWF) rows[r * RPD + CI0 + e] = (double)pre[r][k];
        }
      if (row + 1 < row1) fetch_rows(row + 1);              // in flight under this row's products
    } else {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int yy = y - 1 + r;
      const bool rv = (unsigned)yy < (unsigned)H;
      const float* src = x_t + (size_t)(img * H + (rv ? yy : 0)) * ROWF;
      for (int e = lane; e < ROWF; e += 64) rows[r * RPD + CI0 + e] = rv ? (double)src[e] : 0.0;
    }
    }
    for (int x0 = 0; x0 < W; x0 += 4) {
      const bool pv = x0 + kpix < W;
      double v[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const double d = rows[off[t] + x0 * CI0];              // LDS operations of one wave execute in order: no barrier
        v[t] = (pv && !pad[t]) ? (one[t] ? 1.0 : d) : 0.0;      // pixels past the row end / padding entries contribute nothing
      }
      // G is symmetric: tiles below the diagonal are filled in by gram_reduce_kernel
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = ti; tj < NT; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[ti], v[tj], acc[ti][tj], 0, 0, 0);
    }
  }
  // 4 waves -> one partial per workgroup, fixed order
  __syncthreads();
  double* red = ldsd;
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
// The statistics proper, on tables already in LDS: gs [ng][ng], ws / vs [kp][co] (vs == ws when !tangent), qd [kp][co] scratch.
// Called by every thread of the workgroup (NT threads); the caller has synchronised after filling the tables.
template <int NT>
__device__ __forceinline__ void gram_stats_body(const double* gs, const double* ws, const double* vs, double* qd, int ng, int kp, int co,
                                                double inv_m, int tangent, float* __restrict__ out0, float* __restrict__ out1,
                                                const float* __restrict__ mu_in, const float* __restrict__ rstd_in, int task) {
  const int tid = threadIdx.x;
  for (int e = tid; e < kp * co; e += NT) {
    const int a = e / co, c = e - a * co;
    double rowdot = 0.0;
    for (int b = 0; b < kp; ++b) rowdot = fma(ws[b * co + c], gs[b * ng + a], rowdot);
    qd[e] = vs[e] * rowdot;
  }
  __syncthreads();
  for (int c = tid; c < co; c += NT) {
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
  gram_stats_body<256>(gs, ws, vs, qd, ng, kp, co, inv_m, tangent, out0, out1, mu_in, rstd_in, task);
}

// ---------------------------------------------------------------------------------------------------------------------
// Sparse part of the block-1 weight gradient.  The cotangent du of BN+ReLU+max-pool reaches the conv output at ONE position per
// pooling window and channel (the argmax, if it passed the ReLU); block1_kernel<FWD> left that position in `arg`.
//   S[k][co] = sum over windows of patch_k(pixel of window at position arg[co]) * cot[window][co]
// as four masked MFMA accumulations (one per position q): A operand = patch entries of the window's q-th pixel (lane = entry k,
// two windows per K step), B operand = cot where arg == q else 0.  cot = dp (primal) or c1 dp + gr dpd (tangent, the sparse
// part of R{dz}).  No BatchNorm arithmetic, no conv: 64 MFMAs per 32 windows against 124 in block1_kernel<*_WGRAD>.
// All operands come through raw buffer loads whose descriptors are rebuilt per pooled row / input row on the scalar unit (base =
// start of the row, num_records = the row's bytes, 0 for a row outside the image): windows past the end of a row, rows above /
// below the image and the lanes past the end of an input row read 0 from the range check, the K steps of a chunk are load
// IMMEDIATES, and no address arithmetic, predicate or select is left on the vector unit -- fp32 MFMAs and VALU instructions share
// the SIMD's lanes, so what remains per K step is the four argmax compares + selects (and the c1 / gr combination in tangent mode).
struct SparseB {   // B-side operands of one chunk of CH K steps, as loaded (combined at use: nothing waits at the prefetch)
  float q[8], qd[8];
  unsigned ag[8];
};
template <int CI0, bool TAN, int CO, int CH>
__global__ __launch_bounds__(256) void sparse_wgrad_kernel(SparseWgArgs a) {
  constexpr int K = 9 * CI0;
  constexpr unsigned RSRC = 0x00020000u;
  // LDS: per wave two sets (ping-pong over tiles) of the 4 input rows a pooled row touches, [2][4][RP] with a zero halo pixel left
  // and right; the pitch is padded to 12 (mod 32) floats so the three tap rows land on disjoint banks, and is at least CI0 + 256 so
  // that the 4 x 64 lanes of a row store never leave the row.  The cross-wave reduction at the end re-uses the buffer.
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int task = blockIdx.y, cbase = blockIdx.z * 32, ch = cbase + j;
  const int H = a.hh, W = a.ww, HP = H >> 1, WP = W >> 1;
  const int RP = a.row_pitch, ROWF = W * CI0;
  float* rows = lds + wave * 8 * RP;
  const float* x_t = a.x + (size_t)task * a.n * H * W * CI0;
  const size_t p_task = (size_t)a.n * HP * WP * CO;
  const uint8_t* arg_t = a.arg + (size_t)task * p_task;
  const float* dp_t = a.dp + (size_t)task * p_task;
  const float* dpd_t = TAN ? a.dpd + (size_t)task * p_task : dp_t;
  // A-operand role of this lane: patch entry k = j (tap, channel) of the window pair's window h; rows k >= K of the product are
  // never stored.  Offset of pixel (qy = 0, qx = 0) of window h of pair 0 for this lane's tap, halo pixel included.
  const bool kval = j < K;
  const int tap = kval ? j / CI0 : 0, kc = kval ? j % CI0 : 0;
  const int kdy = tap / 3 - 1, kdx = tap % 3 - 1;
  const float* arow = rows + (kval ? (kdy + 1) * RP + (kdx + 1) * CI0 + kc : 0) + 2 * h * CI0;
  // B-operand role: output channel ch of window h of the pair
  const unsigned lane_b = (unsigned)(h * CO + ch), lane_x = (unsigned)lane * 4u;
  float sA = 1.f, sB = 0.f;
  if (TAN) {
    const float rs = a.rstd[(size_t)task * CO + ch], gm = a.gamma[(size_t)task * a.pstride + ch];
    sA = a.gammad[(size_t)task * a.vstride + ch] * rs + gm * (-rs * rs * a.m2[(size_t)task * CO + ch]);   // c1
    sB = gm * rs;                                                                                             // gr
  }
  floatx16 acc, acc2;          // two accumulation chains: consecutive MFMAs never wait for each other's result
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
  for (int e = lane; e < 8 * RP; e += 64) rows[e] = 0.f;      // halos (and padding) stay zero for the whole kernel

  // one tile = one pooled row (img, wy): WP windows, ceil(WP / 2) K steps of two windows, in chunks of CH steps
  const int nsteps = (WP + 1) >> 1;
  // this workgroup's share of the task's tiles: an even split (shares differ by at most one tile)
  const int tile_base = (int)((long)blockIdx.x * a.ntiles / gridDim.x);
  const int tile_end = (int)((long)(blockIdx.x + 1) * a.ntiles / gridDim.x);
  float rbuf[4][4];
  auto fetch_rows = [&](int tile) {           // the 4 input rows of a tile -> registers (zeros outside the image / past the row)
    const int img = tile / HP, wy = tile - img * HP;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = 2 * wy - 1 + r;
      const bool rv = (unsigned)y < (unsigned)H;
      const mi_rsrc rr = __builtin_amdgcn_make_buffer_rsrc((void*)(x_t + (size_t)(img * H + (rv ? y : 0)) * ROWF), 0, rv ? ROWF * 4 : 0, RSRC);
#pragma unroll
      for (int i = 0; i < 4; ++i) rbuf[r][i] = buf_ld(rr, lane_x + 256u * i);
    }
  };
  auto store_rows = [&](int set) {            // registers -> LDS set (row r <-> input row 2wy - 1 + r)
    float* dst = rows + set * 4 * RP + CI0 + lane;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[r * RP + 64 * i] = rbuf[r][i];
  };
  auto fetch_b = [&](int tile, int s0, SparseB& b) {
    const size_t prow = (size_t)tile * WP * CO;
    const mi_rsrc ra = __builtin_amdgcn_make_buffer_rsrc((void*)(arg_t + prow), 0, WP * CO, RSRC);
    const mi_rsrc rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dp_t + prow), 0, WP * CO * 4, RSRC);
    const mi_rsrc rdd = __builtin_amdgcn_make_buffer_rsrc((void*)(dpd_t + prow), 0, WP * CO * 4, RSRC);
    unsigned vb = lane_b + (unsigned)(s0 * 2 * CO), vb4 = vb * 4u;
    asm volatile("" : "+v"(vb), "+v"(vb4));        // opaque bases: the K steps below become load immediates, not re-associated adds
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      b.ag[i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(ra, vb + (unsigned)(i * 2 * CO), 0, 0);
      b.q[i] = buf_ld(rd, vb4 + (unsigned)(i * 8 * CO));
      if (TAN) b.qd[i] = buf_ld(rdd, vb4 + (unsigned)(i * 8 * CO));
    }
  };
  // flat stream of (tile, chunk) work items: the B operands are fetched one item ahead (also across tile boundaries), the input
  // rows one tile ahead into the other LDS set and two tiles ahead into registers
  int tile = tile_base + ((wave + blockIdx.x) & 3), s0 = 0, set = 0;      // (rotated: a share's odd tile lands on a different SIMD per workgroup)
  SparseB bA, bB;
  if (tile < tile_end) {
    fetch_rows(tile);
    fetch_b(tile, 0, bA);
    store_rows(0);
    if (tile + 4 < tile_end) fetch_rows(tile + 4);
  }
  auto item = [&](const SparseB& cur, SparseB& nxt) {
    if (s0 == 0 && tile + 4 < tile_end) {
      store_rows(set ^ 1);
      if (tile + 8 < tile_end) fetch_rows(tile + 8);
    }
    int ntile = tile, ns0 = s0 + CH, nset = set;
    if (ns0 >= nsteps) { ns0 = 0; ntile += 4; nset ^= 1; }
    if (ntile < tile_end) fetch_b(ntile, ns0, nxt);
    const float* ac = arow + set * 4 * RP + 4 * s0 * CI0;
    const float* ac1 = ac + RP;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if (CH == 7 || s0 + i < nsteps) {            // chunks of 7 are launched only when they tile the row exactly
        const float a0 = ac[4 * i * CI0], a1 = ac[4 * i * CI0 + CI0], a2 = ac1[4 * i * CI0], a3 = ac1[4 * i * CI0 + CI0];
        const unsigned hot = 1u << cur.ag[i];       // one-hot of the argmax byte: bit q -> lane mask of position q (mi_common.h)
        const float cot = TAN ? fmaf(sB, cur.qd[i], sA * cur.q[i]) : cur.q[i];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, lane_keep_where(lane_mask_bit<0>(hot), cot), acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, lane_keep_where(lane_mask_bit<1>(hot), cot), acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, lane_keep_where(lane_mask_bit<2>(hot), cot), acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, lane_keep_where(lane_mask_bit<3>(hot), cot), acc2, 0, 0, 0);
      }
    }
    tile = ntile; s0 = ns0; set = nset;
  };
  while (tile < tile_end) {
    item(bA, bB);
    if (tile >= tile_end) break;
    item(bB, bA);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
  // 4 waves -> one partial per workgroup (same layout as block1_kernel<*_WGRAD>'s partials)
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) lds[wave * 1024 + r * 64 + lane] = acc[r];
  __syncthreads();
  float* pt = a.wpartial + ((size_t)task * gridDim.x + blockIdx.x) * K * CO;
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const int e = tid + 256 * qq;
    const float v = lds[e] + lds[1024 + e] + lds[2048 + e] + lds[3072 + e];
    const int r = e >> 6, l = e & 63;
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    if (row < K) pt[(size_t)row * CO + cbase + col] = v;
  }
}

// The same product for rows of exactly NCH chunks of CH K steps (84-wide RGB: 7 x 3), as a loop over PAIRS of tiles
// with every load issued unconditionally (descriptors of tiles past the wave's range have 0 records: the loads return 0 without
// touching memory).  The number of loads in flight at every point is then the same on every path, so the compiler's s_waitcnt
// counts are exact -- in the work-item loop above the conditional row fetches make it assume the worst and wait for the chunk it
// has just issued, which exposes a full memory latency per chunk.
//
// BF: the same product on the 16-bit matrix pipe at fp32-equivalent accuracy, with the SIX partial products of the split-bf16 operand form
// (bf16_split.h: a b = ah bh + am bh + ah bl + al bh + ah bm + am bm up to 2^-24 |a b|) laid out along K.  A window of the pair owns the
// eight K slots of its lane half of v_mfma_f32_32x32x16_bf16.  THREE instructions per K step cover the four positions of the window:
//   A = (ah, am, ah, al) of position 0 | of position 1,  B = (bh, bh, bl, bh) & mask 0 | & mask 1     [ah bh + am bh + ah bl + al bh]
//   the same for positions 2 | 3 (the window's lower row),
//   A = (ah, am) of positions 0 | 1 | 2 | 3,             B = (bm, bm) & mask 0 | 1 | 2 | 3            [ah bm + am bm]
// -- where the fp32 form needs four 16-pass MFMAs these are three 8-pass ones, and, unlike the fp32 instruction, they leave the SIMD's
// vector lanes to other work.  The A side costs nothing in the loop: the input rows are split ONCE, when they are staged into LDS, and
// stored as the two dwords of a position's slots (P = ah | am << 16, Q = ah | al << 16; two neighbouring values = one ds_read2_b64 straight
// into the operand registers, the four P alone = two ds_read2_b32); the B side is one split per window and channel (shared by the four
// positions) and three ANDs per position.  What bounds it is the vector port (profiles/r6/mfma_dep_probe.txt: an 8-pass MFMA hides about
// five vector instructions; this loop has ten per MFMA).
template <int CI0, bool TAN, int CO, int CH, int NCH, bool BF = false>
__global__ __launch_bounds__(256, BF ? 2 : 4) void sparse_wgrad_rows_kernel(SparseWgArgs a) {
  constexpr int K = 9 * CI0;
  constexpr unsigned RSRC = 0x00020000u;
  constexpr int EW = BF ? 2 : 1;                 // dwords per staged input value
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int task = blockIdx.y, cbase = blockIdx.z * 32, ch = cbase + j;
  const int H = a.hh, W = a.ww, HP = H >> 1, WP = W >> 1;
  const int RP = a.row_pitch, ROWF = W * CI0;
  float* rows = lds + wave * 8 * RP * EW;
  const float* x_t = a.x + (size_t)task * a.n * H * W * CI0;
  const size_t p_task = (size_t)a.n * HP * WP * CO;
  const uint8_t* arg_t = a.arg + (size_t)task * p_task;
  const float* dp_t = a.dp + (size_t)task * p_task;
  const float* dpd_t = TAN ? a.dpd + (size_t)task * p_task : dp_t;
  const bool kval = j < K;
  const int tap = kval ? j / CI0 : 0, kc = kval ? j % CI0 : 0;
  const int kdy = tap / 3 - 1, kdx = tap % 3 - 1;
  const float* arow = rows + ((kval ? (kdy + 1) * RP + (kdx + 1) * CI0 + kc : 0) + 2 * h * CI0) * EW;
  const float* arow1 = arow + RP * EW;
  int same = 0;                                               // (BF) a zero the compiler cannot see through: the same rows through registers of their own
  if constexpr (BF) asm volatile("" : "+v"(same));
  const float *arow_z = arow + same, *arow1_z = arow1 + same;
  const unsigned lane_b = (unsigned)(h * CO + ch), lane_b4 = lane_b * 4u, lane_x = (unsigned)lane * 4u;
  float sA = 1.f, sB = 0.f;
  if (TAN) {
    const float rs = a.rstd[(size_t)task * CO + ch], gm = a.gamma[(size_t)task * a.pstride + ch];
    sA = a.gammad[(size_t)task * a.vstride + ch] * rs + gm * (-rs * rs * a.m2[(size_t)task * CO + ch]);   // c1
    sB = gm * rs;                                                                                             // gr
  }
  floatx16 acc, acc2, zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; zero16[r] = 0.f; }
  // (BF) The sum over the tiles of a wave, of a workgroup's waves and -- in the consumers' folds -- over the workgroups runs in fp64 from each
  // tile's fp32 result on: what reaches the fold does not depend on how the launch cut the task into shares (to ~1e-16), so a task's
  // gradient is the same whether it runs alone or beside 31 others -- the fp32 chain over a share's ~1400 MFMAs was this kernel's part of
  // the engine's sensitivity to launch geometry.
  double dacc[BF ? 16 : 1];
  if constexpr (BF) {
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[r] = 0.0;
  }
  for (int e = lane; e < 8 * RP * EW; e += 64) rows[e] = 0.f;
  // this workgroup's share of the task's tiles: an even split (shares differ by at most one tile)
  const int tile_base = (int)((long)blockIdx.x * a.ntiles / gridDim.x);
  const int tile_end = (int)((long)(blockIdx.x + 1) * a.ntiles / gridDim.x);
  float rbuf[4][4];
  auto fetch_rows = [&](int tile) {
    const bool tv = tile < tile_end;
    const int tl = tv ? tile : 0;
    const int img = tl / HP, wy = tl - img * HP;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = 2 * wy - 1 + r;
      const bool rv = tv && (unsigned)y < (unsigned)H;
      const mi_rsrc rr = __builtin_amdgcn_make_buffer_rsrc((void*)(x_t + (size_t)(img * H + (rv ? y : 0)) * ROWF), 0, rv ? ROWF * 4 : 0, RSRC);
#pragma unroll
      for (int i = 0; i < 4; ++i) rbuf[r][i] = buf_ld(rr, lane_x + 256u * i);
    }
  };
  auto store_rows = [&](int set) {
    if constexpr (BF) {
      // every staged value split here, once: (P, Q) = (ah | am << 16, ah | al << 16), one ds_write_b64 per value
      u32x2* dst = reinterpret_cast<u32x2*>(rows) + set * 4 * RP + CI0 + lane;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
          unsigned sh, sm, sl;
          bf16_split2(floatx2{rbuf[r][i], rbuf[r][i + 1]}, sh, sm, sl);
          dst[r * RP + 64 * i] = u32x2{__builtin_amdgcn_perm(sm, sh, 0x05040100u), __builtin_amdgcn_perm(sl, sh, 0x05040100u)};
          dst[r * RP + 64 * (i + 1)] = u32x2{__builtin_amdgcn_perm(sm, sh, 0x07060302u), __builtin_amdgcn_perm(sl, sh, 0x07060302u)};
        }
    } else {
      float* dst = rows + set * 4 * RP + CI0 + lane;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[r * RP + 64 * i] = rbuf[r][i];
    }
  };
  auto fetch_b = [&](int tile, int c, SparseB& b) {
    const bool tv = tile < tile_end;
    const size_t prow = (size_t)(tv ? tile : 0) * WP * CO;
    const int rec = tv ? WP * CO : 0;
    const mi_rsrc ra = __builtin_amdgcn_make_buffer_rsrc((void*)(arg_t + prow), 0, rec, RSRC);
    const mi_rsrc rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dp_t + prow), 0, rec * 4, RSRC);
    const mi_rsrc rdd = __builtin_amdgcn_make_buffer_rsrc((void*)(dpd_t + prow), 0, rec * 4, RSRC);
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      b.ag[i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(ra, lane_b + (unsigned)((c * CH + i) * 2 * CO), 0, 0);
      b.q[i] = buf_ld(rd, lane_b4 + (unsigned)((c * CH + i) * 8 * CO));
      if (TAN) b.qd[i] = buf_ld(rdd, lane_b4 + (unsigned)((c * CH + i) * 8 * CO));
    }
  };
  // (BF) the A operands of one chunk, read from LDS one chunk ahead of their MFMAs: [K step][top / bottom row of the window]
  struct SparseA { mi_u32x4 v[CH][3]; };       // per K step: (P, Q) of positions 0 | 1, of positions 2 | 3, and the four P alone
  auto fetch_a = [&](int set, int c, SparseA& A) {
    const u32x2* ac = reinterpret_cast<const u32x2*>(arow) + set * 4 * RP;
    const u32x2* ac1 = reinterpret_cast<const u32x2*>(arow1) + set * 4 * RP;
    const unsigned* az = reinterpret_cast<const unsigned*>(arow_z) + set * 8 * RP;       // (a second read of the P dwords: LDS cycles instead of
    const unsigned* az1 = reinterpret_cast<const unsigned*>(arow1_z) + set * 8 * RP;     //  four register moves on the busier vector port)
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int o = 4 * (c * CH + i) * CI0;
      const u32x2 a0 = ac[o], a1 = ac[o + CI0], a2 = ac1[o], a3 = ac1[o + CI0];
      A.v[i][0] = mi_u32x4{a0[0], a0[1], a1[0], a1[1]};
      A.v[i][1] = mi_u32x4{a2[0], a2[1], a3[0], a3[1]};
      A.v[i][2] = mi_u32x4{az[2 * o], az[2 * (o + CI0)], az1[2 * o], az1[2 * (o + CI0)]};
    }
  };
  auto compute = [&](int set, int c, const SparseB& b, const SparseA& A) {
    if constexpr (BF) {
      // (operand layout: the kernel's header)
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        {
          // the cotangent's three pieces, each as the packed pair the K slots want, straight out of the conversions: cvt_pk(c, c) = bh|bh,
          // cvt_pk(r, r) = bm|bm with r = c - bh, cvt_pk(r - bm, c) = bl|bh (the subtractions are exact)
          const float cv = TAN ? fmaf(sB, b.qd[i], sA * b.q[i]) : b.q[i];
          const unsigned bhh = __builtin_bit_cast(unsigned, __builtin_convertvector(floatx2{cv, cv}, bf16x2));
          const float r1 = bf16_sub(cv, __uint_as_float(bhh & 0xffff0000u));
          const unsigned bmm = __builtin_bit_cast(unsigned, __builtin_convertvector(floatx2{r1, r1}, bf16x2));
          const float r2 = bf16_sub(r1, __uint_as_float(bmm & 0xffff0000u));
          const unsigned blh = __builtin_bit_cast(unsigned, __builtin_convertvector(floatx2{r2, cv}, bf16x2));
          const unsigned hot = 1u << b.ag[i];
          const unsigned m0 = (unsigned)lane_mask_bit<0>(hot), m1 = (unsigned)lane_mask_bit<1>(hot), m2 = (unsigned)lane_mask_bit<2>(hot), m3 = (unsigned)lane_mask_bit<3>(hot);
          const bf16x8 top = __builtin_bit_cast(bf16x8, A.v[i][0]), bot = __builtin_bit_cast(bf16x8, A.v[i][1]), pz = __builtin_bit_cast(bf16x8, A.v[i][2]);
          // (a tile starts from C = 0 -- an inline constant --: its fp32 sum is the same whichever wave of whichever launch geometry computes it)
          const bool first = c == 0 && i == 0;
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(top, __builtin_bit_cast(bf16x8, (mi_u32x4{bhh & m0, blh & m0, bhh & m1, blh & m1})), first ? zero16 : acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bot, __builtin_bit_cast(bf16x8, (mi_u32x4{bhh & m2, blh & m2, bhh & m3, blh & m3})), first ? zero16 : acc2, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pz, __builtin_bit_cast(bf16x8, (mi_u32x4{bmm & m0, bmm & m1, bmm & m2, bmm & m3})), acc, 0, 0, 0);
        }
      }
      return;
    }
    const float* ac = arow + set * 4 * RP, * ac1 = arow1 + set * 4 * RP;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int o = 4 * (c * CH + i) * CI0;
      const float a0 = ac[o], a1 = ac[o + CI0], a2 = ac1[o], a3 = ac1[o + CI0];
      // B of position q is the cotangent where the stored argmax byte is q: one-hot of the byte, its bit q smeared into a lane
      // mask, one AND (mi_common.h: a compare + v_cndmask pair per position costs more than twice as much issue time)
      const unsigned hot = 1u << b.ag[i];
      const float cot = TAN ? fmaf(sB, b.qd[i], sA * b.q[i]) : b.q[i];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, lane_keep_where(lane_mask_bit<0>(hot), cot), acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, lane_keep_where(lane_mask_bit<1>(hot), cot), acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, lane_keep_where(lane_mask_bit<2>(hot), cot), acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, lane_keep_where(lane_mask_bit<3>(hot), cot), acc2, 0, 0, 0);
    }
  };
  int tile = tile_base + ((wave + blockIdx.x) & 3);   // this wave's tiles: tile, tile + 4, ... (start rotated per workgroup); pairs per iteration
  if constexpr (BF) {
    // The bf16 MFMAs of a chunk take 12 x 32 cycles where the fp32 ones took 12 x 64, and two waves share a SIMD instead of four: one chunk
    // of run-ahead no longer covers a global load.  The B operands run BD chunks ahead (buffers indexed by the chunk's place in its pooled
    // row: the same register set for the same place in every tile).
    constexpr int BD = TAN ? 2 : 3;      // (tangent mode loads one more operand per K step: one chunk less of run-ahead keeps it inside the register file)
    SparseB bb[NCH];
    fetch_rows(tile);
    store_rows(0);
    fetch_rows(tile + 4);
#pragma unroll
    for (int c = 0; c < BD; ++c) fetch_b(tile, c, bb[c]);
    __builtin_amdgcn_sched_barrier(0);
    while (tile < tile_end) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {       // half 0: tile (LDS set 0), half 1: tile + 4 (set 1)
        const int cur_tile = tile + 4 * half, nxt_tile = cur_tile + 4;
        store_rows(half ^ 1);                      // rows of the next tile (in registers since the previous half) -> the other set
        fetch_rows(cur_tile + 8);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          if (c + BD < NCH) fetch_b(cur_tile, c + BD, bb[c + BD]);
          else fetch_b(nxt_tile, c + BD - NCH, bb[c + BD - NCH]);
          SparseA aa;                                        // (the chunk's A operands: all nine LDS reads up front; reading them a chunk ahead
          fetch_a(half, c, aa);                              //  measured the same and costs 36 registers this kernel does not have)
          __builtin_amdgcn_sched_barrier(0);
          if (cur_tile < tile_end) compute(half, c, bb[c], aa);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (cur_tile < tile_end) {
#pragma unroll
          for (int r = 0; r < 16; ++r) dacc[r] += (double)(acc[r] + acc2[r]);
        }
      }
      tile += 8;
    }
  } else {
  SparseB bb[2];
  SparseA aa[1];
  fetch_rows(tile);
  store_rows(0);
  fetch_rows(tile + 4);
  fetch_b(tile, 0, bb[0]);                       // (after the rows, as at the end of every loop half: same loads in flight on both paths into the loop)
  __builtin_amdgcn_sched_barrier(0);
  while (tile < tile_end) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {       // half 0: tile (LDS set 0), half 1: tile + 4 (set 1)
      const int cur_tile = tile + 4 * half, nxt_tile = cur_tile + 4;
      store_rows(half ^ 1);                      // rows of the next tile (in registers since the previous half) -> the other set
      fetch_rows(cur_tile + 8);
      __builtin_amdgcn_sched_barrier(0);         // the prefetches stay where they are written: ahead of the MFMAs that hide them
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int m = half * NCH + c;            // chunk index within the iteration: B buffers alternate
        if (c + 1 < NCH) fetch_b(cur_tile, c + 1, bb[(m + 1) & 1]);
        else fetch_b(nxt_tile, 0, bb[(m + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (cur_tile < tile_end) compute(half, c, bb[m & 1], aa[0]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    tile += 8;
  }
  }
  // (lane index and column base re-derived here rather than kept live across the loop: the kernel sits at its 128-register budget)
  const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  __syncthreads();
  if constexpr (BF) {
    double* ldsd = reinterpret_cast<double*>(lds);
#pragma unroll
    for (int r = 0; r < 16; ++r) ldsd[wave * 1024 + r * 64 + ln] = dacc[r];
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) lds[wave * 1024 + r * 64 + ln] = acc[r] + acc2[r];
  }
  __syncthreads();
  float* pt = a.wpartial + ((size_t)blockIdx.y * gridDim.x * (BF ? 2 : 1) + blockIdx.x) * K * CO + blockIdx.z * 32;
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const int e = wave * 64 + ln + 256 * qq;
    float v, vlo = 0.f;
    if constexpr (BF) {
      // the workgroup's fp64 sum leaves as TWO fp32 partials, v + vlo (vlo in the second half of the task's partial blocks: the consumers fold
      // 2 * gridDim.x partials in fp64, which restores the sum to 2^-48)
      const double* ldsd = reinterpret_cast<const double*>(lds);
      const double d = ((ldsd[e] + ldsd[1024 + e]) + ldsd[2048 + e]) + ldsd[3072 + e];
      v = (float)d;
      vlo = (float)(d - (double)v);
    } else {
      v = lds[e] + lds[1024 + e] + lds[2048 + e] + lds[3072 + e];
    }
    const int r = e >> 6;
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), col = ln & 31;
    if (row < K) {
      pt[(size_t)row * CO + col] = v;
      if constexpr (BF) pt[(size_t)gridDim.x * K * CO + (size_t)row * CO + col] = vlo;
    }
  }
}

// Dense parts and assembly, one workgroup per task (fp64).  With s = patch sums, GW = G w, Z = sum patch x zhat = r (GW - mu s):
//   primal :  dW    = gr (S - dbm s - dgm Z)
//   tangent:  R{dW} = S - (c1 dbm + gr rbm) s - (c1 dgm + gr rgm) Z - gr dgm Zd,   Zd = sum patch x zhatd = r (G wd - m1 s - m2 Z)
// (dbm = dbeta/M, dgm = dgamma/M, rbm / rgm their tangents; S already carries c1 / gr in tangent mode.)
// S = the element's folded sparse part (canonical fold of fold.h over the nblk workgroup partials, fp64)
__device__ __forceinline__ const float* gram_wgrad_partials(const GramWgArgs& a, int kp, int task, int k, int c) {
  return a.spartial + (size_t)task * a.nblk * kp * a.co + k * a.co + c;
}
__device__ __forceinline__ float gram_wgrad_elem(const GramWgArgs& a, int ng, int kp, int tangent, int task, int k, int c, double S) {
  const int co = a.co;
  const double* grow = a.g + (size_t)task * ng * ng + (size_t)k * ng;      // G[k][.]; G[k][kp] = s[k]
  const float* wc = a.w + (size_t)task * a.wstride + c;
  const float* vc = tangent ? a.wd + (size_t)task * a.vstride + c : wc;
  double gw = 0.0, gwd = 0.0;
#pragma unroll 9
  for (int b2 = 0; b2 < kp; ++b2) {
    gw = fma(grow[b2], (double)wc[(size_t)b2 * co], gw);
    if (tangent) gwd = fma(grow[b2], (double)vc[(size_t)b2 * co], gwd);
  }
  const double sk = grow[kp];
  const double mu = (double)a.mu[(size_t)task * co + c], rs = (double)a.rstd[(size_t)task * co + c];
  const double gm = (double)a.gamma[(size_t)task * a.pstride + c];
  const double gr = gm * rs;
  const double dgm = (double)a.dgamma[(size_t)task * a.gstride + c] * a.inv_m, dbm = (double)a.dbeta[(size_t)task * a.gstride + c] * a.inv_m;
  const double Z = rs * (gw - mu * sk);
  double out;
  if (!tangent) {
    out = gr * (S - dbm * sk - dgm * Z);
  } else {
    const double m1 = (double)a.m1[(size_t)task * co + c], m2 = (double)a.m2[(size_t)task * co + c];
    const double gmd = (double)a.gammad[(size_t)task * a.vstride + c];
    const double c1 = gmd * rs + gm * (-rs * rs * m2);
    const double rgm = (double)a.rdgamma[(size_t)task * a.hstride + c] * a.inv_m, rbm = (double)a.rdbeta[(size_t)task * a.hstride + c] * a.inv_m;
    const double Zd = rs * (gwd - m1 * sk - m2 * Z);
    out = S - (c1 * dbm + gr * rbm) * sk - (c1 * dgm + gr * rgm) * Z - gr * dgm * Zd;
  }
  return (float)out;
}

__global__ __launch_bounds__(64) void gram_wgrad_kernel(GramWgArgs a, int ng, int kp, int tangent) {
  // one small workgroup per (task, patch entry k): thread = output channel
  const int task = blockIdx.x, k = blockIdx.y, co = a.co;
  for (int c = threadIdx.x; c < co; c += 64) {
    const double S = fold_all<double>(gram_wgrad_partials(a, kp, task, k, c), (size_t)kp * co, a.nblk);
    a.out[(size_t)task * a.ostride + k * co + c] = gram_wgrad_elem(a, ng, kp, tangent, task, k, c, S);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The tail of a pass as one launch (kernels.h, AdvanceArgs).  1024 threads per workgroup, grid (chunk workgroups + row workgroups, tasks).
//   chunk workgroups: thread (slice, element) -- S = the launch's largest fold_slices() slices of 1024 / S elements.  Every element of the
//     task's gradient-shaped vector g is finished: folded from weight-gradient partials (the canonical order of fold.h: the slices of an
//     element meet in LDS), zeroed, or simply read; then out = a - alpha g.
//   row workgroups (a.b1_wgrad): the Gram-matrix assembly of block 1's dW1[k][.] (gram_wgrad_elem), thread (row k, slice, channel), the
//     sparse part folded in fp64 -- a workgroup takes 1024 / (co * slices) rows.  The chunk workgroups skip those elements.
//   Gram statistics of the next pass (a.stats): they need ALL of block 1's finished weights, which several workgroups produce -- those
//     workgroups store them write-through, count their arrival (finalize.h protocol: acknowledged stores, barrier, one relaxed
//     agent-scope add per workgroup) and the one that arrives last reads them back with agent-scope loads and runs gram_stats_body.
// Because the row workgroups read the direction's block-1 entries (GramWgArgs::wd, gammad) that other workgroups update, the update must
// not be in place when a.b1_wgrad is set: the engine ping-pongs lam between two buffers.
struct AdvanceElem { int kind; float v; };   // kind 0: v is the value (g untouched), 1: v is the value and g must be written, 2: another workgroup owns it
__device__ __forceinline__ AdvanceElem advance_classify(const AdvanceArgs& a, int task, unsigned e, int kp, int S, int sl, float* part, int epw, int el) {
  const float* g_t = a.g + (size_t)task * a.gstride;
  for (int z = 0; z < a.nzero; ++z)
    if (e - a.zoff[z] < a.zlen[z]) return AdvanceElem{1, 0.f};
  for (int sgi = 0; sgi < a.nseg; ++sgi) {
    const AdvanceSeg& sg = a.seg[sgi];
    const unsigned r = e - sg.off;
    if (r < sg.nelem) {
      const int Ss = fold_slices(sg.nchunks);
      float v = 0.f;
      if (sl < Ss) v = fold_slice<float>(sg.partial + (size_t)task * sg.nchunks * sg.nelem + r, (size_t)sg.nelem, sg.nchunks, Ss, sl);
      if (Ss > 1) part[sl * epw + el] = v;
      return AdvanceElem{Ss > 1 ? 3 + Ss : 1, v};          // kind > 3: the slices 0 .. kind - 4 sit in `part`
    }
  }
  if (a.b1_wgrad && e - a.off_w1 < (unsigned)(kp * a.co)) return AdvanceElem{2, 0.f};
  return AdvanceElem{0, sl == 0 ? g_t[e] : 0.f};
}

// arrival of one contributor of block 1's finished weights; true in the workgroup that arrived last (uniform per workgroup)
__device__ __forceinline__ bool advance_arrive(unsigned* counter, unsigned arrivals, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this thread's write-through stores are acknowledged (finalize.h)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (prev + 1u == arrivals);
    if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = last;
  }
  __syncthreads();
  const bool last = *flag != 0;
  if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return last;
}

__global__ __launch_bounds__(1024) void advance_kernel(AdvanceArgs a, int ng, int kp, int S, int nchunk_wg, unsigned arrivals) {
  extern __shared__ double sm[];
  const int task = blockIdx.y, tid = threadIdx.x;
  float* part = reinterpret_cast<float*>(sm);            // [S][epw] slice sums (4 KB); row workgroups: [slices][co] doubles
  int* flag = reinterpret_cast<int*>(sm + 512);          // (byte 4096)
  const unsigned w1_lo = a.off_w1, w1_n = (unsigned)(kp * a.co);
  const float* res_t = (a.out ? a.out : a.g) + (size_t)task * (a.out ? a.ostride : a.gstride);    // where the finished vector lives
  bool contributor = false;
  if ((int)blockIdx.x < nchunk_wg) {
    const int epw = 1024 / S, el = tid % epw, sl = tid / epw;
    const unsigned e = blockIdx.x * (unsigned)epw + el;
    AdvanceElem r{2, 0.f};
    if (e < a.n) r = advance_classify(a, task, e, kp, S, sl, part, epw, el);
    if (S > 1) __syncthreads();
    if (sl == 0 && e < a.n && r.kind != 2) {
      float gv = r.v;
      if (r.kind > 3) {
        gv = part[el];
        for (int q = 1; q < r.kind - 3; ++q) gv += part[q * epw + el];
      }
      const bool in_w1 = a.stats && !a.b1_wgrad && e - w1_lo < w1_n;     // a contributor's element: must be visible to the last arriver
      float* g_t = a.g + (size_t)task * a.gstride;
      if (r.kind != 0) {
        if (in_w1 && !a.out) __hip_atomic_store(g_t + e, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else g_t[e] = gv;
      }
      if (a.out) {
        const float res = a.a[(size_t)task * a.ostride + e] - a.alpha * gv;
        float* o = a.out + (size_t)task * a.ostride + e;
        if (in_w1) __hip_atomic_store(o, res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *o = res;
      }
    }
    // (uniform per workgroup) does this chunk hold any of block 1's weights while no row workgroups exist?
    const unsigned lo = blockIdx.x * (unsigned)epw, hi = lo + (unsigned)epw;
    contributor = a.stats && !a.b1_wgrad && lo < w1_lo + w1_n && hi > w1_lo;
  } else {
    // ---- row workgroup: rows k of dW1[k][.] of block 1 from the Gram matrix -- 1024 / (co * Sb) rows per workgroup, thread
    // (row, slice, channel): two workgroups per task at 32 tasks per call (2 slices), seven at 4 tasks or fewer (8 slices)
    const int co = a.co;
    const int Sb = fold_slices(a.gw.nblk);
    const int rpw = 1024 / (co * Sb);
    const int c = tid % co, sl = (tid / co) % Sb, rl = tid / (co * Sb);
    const int k = ((int)blockIdx.x - nchunk_wg) * rpw + rl;
    const bool live = rl < rpw && k < kp;
    double* dpart = sm + 520;                              // [rpw][Sb][co] doubles (behind part / flag)
    double sv = 0.0;
    if (live) sv = fold_slice<double>(gram_wgrad_partials(a.gw, kp, task, k, c), (size_t)kp * co, a.gw.nblk, Sb, sl);
    if (Sb > 1) {
      if (live) dpart[(rl * Sb + sl) * co + c] = sv;
      __syncthreads();
    }
    if (live && sl == 0) {
      double Ssum = sv;
      for (int q = 1; q < Sb; ++q) Ssum += dpart[(rl * Sb + q) * co + c];
      const float gv = gram_wgrad_elem(a.gw, ng, kp, a.gw_tangent, task, k, c, Ssum);
      const unsigned e = w1_lo + (unsigned)(k * co + c);
      float* g_t = a.g + (size_t)task * a.gstride;
      if (a.stats && !a.out) __hip_atomic_store(g_t + e, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else g_t[e] = gv;
      if (a.out) {
        const float res = a.a[(size_t)task * a.ostride + e] - a.alpha * gv;
        float* o = a.out + (size_t)task * a.ostride + e;
        if (a.stats) __hip_atomic_store(o, res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *o = res;
      }
    }
    contributor = a.stats != 0;
  }
  if (!contributor) return;                                // uniform per workgroup
  if (!advance_arrive(a.counter + task, arrivals, flag)) return;
  // ---- last arriver: block 1's BatchNorm statistics of the next pass from the finished weights (or direction)
  const int co = a.co, tangent = a.stats == 2;
  double* gs = sm + 520;
  double* ws = gs + ng * ng;
  double* vs = ws + kp * co;
  double* qd = vs + (tangent ? kp * co : 0);
  __syncthreads();                                         // (dpart is dead)
  for (int e = tid; e < ng * ng; e += 1024) gs[e] = a.gram[(size_t)task * ng * ng + e];
  for (int e = tid; e < kp * co; e += 1024) {
    const double fin = (double)__hip_atomic_load(res_t + w1_lo + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tangent) {
      ws[e] = (double)a.sw[(size_t)task * a.swstride + e];
      vs[e] = fin;
    } else {
      ws[e] = fin;
    }
  }
  __syncthreads();
  if (!tangent) vs = ws;
  gram_stats_body<1024>(gs, ws, vs, qd, ng, kp, co, a.inv_m, tangent, a.out0, a.out1, a.mu_in, a.rstd_in, task);
}

static int sparse_row_pitch(int w, int ci, bool bf = false) {
  int rp = (w + 4) * ci;                       // halo pixel + row + halo pixel (+ the pixels an odd row's last half window reaches)
  if (rp < ci + 256) rp = ci + 256;            // a row store writes 4 x 64 lanes (zeros past the row)
  // fp32 rows: pitch = 12 (mod 32) floats, the three tap rows on disjoint banks.  Split rows (8 bytes per value, ds_read_b64: 64 banks per
  // 32-lane half): the taps of a row span 18 dwords, 2 * pitch = 20 (mod 64) dwords keeps the three rows' spans apart
  while (rp % 32 != (bf ? 10 : 12)) ++rp;
  return rp;
}
// one input row = at most 4 floats per lane; 32 or 64 filters per column tile group; per-task tensors addressable by 32-bit offsets
bool sparse_wgrad_supported(int w, int ci, int co) { return w * ci <= 256 && (ci == 1 || ci == 3) && (co == 32 || co == 64); }
// The rows kernel on the split-bf16 operand form (84-wide RGB input).  -1 = follow the hidden blocks' operand form (split forms: on; fp32 pipe:
// off -- bench.py's fp32_pipe leg and the tests' fp32_pipe parameter stay on fp32-input MFMAs throughout); MI_SPARSE_WGRAD_BF16 = 0 / 1 or
// mi_sparse_wgrad_set_split_bf16 force it.
static int g_sparse_bf = -2;
static bool sparse_wgrad_bf() {
  if (g_sparse_bf == -2) {
    const char* e = getenv("MI_SPARSE_WGRAD_BF16");
    g_sparse_bf = e ? (atoi(e) != 0) : -1;
  }
  return g_sparse_bf >= 0 ? g_sparse_bf != 0 : conv_operand_form() != 0;
}
extern "C" int mi_sparse_wgrad_set_split_bf16(int on) {      // on < 0: back to following the hidden blocks' form; returns the form in force before
  const int was = sparse_wgrad_bf() ? 1 : 0;
  g_sparse_bf = on < 0 ? -1 : (on != 0);
  return was;
}
int sparse_wgrad_split_form() { return sparse_wgrad_bf() ? 1 : 0; }
static bool sparse_rows_bf(int w, int ci) { return ci == 3 && w == 84 && sparse_wgrad_bf(); }
static void sparse_wgrad_grid(int n, int h, int w, int co, int tasks, int& ntiles, int& tpw, dim3& grid, bool bf = false) {
  ntiles = n * (h / 2);                       // pooled rows
  // 4 workgroups of 4 waves are resident per CU (LDS, 128 VGPRs; the split form: 2, its rows take twice the LDS): aim at exactly one resident
  // set (1024 / 512 workgroups) with even shares -- a last, partly filled round of workgroups costs its full time, and so does a share one
  // tile larger on a few SIMDs only
  const long col_tiles = (long)tasks * (co / 32);
  long nblk = (bf ? 512 : 1024) / col_tiles;
  if (nblk < 1) nblk = 1;
  if (nblk > ceil_div(ntiles, 4)) nblk = ceil_div(ntiles, 4);       // at least one tile per wave
  tpw = ceil_div(ceil_div(ntiles, (int)nblk), 4);
  grid = dim3((unsigned)nblk, tasks, co / 32);
}
int sparse_wgrad_blocks_per_task(int n, int h, int w, int co, int tasks) {
  int ntiles, tpw;
  dim3 grid;
  sparse_wgrad_grid(n, h, w, co, tasks, ntiles, tpw, grid, false);      // (sizes the partial buffers: the larger of the two forms' needs)
  const int fp32_blocks = (int)grid.x;
  sparse_wgrad_grid(n, h, w, co, tasks, ntiles, tpw, grid, true);
  return fp32_blocks > 2 * (int)grid.x ? fp32_blocks : 2 * (int)grid.x;
}
static hipError_t sparse_dynamic_lds(const void* k, size_t lds, unsigned* done_mask) {     // (conv_mfma.hip's ensure_dynamic_lds: once per kernel and device)
  if (lds <= 64 * 1024) return hipSuccess;
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  const unsigned bit = 1u << (dev & 31);
  if (*done_mask & bit) return hipSuccess;
  if (hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e != hipSuccess) return e;
  *done_mask |= bit;
  return hipSuccess;
}
template <bool TAN, int CO>
static hipError_t launch_sparse_rows_bf(hipStream_t st, dim3 grid, size_t smem, const SparseWgArgs& a) {
  auto k = sparse_wgrad_rows_kernel<3, TAN, CO, 3, 7, true>;
  static unsigned attr_done = 0;
  if (hipError_t e = sparse_dynamic_lds(reinterpret_cast<const void*>(k), smem, &attr_done); e != hipSuccess) return e;
  hipLaunchKernelGGL(k, grid, dim3(256), smem, st, a);
  return hipGetLastError();
}
template <int CI, bool TAN>
static hipError_t launch_sparse_t(hipStream_t st, dim3 grid, size_t smem, const SparseWgArgs& a, bool rows, bool bf) {
  if (CI == 3 && rows && bf) return a.co == 32 ? launch_sparse_rows_bf<TAN, 32>(st, grid, smem, a) : launch_sparse_rows_bf<TAN, 64>(st, grid, smem, a);
  if (CI == 3 && rows && a.co == 32) hipLaunchKernelGGL((sparse_wgrad_rows_kernel<3, TAN, 32, 3, 7>), grid, dim3(256), smem, st, a);
  else if (CI == 3 && rows) hipLaunchKernelGGL((sparse_wgrad_rows_kernel<3, TAN, 64, 3, 7>), grid, dim3(256), smem, st, a);
  else if (a.co == 32) hipLaunchKernelGGL((sparse_wgrad_kernel<CI, TAN, 32, 8>), grid, dim3(256), smem, st, a);
  else hipLaunchKernelGGL((sparse_wgrad_kernel<CI, TAN, 64, 8>), grid, dim3(256), smem, st, a);
  return hipGetLastError();
}
hipError_t launch_sparse_wgrad(hipStream_t st, SparseWgArgs a, int tasks, int ci, int tangent, int* blocks_per_task) {
  int ntiles, tpw;
  dim3 grid;
  if (!sparse_wgrad_supported(a.ww, ci, a.co)) return hipErrorInvalidValue;
  if (a.co != 32 && a.co != 64) return hipErrorInvalidValue;
  const bool rows = ci == 3 && a.ww == 84;       // rows of exactly 7 chunks of 3 steps: the mini-ImageNet input
  const bool bf = rows && sparse_rows_bf(a.ww, ci);
  sparse_wgrad_grid(a.n, a.hh, a.ww, a.co, tasks, ntiles, tpw, grid, bf);
  a.ntiles = ntiles;
  a.tiles_per_wave = tpw;
  a.row_pitch = sparse_row_pitch(a.ww, ci, bf);
  size_t smem = (size_t)a.row_pitch * 32 * sizeof(float) * (bf ? 2 : 1);          // 4 waves x 2 sets x 4 rows
  if (smem < 4 * 1024 * sizeof(float)) smem = 4 * 1024 * sizeof(float);
  if (blocks_per_task) *blocks_per_task = bf ? 2 * grid.x : grid.x;        // (the split form writes every workgroup's fp64 sum as two fp32 partials)
  if (ci == 3) {
    return tangent ? launch_sparse_t<3, true>(st, grid, smem, a, rows, bf) : launch_sparse_t<3, false>(st, grid, smem, a, rows, bf);
  } else if (ci == 1) {
    return tangent ? launch_sparse_t<1, true>(st, grid, smem, a, rows, bf) : launch_sparse_t<1, false>(st, grid, smem, a, rows, bf);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_gram_wgrad(hipStream_t st, const GramWgArgs& a, int tasks, int tangent) {
  const int ng = a.ci == 3 ? GramDims<3>::NG : GramDims<1>::NG, kp = 9 * a.ci;
  hipLaunchKernelGGL(gram_wgrad_kernel, dim3(tasks, kp), dim3(64), 0, st, a, ng, kp, tangent);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
static int gram_ng(int ci) { return ci == 3 ? GramDims<3>::NG : GramDims<1>::NG; }
static const int kGramRowsPerWave = 8;

int gram_blocks_per_task(int n, int h) { return ceil_div(n * h, 4 * kGramRowsPerWave); }
size_t gram_partial_doubles(int tasks, int n, int h, int ci) {
  return (size_t)tasks * gram_blocks_per_task(n, h) * gram_ng(ci) * gram_ng(ci);
}
bool gram_supported(int w, int ci) { return (size_t)4 * 3 * (w + 6) * ci * sizeof(double) <= 64 * 1024; }
size_t gram_doubles(int tasks, int ci) { return (size_t)tasks * gram_ng(ci) * gram_ng(ci); }

// x [T][n][H][W][ci] -> g [T][NG][NG] (row/column 9*ci = the constant-one entry: G[a][9ci] = s[a], G[9ci][9ci] = pixel count)
hipError_t launch_input_gram(hipStream_t st, const float* x, int tasks, int n, int h, int w, int ci, double* partial, double* g) {
  const int nblk = gram_blocks_per_task(n, h), ng = gram_ng(ci);
  // row pitch in doubles: halo pixel + row + halo pixel + 4 pixels of zero padding for the last 4-pixel step
  const int rpd = (w + 2 + 4) * ci;
  const int nt = ng / 16;
  size_t smem = (size_t)4 * 3 * rpd * sizeof(double);
  const size_t red = (size_t)4 * nt * nt * 256 * sizeof(double);
  if (smem < red) smem = red;
  if (smem > 64 * 1024) return hipErrorInvalidValue;      // W up to ~670 (ci 1) / ~220 (ci 3)
  if (ci == 3)
    hipLaunchKernelGGL(input_gram_kernel<3>, dim3(nblk, tasks), dim3(256), smem, st, x, n, h, w, kGramRowsPerWave, rpd, partial);
  else if (ci == 1)
    hipLaunchKernelGGL(input_gram_kernel<1>, dim3(nblk, tasks), dim3(256), smem, st, x, n, h, w, kGramRowsPerWave, rpd, partial);
  else
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(ceil_div(ng * ng, 256), tasks), dim3(256), 0, st, partial, nblk, ng, g);
  return hipGetLastError();
}

hipError_t launch_advance(hipStream_t st, const AdvanceArgs& a_in, int tasks) {
  AdvanceArgs a = a_in;
  // timing experiments (wrong results): MI_ADV_DBG bit 1 = no Gram statistics, bit 2 = no Gram assembly of block 1's weight gradient,
  // bit 3 = no weight-gradient folds
  static const int dbg = getenv("MI_ADV_DBG") ? atoi(getenv("MI_ADV_DBG")) : 0;
  if (dbg & 2) a.stats = 0;
  if (dbg & 4) a.b1_wgrad = 0;
  if (dbg & 8) a.nseg = 0;
  const int ng = a.ci ? gram_ng(a.ci) : 0, kp = 9 * a.ci;
  if ((a.stats || a.b1_wgrad) && (a.ci != 1 && a.ci != 3)) return hipErrorInvalidValue;
  if (a.nseg > 8 || a.nzero > 10) return hipErrorInvalidValue;
  if (a.stats && !a.counter) return hipErrorInvalidValue;
  if (a.b1_wgrad && a.out && a.out == a.a) return hipErrorInvalidValue;      // the row workgroups read what an in-place update overwrites
  if ((a.stats || a.b1_wgrad) && (1024 % a.co != 0)) return hipErrorInvalidValue;
  int S = 1;
  for (int q = 0; q < a.nseg; ++q) { const int sq = fold_slices(a.seg[q].nchunks); if (sq > S) S = sq; }
  const int epw = 1024 / S;
  const int nchunk_wg = ceil_div((int)a.n, epw);
  const int nrow_wg = a.b1_wgrad ? ceil_div(kp, 1024 / (a.co * fold_slices(a.gw.nblk))) : 0;
  unsigned arrivals = 0;
  if (a.stats) {
    if (a.b1_wgrad) arrivals = (unsigned)nrow_wg;
    else arrivals = (unsigned)(ceil_div((int)a.off_w1 + kp * a.co, epw) - (int)a.off_w1 / epw);
  }
  // LDS: slice sums / flag (4096 + 64 B), then the row workgroups' fp64 slice sums and, in the last arriver, the statistics tables
  size_t smem = 4160 + (size_t)1024 * sizeof(double);
  if (a.stats) smem = 4160 + ((size_t)ng * ng + (size_t)(a.stats == 2 ? 3 : 2) * kp * a.co) * sizeof(double);
  if (smem < 4160 + (size_t)1024 * sizeof(double)) smem = 4160 + (size_t)1024 * sizeof(double);
  hipLaunchKernelGGL(advance_kernel, dim3(nchunk_wg + nrow_wg, tasks), dim3(1024), smem, st, a, ng, kp, S, nchunk_wg, arrivals);
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

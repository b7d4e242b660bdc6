// Batch-stat BatchNorm + ReLU + MaxPool2d(2,2) for a whole meta-batch: forward, backward, and their tangents
// (R-operator) -- the elementwise / per-channel-reduction half of ConvBlock.forward
// (reference core_functions/vision_models.py:188-193: normalize -> relu -> max_pool; BatchNorm2d always in TRAIN mode,
// SURVEY.md section 0.4), replacing ATen batch_norm / threshold_backward / max_pool2d_with_indices and their
// backward / double-backward.
//
// All kernels are HBM-streaming: NHWC, one thread owns a 2x2 pooling window (or one pixel without pooling) x 4 channels
// (one 16-B load per position), 8 (C=32) or 16 (C=64) consecutive lanes cover one pixel's channel vector, consecutive
// lane groups walk consecutive windows => fully coalesced 128/256-B segments.  Nothing but z (conv output) is stored by
// the forward: the ReLU mask and pooling argmax are re-derived from z in every backward/tangent kernel with bit-identical
// arithmetic (bn_zh/bn_u).  Per-channel reductions accumulate in fp64 per thread, reduce through LDS and leave one
// deterministic partial per workgroup; bn_finalize_kernel folds the partials in a fixed order.
#include "mi_common.h"
#include "kernels.h"

#include "bn_window.h"

#define BN_THREAD_SETUP(POOL)                                                                   \
  const int quads = a.c >> 2;                                                                   \
  const int quad = threadIdx.x % quads, wl = threadIdx.x / quads, wpb = 256 / quads;            \
  const int task = blockIdx.y, c0 = quad * 4;                                                   \
  const WinIter<POOL> it(a);                                                                    \
  const size_t z_task = (size_t)a.n * a.ho * a.wo * a.c;                                        \
  const size_t p_task = (size_t)a.n * it.hp * it.wp * a.c;                                      \
  const ChanConst k = load_consts(a, task, c0);                                                 \
  const float* z_t = a.z + (size_t)task * z_task;                                               \
  (void)p_task; (void)z_t;

// fp64 block reduction of 8 per-thread accumulators (4 channels x 2 quantities) -> partial[task][blk][2][c]
__device__ __forceinline__ void block_reduce_write(const double* acc0, const double* acc1, const BnArgs& a, int quads,
                                                   int task) {
  __shared__ double red[256 * 8];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    red[threadIdx.x * 8 + c] = acc0[c];
    red[threadIdx.x * 8 + 4 + c] = acc1[c];
  }
  __syncthreads();
  if ((int)threadIdx.x < a.c) {
    const int q = threadIdx.x >> 2, comp = threadIdx.x & 3, wpb = 256 / quads;
    double s0 = 0.0, s1 = 0.0;
    for (int w = 0; w < wpb; ++w) {
      s0 += red[(w * quads + q) * 8 + comp];
      s1 += red[(w * quads + q) * 8 + 4 + comp];
    }
    double* pb = a.partial + ((size_t)task * gridDim.x + blockIdx.x) * 2 * a.c;
    mi_partial_store(pb + threadIdx.x, s0, a.fin);
    mi_partial_store(pb + a.c + threadIdx.x, s1, a.fin);
  }
  mi_finalize_last(a.fin, a.partial + (size_t)task * gridDim.x * 2 * a.c, gridDim.x, a.c, task, gridDim.x, red);
}

// ---------------------------------------------------------------------------------------------------------------------
template <int POOL>
__global__ __launch_bounds__(256) void bn_fwd_kernel(BnArgs a) {
  BN_THREAD_SETUP(POOL)
  float* out_t = a.out + (size_t)task * p_task;
  float* zh_t = a.zh_out ? a.zh_out + (size_t)task * p_task : nullptr;
  unsigned am = 0u;                                          // largest magnitude written (BnArgs::amax_out)
  Window<POOL> w;
  for (int win = blockIdx.x * wpb + wl; win < it.nwin; win += gridDim.x * wpb) {
    w.locate(a, it, win, c0);
    if (!w.pooled) continue;
    floatx4 umax, zh_at, zd_at;
    scan_window<POOL, false>(w, z_t, nullptr, k, umax, zh_at, zd_at);      // running maximum, zhat carried along (no per-position arrays)
    floatx4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) { o[c] = fmaxf(umax[c], 0.f); mi_amax_acc(am, o[c]); }
    *reinterpret_cast<floatx4*>(out_t + w.poff) = o;
    if (zh_t) *reinterpret_cast<floatx4*>(zh_t + w.poff) = zh_at;
  }
  if (a.amax_out) mi_amax_commit(am, a.amax_out, task);
}

template <int POOL>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnArgs a) {
  BN_THREAD_SETUP(POOL)
  const float* dp_t = a.dp + (size_t)task * p_task;
  double dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
  Window<POOL> w;
  for (int win = blockIdx.x * wpb + wl; win < it.nwin; win += gridDim.x * wpb) {
    w.locate(a, it, win, c0);
    if (!w.pooled) continue;                              // pooled windows have all their positions inside the image
    floatx4 umax, zh_at, zd_at;
    scan_window<POOL, false>(w, z_t, nullptr, k, umax, zh_at, zd_at);
    const floatx4 d = *reinterpret_cast<const floatx4*>(dp_t + w.poff);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float du = umax[c] > 0.f ? d[c] : 0.f;
      db[c] += (double)du;
      dg[c] += (double)du * (double)zh_at[c];
    }
  }
  block_reduce_write(dg, db, a, quads, task);
}

template <int POOL>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnArgs a) {
  BN_THREAD_SETUP(POOL)
  const float* dp_t = a.dp + (size_t)task * p_task;
  float* out_t = a.out + (size_t)task * z_task;
  float dgm[4], dbm[4], gr[4];
  load4(a.dgamma + (size_t)task * a.gstride + c0, dgm);
  load4(a.dbeta + (size_t)task * a.gstride + c0, dbm);
#pragma unroll
  for (int c = 0; c < 4; ++c) { dgm[c] *= a.inv_m; dbm[c] *= a.inv_m; gr[c] = k.g[c] * k.r[c]; }
  unsigned am = 0u;                                          // largest magnitude written (BnArgs::amax_out)
  Window<POOL> w;
  for (int win = blockIdx.x * wpb + wl; win < it.nwin; win += gridDim.x * wpb) {
    w.locate(a, it, win, c0);
    w.analyse(z_t, k);
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    if (w.pooled) load4(dp_t + w.poff, d);
#pragma unroll
    for (int p = 0; p < Window<POOL>::NP; ++p) {
      if (!w.exists[p]) continue;
      float o[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float du = (w.pooled && w.arg[c] == p && w.umax[c] > 0.f) ? d[c] : 0.f;
        o[c] = gr[c] * (du - dbm[c] - w.zh[p][c] * dgm[c]);
        mi_amax_acc(am, o[c]);
      }
      store4(out_t + w.off[p], o);
    }
  }
  if (a.amax_out) mi_amax_commit(am, a.amax_out, task);
}

// tangent forward: pd = [u>0 at argmax] * (gammad*zh + gamma*zhd + betad),  zhd = r (zd - m1 - zh m2)
template <int POOL>
__global__ __launch_bounds__(256) void bn_tan_fwd_kernel(BnArgs a) {
  BN_THREAD_SETUP(POOL)
  const float* zd_t = a.zd + (size_t)task * z_task;
  float* out_t = a.out + (size_t)task * p_task;
  float m1[4], m2[4], gd[4], bd[4];
  load4(a.m1 + (size_t)task * a.c + c0, m1);
  load4(a.m2 + (size_t)task * a.c + c0, m2);
  load4(a.gammad + (size_t)task * a.vstride + c0, gd);
  load4(a.betad + (size_t)task * a.vstride + c0, bd);
  unsigned am = 0u;                                          // largest magnitude written (BnArgs::amax_out)
  Window<POOL> w;
  for (int win = blockIdx.x * wpb + wl; win < it.nwin; win += gridDim.x * wpb) {
    w.locate(a, it, win, c0);
    if (!w.pooled) continue;
    floatx4 umax, zh_at, zd_at;
    scan_window<POOL, true>(w, z_t, zd_t, k, umax, zh_at, zd_at);
    floatx4 o, zo;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float zhd = k.r[c] * (zd_at[c] - m1[c] - zh_at[c] * m2[c]);
      const float ud = gd[c] * zh_at[c] + k.g[c] * zhd + bd[c];
      o[c] = (umax[c] > 0.f) ? ud : 0.f;
      zo[c] = zhd;
      mi_amax_acc(am, o[c]);
    }
    *reinterpret_cast<floatx4*>(out_t + w.poff) = o;
    if (a.zh_out) *reinterpret_cast<floatx4*>(a.zh_out + (size_t)task * p_task + w.poff) = zo;
  }
  if (a.amax_out) mi_amax_commit(am, a.amax_out, task);
}

// tangent backward reductions: R{dbeta} = sum dud ; R{dgamma} = sum (dud zh + du zhd)   (argmax positions only)
template <int POOL>
__global__ __launch_bounds__(256) void bn_tan_bwd_reduce_kernel(BnArgs a) {
  BN_THREAD_SETUP(POOL)
  const float* zd_t = a.zd + (size_t)task * z_task;
  const float* dp_t = a.dp + (size_t)task * p_task;
  const float* dpd_t = a.dpd + (size_t)task * p_task;
  float m1[4], m2[4];
  load4(a.m1 + (size_t)task * a.c + c0, m1);
  load4(a.m2 + (size_t)task * a.c + c0, m2);
  double rg[4] = {0, 0, 0, 0}, rb[4] = {0, 0, 0, 0};
  Window<POOL> w;
  for (int win = blockIdx.x * wpb + wl; win < it.nwin; win += gridDim.x * wpb) {
    w.locate(a, it, win, c0);
    if (!w.pooled) continue;
    floatx4 umax, zh_at, zd_at;
    scan_window<POOL, true>(w, z_t, zd_t, k, umax, zh_at, zd_at);
    const floatx4 d = *reinterpret_cast<const floatx4*>(dp_t + w.poff);
    const floatx4 dd = *reinterpret_cast<const floatx4*>(dpd_t + w.poff);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool on = umax[c] > 0.f;
      const float du = on ? d[c] : 0.f, dud = on ? dd[c] : 0.f;
      const float zhd = k.r[c] * (zd_at[c] - m1[c] - zh_at[c] * m2[c]);
      rb[c] += (double)dud;
      rg[c] += (double)dud * (double)zh_at[c] + (double)du * (double)zhd;
    }
  }
  block_reduce_write(rg, rb, a, quads, task);
}

// tangent backward apply:
//   E = du - dbeta/M - zh dgamma/M
//   R{dz} = (gammad r + gamma rd) E + gamma r (dud - R{dbeta}/M - zhd dgamma/M - zh R{dgamma}/M),  rd = -r^2 m2
template <int POOL>
__global__ __launch_bounds__(256) void bn_tan_bwd_apply_kernel(BnArgs a) {
  BN_THREAD_SETUP(POOL)
  const float* zd_t = a.zd + (size_t)task * z_task;
  const float* dp_t = a.dp + (size_t)task * p_task;
  const float* dpd_t = a.dpd + (size_t)task * p_task;
  float* out_t = a.out + (size_t)task * z_task;
  float m1[4], m2[4], gd[4], dgm[4], dbm[4], rgm[4], rbm[4], c1[4], gr[4];
  load4(a.m1 + (size_t)task * a.c + c0, m1);
  load4(a.m2 + (size_t)task * a.c + c0, m2);
  load4(a.gammad + (size_t)task * a.vstride + c0, gd);
  load4(a.dgamma + (size_t)task * a.gstride + c0, dgm);
  load4(a.dbeta + (size_t)task * a.gstride + c0, dbm);
  load4(a.rdgamma + (size_t)task * a.hstride + c0, rgm);
  load4(a.rdbeta + (size_t)task * a.hstride + c0, rbm);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    dgm[c] *= a.inv_m; dbm[c] *= a.inv_m; rgm[c] *= a.inv_m; rbm[c] *= a.inv_m;
    const float rd = -k.r[c] * k.r[c] * m2[c];
    c1[c] = gd[c] * k.r[c] + k.g[c] * rd;
    gr[c] = k.g[c] * k.r[c];
  }
  unsigned am = 0u;                                          // largest magnitude written (BnArgs::amax_out)
  Window<POOL> w;
  for (int win = blockIdx.x * wpb + wl; win < it.nwin; win += gridDim.x * wpb) {
    w.locate(a, it, win, c0);
    w.analyse(z_t, k);
    float d[4] = {0.f, 0.f, 0.f, 0.f}, dd[4] = {0.f, 0.f, 0.f, 0.f};
    if (w.pooled) { load4(dp_t + w.poff, d); load4(dpd_t + w.poff, dd); }
#pragma unroll
    for (int p = 0; p < Window<POOL>::NP; ++p) {
      if (!w.exists[p]) continue;
      float zdv[4], o[4];
      load4(zd_t + w.off[p], zdv);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool sel = w.pooled && w.arg[c] == p && w.umax[c] > 0.f;
        const float du = sel ? d[c] : 0.f, dud = sel ? dd[c] : 0.f;
        const float zh = w.zh[p][c];
        const float zhd = k.r[c] * (zdv[c] - m1[c] - zh * m2[c]);
        const float e = du - dbm[c] - zh * dgm[c];
        o[c] = c1[c] * e + gr[c] * (dud - rbm[c] - zhd * dgm[c] - zh * rgm[c]);
        mi_amax_acc(am, o[c]);
      }
      store4(out_t + w.off[p], o);
    }
  }
  if (a.amax_out) mi_amax_commit(am, a.amax_out, task);
}

// ---------------------------------------------------------------------------------------------------------------------
// Fold per-workgroup fp64 partials [T][nblk][2][c] in block order.
//  FIN_STATS : (sum z, sum z^2)        -> out0 = mean, out1 = 1/sqrt(biased var + eps)
//  FIN_TSTATS: (sum zd, sum zh zd)     -> out0 = m1,   out1 = m2
//  FIN_SUMS  : (first, second)         -> out0 = first, out1 = second (e.g. dgamma, dbeta)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ partial, int nblk, int c, FinArgs f) {
  // one workgroup per task (the stand-alone form of the fold the producers run in their last workgroup, finalize.h)
  __shared__ double red[2 * 256];
  mi_fold_partials<false>(partial + (size_t)blockIdx.x * nblk * 2 * c, nblk, c, f, blockIdx.x, red);
}

// ---------------------------------------------------------------------------------------------------------------------
// dgamma / dbeta (and their tangents) of a fused block 1 from pooled-resolution tensors.  The fused forward kernels store
// zhat (TFWD: its tangent) at every window's argmax; the cotangent dp reaches z only there and only if the maximum passed
// the ReLU (p > 0), so   dgamma = sum [p>0] dp zhat,  dbeta = sum [p>0] dp,
//                        R{dgamma} = sum [p>0] (dpd zhat + dp zhatd),  R{dbeta} = sum [p>0] dpd
// -- the sums block1_kernel<BWD_REDUCE / TBWD_REDUCE> forms after recomputing conv1, here as one streaming pass.
template <bool TAN>
__global__ __launch_bounds__(256) void pooled_reduce_kernel(PoolRedArgs a) {
  __shared__ double red[256 * 8];
  const int task = blockIdx.y, quads = a.c >> 2, wpb = 256 / quads;
  const int q = threadIdx.x % quads, rl = threadIdx.x / quads;
  const size_t base = (size_t)task * a.rows * a.c + 4 * q;
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  if (rl < wpb)
    for (int row = blockIdx.x * wpb + rl; row < a.rows; row += gridDim.x * wpb) {
      const size_t off = base + (size_t)row * a.c;
      float p[4], d[4], zh[4], dd[4], zhd[4];
      load4(a.p + off, p); load4(a.dp + off, d); load4(a.zh + off, zh);
      if (TAN) { load4(a.dpd + off, dd); load4(a.zhd + off, zhd); }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool on = p[c] > 0.f;
        const float dv = on ? d[c] : 0.f;
        if (!TAN) {
          s0[c] += (double)dv * (double)zh[c];
          s1[c] += (double)dv;
        } else {
          const float ddv = on ? dd[c] : 0.f;
          s0[c] += (double)ddv * (double)zh[c] + (double)dv * (double)zhd[c];
          s1[c] += (double)ddv;
        }
      }
    }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    red[threadIdx.x * 8 + c] = s0[c];
    red[threadIdx.x * 8 + 4 + c] = s1[c];
  }
  __syncthreads();
  if ((int)threadIdx.x < a.c) {
    const int qq = threadIdx.x >> 2, comp = threadIdx.x & 3;
    double t0 = 0.0, t1 = 0.0;
    for (int w = 0; w < wpb; ++w) {
      t0 += red[(w * quads + qq) * 8 + comp];
      t1 += red[(w * quads + qq) * 8 + 4 + comp];
    }
    double* pb = a.partial + ((size_t)task * gridDim.x + blockIdx.x) * 2 * a.c;
    mi_partial_store(pb + threadIdx.x, t0, a.fin);
    mi_partial_store(pb + a.c + threadIdx.x, t1, a.fin);
  }
  mi_finalize_last(a.fin, a.partial + (size_t)task * gridDim.x * 2 * a.c, gridDim.x, a.c, task, gridDim.x, red);
}

int pooled_reduce_blocks(int rows, int c, int tasks) {
  const int wpb = 256 / (c / 4);
  int blocks = ceil_div(rows, wpb);
  const int cap = ceil_div(2048, tasks);
  if (blocks > cap) blocks = cap;
  return blocks < 1 ? 1 : blocks;
}
hipError_t launch_pooled_reduce(hipStream_t st, const PoolRedArgs& a, int tasks, int tangent, int* nblk) {
  if (a.c % 4 || a.c > 256 || 256 % (a.c / 4)) return hipErrorInvalidValue;
  const int blocks = pooled_reduce_blocks(a.rows, a.c, tasks);
  if (tangent) hipLaunchKernelGGL(pooled_reduce_kernel<true>, dim3(blocks, tasks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(pooled_reduce_kernel<false>, dim3(blocks, tasks), dim3(256), 0, st, a);
  *nblk = blocks;
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
int bn_blocks_per_task(int n, int ho, int wo, int c, int pool, int tasks) {
  const int nwin = pool ? n * ((ho + 1) / 2) * ((wo + 1) / 2) : n * ho * wo;
  const int wpb = 256 / (c / 4);
  int blocks = ceil_div(nwin, wpb);
  // cap the grid near 8 workgroups per CU over the whole launch, grid-stride the rest
  const int cap = ceil_div(2048, tasks);
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return blocks;
}

hipError_t launch_bn_finalize(hipStream_t st, const double* partial, int nblk, int tasks, int c, double inv_m, int mode,
                              float* out0, size_t stride0, float* out1, size_t stride1) {
  if (c > 256 || 256 % c != 0) return hipErrorInvalidValue;
  const FinArgs f{nullptr, out0, out1, stride0, stride1, inv_m, mode};
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(tasks), dim3(256), 0, st, partial, nblk, c, f);
  return hipGetLastError();
}

#define BN_LAUNCH(KERNEL)                                                                          \
  const int blocks = bn_blocks_per_task(a.n, a.ho, a.wo, a.c, pool, tasks);                        \
  dim3 grid(blocks, tasks);                                                                        \
  if (pool) hipLaunchKernelGGL(KERNEL<1>, grid, dim3(256), 0, st, a);                              \
  else hipLaunchKernelGGL(KERNEL<0>, grid, dim3(256), 0, st, a);

hipError_t launch_bn_fwd(hipStream_t st, const BnArgs& a, int tasks, int pool) {
  BN_LAUNCH(bn_fwd_kernel)
  return hipGetLastError();
}
hipError_t launch_bn_bwd_reduce(hipStream_t st, const BnArgs& a, int tasks, int pool, int* nblk) {
  BN_LAUNCH(bn_bwd_reduce_kernel)
  *nblk = blocks;
  return hipGetLastError();
}
hipError_t launch_bn_bwd_apply(hipStream_t st, const BnArgs& a, int tasks, int pool) {
  BN_LAUNCH(bn_bwd_apply_kernel)
  return hipGetLastError();
}
hipError_t launch_bn_tan_fwd(hipStream_t st, const BnArgs& a, int tasks, int pool) {
  BN_LAUNCH(bn_tan_fwd_kernel)
  return hipGetLastError();
}
hipError_t launch_bn_tan_bwd_reduce(hipStream_t st, const BnArgs& a, int tasks, int pool, int* nblk) {
  BN_LAUNCH(bn_tan_bwd_reduce_kernel)
  *nblk = blocks;
  return hipGetLastError();
}
hipError_t launch_bn_tan_bwd_apply(hipStream_t st, const BnArgs& a, int tasks, int pool) {
  BN_LAUNCH(bn_tan_bwd_apply_kernel)
  return hipGetLastError();
}

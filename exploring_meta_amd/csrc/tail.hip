// The tail of a pass, one workgroup per task: the LAST ConvBlock's BatchNorm + ReLU + MaxPool, the classifier head with its
// cross-entropy, the head's backward and the last block's BatchNorm backward (sums + apply) -- or their tangents -- in ONE launch.
//
// Replaces (reference): the last `normalize -> relu -> max_pool` of ConvBase (core_functions/vision_models.py:188-193), `self.linear(x.view(-1,
// 25*hidden))` (:109), `loss(learner(adapt_data), adapt_labels)` / `accuracy` (core_functions/vision.py:11,16-18,21-23) and the autograd backward /
// double-backward of those ops -- five launches per pass before (bn_fwd, head_rows, head_grads, bn_bwd_reduce, bn_bwd_apply; tangent: bn_tan_fwd,
// head_rows<T>, head_grads<T>, bn_tan_bwd_reduce, bn_tan_bwd_apply).
//
// Why one workgroup per task and not a cluster with in-kernel barriers: at this end of the net a task's tensors are small (z of the last block:
// 25 images x 10 x 10 x 32 floats = 320 KB) and every stage needs ALL of the task's rows or columns (the head transposes rows into columns, the
// BatchNorm sums run over every image), so a cluster would exchange its whole working set through memory at every stage behind a cross-workgroup
// hand-off that costs what a kernel boundary costs on this chip (MI355X_MICROARCH.md price list: barrier 4-7 us, boundary 1.5-2 us).  Inside ONE
// workgroup the stages are separated by workgroup barriers (a hundred cycles), everything a stage hands to the next stays in this CU's L1 / the
// XCD's L2, and nothing depends on dispatch order or placement.  32 tasks keep 32 CUs busy for ~15 us instead of 256 CUs for five latency-bound
// launches of 7-19 us each; one task per call pays one launch instead of five.
//
// Arithmetic: the stage bodies are the ones the separate kernels run (bn_window.h, head_bodies.h) -- same instructions, same order -- so p, the
// logits, prob / dlogits, loss, accuracy, dWl, dbl, df are bit-identical to the separate launches.  The BatchNorm-backward sums (dgamma, dbeta and
// their tangents) are fp64 sums folded in a FIXED order that does not depend on the number of tasks per call (thread partials over the task's windows,
// then the threads in order); the separate kernels fold the same fp64 terms per workgroup and then across workgroups, so the two agree to the last
// bit of the fp64 sum's rounding, i.e. bit-identically in fp32 except where that sum lies within ~1e-16 of a rounding boundary.
#include "mi_common.h"
#include "kernels.h"
#include "bn_window.h"
#include "head_bodies.h"

#define TAIL_THREADS 512

#define TAIL_SETUP(POOL)                                                                          \
  const BnArgs& a = t.bn;                                                                         \
  const int tid = threadIdx.x, task = blockIdx.x;                                                 \
  const int quads = a.c >> 2;                                                                     \
  const int quad = tid % quads, wl = tid / quads, wpb = TAIL_THREADS / quads;                     \
  const int c0 = quad * 4;                                                                        \
  const WinIter<POOL> it(a);                                                                      \
  const size_t z_task = (size_t)a.n * a.ho * a.wo * a.c;                                          \
  const size_t p_task = (size_t)a.n * it.hp * it.wp * a.c;                                        \
  const ChanConst k = load_consts(a, task, c0);                                                   \
  const float* z_t = a.z + (size_t)task * z_task;

// fp64 reduction of 8 per-thread accumulators (4 channels x 2 quantities) over the workgroup, threads in order -> out0 / out1 [c] as fp32
__device__ __forceinline__ void tail_reduce_store(const double* acc0, const double* acc1, double* red, int quads, int c, float* out0, float* out1) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) {
    red[tid * 8 + ch] = acc0[ch];
    red[tid * 8 + 4 + ch] = acc1[ch];
  }
  __syncthreads();
  if (tid < c) {
    const int q = tid >> 2, comp = tid & 3, wpb = TAIL_THREADS / quads;
    double s0 = 0.0, s1 = 0.0;
    for (int w = 0; w < wpb; ++w) {
      s0 += red[(w * quads + q) * 8 + comp];
      s1 += red[(w * quads + q) * 8 + 4 + comp];
    }
    out0[tid] = (float)s0;
    out1[tid] = (float)s1;
  }
}

// the head's gradient stage for one task by the whole workgroup: two groups of 256 threads take the 64-column chunks alternately
template <bool TANGENT>
__device__ __forceinline__ void tail_head_grads(const HeadArgs& h, int task, float* sm) {
  const int tid = threadIdx.x, N = h.n, WY = h.ways;
  float* s_a = sm;
  float* s_b = sm + N * WY;
  float* s_red = sm + ((2 * N * WY + 63) & ~63) + (tid >> 8) * (3 * 8 * 64);
  head_stage_dl<TANGENT>(h, task, tid, TAIL_THREADS, s_a, s_b);
  __syncthreads();
  const int nch = (h.feat + 63) / 64, groups = TAIL_THREADS / 256;
  for (int c = tid >> 8; c < ((nch + groups - 1) / groups) * groups; c += groups) {      // (every thread the same number of trips: barriers inside)
    head_grads_chunk<TANGENT>(h, task, c, tid & 255, s_a, s_b, s_red);
    __syncthreads();                                                                     // s_red is rewritten by the next trip
  }
  head_task_sums<TANGENT>(h, task, tid, TAIL_THREADS, s_a);
}

template <int POOL>
__global__ __launch_bounds__(TAIL_THREADS) void tail_fwd_bwd_kernel(TailArgs t) {
  __shared__ double red[TAIL_THREADS * 8];
  TAIL_SETUP(POOL)
  // ---- BatchNorm + ReLU + MaxPool of the last block (bn_fwd_kernel's body)
  {
    float* out_t = t.pooled + (size_t)task * p_task;
    Window<POOL> w;
    for (int win = wl; win < it.nwin; win += wpb) {
      w.locate(a, it, win, c0);
      if (!w.pooled) continue;
      floatx4 umax, zh_at, zd_at;
      scan_window<POOL, false>(w, z_t, nullptr, k, umax, zh_at, zd_at);
      floatx4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = fmaxf(umax[c], 0.f);
      *reinterpret_cast<floatx4*>(out_t + w.poff) = o;
    }
  }
  __syncthreads();                                           // p = the head's feature rows: written by this workgroup, read by it
  // ---- Linear + cross-entropy, one wave per sample row (head_rows_kernel's body)
  const HeadArgs& h = t.hd;
  for (int n = tid >> 6; n < h.n; n += TAIL_THREADS / 64) head_row<false>(h, task, n, tid & 63);
  __syncthreads();
  if (!t.with_grad || task >= t.bwd_tasks) {                 // evaluation / validation task: loss and accuracy, nothing else (head_reduce_kernel)
    if (tid == 0 && h.loss) {
      float ls = 0.f, cs = 0.f;
      for (int r = 0; r < h.n; ++r) { ls += h.rowloss[(size_t)task * h.n + r]; cs += h.rowhit[(size_t)task * h.n + r]; }
      h.loss[task] = ls / (float)h.n;
      h.acc[task] = cs / (float)h.n;
    }
    return;
  }
  // ---- dWl, dbl, df (= the cotangent of p), loss, accuracy (head_grads_kernel's body)
  tail_head_grads<false>(h, task, reinterpret_cast<float*>(red));
  __syncthreads();
  // ---- BatchNorm backward sums: dbeta = sum [u > 0 at the argmax] dp, dgamma = sum [...] dp zhat (bn_bwd_reduce_kernel's terms)
  const float* dp_t = a.dp + (size_t)task * p_task;
  {
    double dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
    Window<POOL> w;
    for (int win = wl; win < it.nwin; win += wpb) {
      w.locate(a, it, win, c0);
      if (!w.pooled) continue;
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
    tail_reduce_store(dg, db, red, quads, a.c, t.sum0 + (size_t)task * t.sum_stride, t.sum1 + (size_t)task * t.sum_stride);
  }
  __syncthreads();
  // ---- BatchNorm backward apply: dz (bn_bwd_apply_kernel's body)
  {
    float* out_t = a.out + (size_t)task * z_task;
    float dgm[4], dbm[4], gr[4];
    load4(t.sum0 + (size_t)task * t.sum_stride + c0, dgm);
    load4(t.sum1 + (size_t)task * t.sum_stride + c0, dbm);
#pragma unroll
    for (int c = 0; c < 4; ++c) { dgm[c] *= a.inv_m; dbm[c] *= a.inv_m; gr[c] = k.g[c] * k.r[c]; }
    Window<POOL> w;
    for (int win = wl; win < it.nwin; win += wpb) {
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
        }
        store4(out_t + w.off[p], o);
      }
    }
  }
}

template <int POOL>
__global__ __launch_bounds__(TAIL_THREADS) void tail_tangent_kernel(TailArgs t) {
  __shared__ double red[TAIL_THREADS * 8];
  TAIL_SETUP(POOL)
  const float* zd_t = a.zd + (size_t)task * z_task;
  float m1[4], m2[4], gd[4];
  load4(a.m1 + (size_t)task * a.c + c0, m1);
  load4(a.m2 + (size_t)task * a.c + c0, m2);
  load4(a.gammad + (size_t)task * a.vstride + c0, gd);
  // ---- tangent of BatchNorm + ReLU + MaxPool (bn_tan_fwd_kernel's body)
  {
    float bd[4];
    load4(a.betad + (size_t)task * a.vstride + c0, bd);
    float* out_t = t.pooled + (size_t)task * p_task;
    Window<POOL> w;
    for (int win = wl; win < it.nwin; win += wpb) {
      w.locate(a, it, win, c0);
      if (!w.pooled) continue;
      floatx4 umax, zh_at, zd_at;
      scan_window<POOL, true>(w, z_t, zd_t, k, umax, zh_at, zd_at);
      floatx4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float zhd = k.r[c] * (zd_at[c] - m1[c] - zh_at[c] * m2[c]);
        const float ud = gd[c] * zh_at[c] + k.g[c] * zhd + bd[c];
        o[c] = (umax[c] > 0.f) ? ud : 0.f;
      }
      *reinterpret_cast<floatx4*>(out_t + w.poff) = o;
    }
  }
  __syncthreads();
  // ---- logit tangents and R{dlogits} (head_rows_kernel<true>'s body), then R{dWl}, R{dbl}, R{df} (head_grads_kernel<true>'s)
  const HeadArgs& h = t.hd;
  for (int n = tid >> 6; n < h.n; n += TAIL_THREADS / 64) head_row<true>(h, task, n, tid & 63);
  __syncthreads();
  tail_head_grads<true>(h, task, reinterpret_cast<float*>(red));
  __syncthreads();
  // ---- tangent BatchNorm backward sums: R{dbeta} = sum dud, R{dgamma} = sum (dud zh + du zhd) (bn_tan_bwd_reduce_kernel's terms)
  const float* dp_t = a.dp + (size_t)task * p_task;
  const float* dpd_t = a.dpd + (size_t)task * p_task;
  {
    double rg[4] = {0, 0, 0, 0}, rb[4] = {0, 0, 0, 0};
    Window<POOL> w;
    for (int win = wl; win < it.nwin; win += wpb) {
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
    tail_reduce_store(rg, rb, red, quads, a.c, t.sum0 + (size_t)task * t.sum_stride, t.sum1 + (size_t)task * t.sum_stride);
  }
  __syncthreads();
  // ---- tangent BatchNorm backward apply: R{dz} (bn_tan_bwd_apply_kernel's body)
  {
    float* out_t = a.out + (size_t)task * z_task;
    float dgm[4], dbm[4], rgm[4], rbm[4], c1[4], gr[4];
    load4(a.dgamma + (size_t)task * a.gstride + c0, dgm);
    load4(a.dbeta + (size_t)task * a.gstride + c0, dbm);
    load4(t.sum0 + (size_t)task * t.sum_stride + c0, rgm);
    load4(t.sum1 + (size_t)task * t.sum_stride + c0, rbm);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      dgm[c] *= a.inv_m; dbm[c] *= a.inv_m; rgm[c] *= a.inv_m; rbm[c] *= a.inv_m;
      const float rd = -k.r[c] * k.r[c] * m2[c];
      c1[c] = gd[c] * k.r[c] + k.g[c] * rd;
      gr[c] = k.g[c] * k.r[c];
    }
    Window<POOL> w;
    for (int win = wl; win < it.nwin; win += wpb) {
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
        }
        store4(out_t + w.off[p], o);
      }
    }
  }
}

// The stage bodies assume: 4 | c, (c / 4) | TAIL_THREADS, c <= TAIL_THREADS, and the staged dlogits + the two groups' fold buffers inside
// the reduction buffer (TAIL_THREADS * 8 doubles).
bool tail_supported(int n, int c, int ways) {
  if (c < 4 || c % 4 || TAIL_THREADS % (c / 4) || c > TAIL_THREADS || ways < 1 || ways > 64 || n < 1) return false;
  const size_t floats = (size_t)((2 * n * ways + 63) & ~63) + (size_t)(TAIL_THREADS / 256) * 3 * 8 * 64;
  return floats * sizeof(float) <= (size_t)TAIL_THREADS * 8 * sizeof(double);
}

hipError_t launch_tail(hipStream_t st, const TailArgs& t, int tasks, int pool, int tangent) {
  if (!tail_supported(t.bn.n, t.bn.c, t.hd.ways)) return hipErrorInvalidValue;
  if (tangent) {
    if (pool) hipLaunchKernelGGL(tail_tangent_kernel<1>, dim3(tasks), dim3(TAIL_THREADS), 0, st, t);
    else hipLaunchKernelGGL(tail_tangent_kernel<0>, dim3(tasks), dim3(TAIL_THREADS), 0, st, t);
  } else {
    if (pool) hipLaunchKernelGGL(tail_fwd_bwd_kernel<1>, dim3(tasks), dim3(TAIL_THREADS), 0, st, t);
    else hipLaunchKernelGGL(tail_fwd_bwd_kernel<0>, dim3(tasks), dim3(TAIL_THREADS), 0, st, t);
  }
  return hipGetLastError();
}

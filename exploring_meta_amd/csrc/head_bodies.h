// Device bodies of the classifier head (Linear(F, ways) + CrossEntropyLoss(mean) forward / backward / tangent), shared by the kernels of
// head.hip (one launch per stage over all tasks) and the per-task tail kernel of tail.hip (all stages of a task in one workgroup).
#pragma once
#include "mi_common.h"
#include "kernels.h"

// logits[n][w] = bl[w] + sum_i f[n][i] * wl[w][i]  (+ tangent terms), one (n,w) pair per wave iteration.
__device__ __forceinline__ float wave_dot(const float* __restrict__ x, const float* __restrict__ y, int len, int lane) {
  float s = 0.f;
  for (int i = lane; i < len; i += 64) s = fmaf(x[i], y[i], s);
  return wave_sum(s);
}

// (head.hip) The head runs as two launches so that a 32-task meta-batch fills the chip:
//   rows kernel   grid (T, ceil(N/4)): one wave per sample row -> `ways` dot products, softmax, prob / dlogits, row loss, row hit
//   grads kernel  grid (T, ceil(F/64)): 64 feature columns x 4 row groups per workgroup -> dWl[:, i], df[:, i]; chunk 0 also reduces loss, acc, dbl
// TANGENT = the R-operator version: rows compute ld = fd wl^T + f wld^T + bld and R{dl}; grads add the second products.
// One sample row by one wave (lane = 0..63): `ways` dot products over the feature row, softmax, prob / dlogits, row loss, row hit
// (TANGENT: ld = fd wl^T + f wld^T + bld and R{dl}).  Shared by head_rows_kernel (head.hip) and the per-task tail kernel (tail.hip):
// the same instructions in the same order, whoever calls it.
// f_n / fd_n: the row's features (and their tangents), wl_t / wld_t: the task's head weights (and the direction's) -- wherever they live
// (global memory in head.hip; LDS copies in tail.hip).
template <bool TANGENT>
// row_copy (optional, LDS): the row's dlogits (TANGENT: R{dlogits}) [WY], then -- primal only -- its loss and hit, for a caller that keeps working on them.
// bl_t / bld_t: the task's head bias (the direction's); y: the row's label (primal).
__device__ __forceinline__ void head_row_at(const HeadArgs& a, int task, int n, int lane, const float* f_n, const float* fd_n,
                                            const float* wl_t, const float* wld_t, const float* bl_t, const float* bld_t, int y,
                                            float* row_copy = nullptr) {
  const int N = a.n, F = a.feat, WY = a.ways;
  // lane w (< WY) ends up holding logit w of this row.  All `ways` dot products of the row advance together: one pass over the
  // feature row, 16-byte loads, every load of the pass independent of the others (the former one-dot-at-a-time loop with 4-byte
  // loads waited for memory once per 64 features and dot product: 57 us for a 5-way tangent row kernel).
  float mine = 0.f;
  const bool vec = (F % 4 == 0) && WY <= 8 &&
                   ((reinterpret_cast<uintptr_t>(f_n) | reinterpret_cast<uintptr_t>(wl_t) | reinterpret_cast<uintptr_t>(fd_n) |
                     reinterpret_cast<uintptr_t>(wld_t)) & 15) == 0;
  if (vec) {
    float part[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) part[w] = 0.f;
    // two 256-float steps per round with all of their loads issued before the first multiply-add (a 5-way tangent row has 24 16-byte
    // loads per round; one step at a time the wave waited for memory four times per 800-feature row); same accumulation order
    const floatx4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = lane * 4; i0 < F; i0 += 512) {
      floatx4 fv[2], fdv[2], wv[2][8], wdv[2][8];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = i0 + 256 * u;
        const bool ok = i < F;
        fv[u] = ok ? *reinterpret_cast<const floatx4*>(f_n + i) : z4;
        fdv[u] = (TANGENT && fd_n && ok) ? *reinterpret_cast<const floatx4*>(fd_n + i) : z4;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          wv[u][w] = z4; wdv[u][w] = z4;
          if (w < WY && ok) {
            if (!TANGENT || fd_n) wv[u][w] = *reinterpret_cast<const floatx4*>(wl_t + (size_t)w * F + i);
            if (TANGENT) wdv[u][w] = *reinterpret_cast<const floatx4*>(wld_t + (size_t)w * F + i);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (i0 + 256 * u >= F) continue;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          if (w < WY) {
            if (!TANGENT) {
              part[w] = fmaf(fv[u][0], wv[u][w][0], fmaf(fv[u][1], wv[u][w][1], fmaf(fv[u][2], wv[u][w][2], fmaf(fv[u][3], wv[u][w][3], part[w]))));
            } else {
              part[w] = fmaf(fv[u][0], wdv[u][w][0], fmaf(fv[u][1], wdv[u][w][1], fmaf(fv[u][2], wdv[u][w][2], fmaf(fv[u][3], wdv[u][w][3], part[w]))));
              if (fd_n)
                part[w] = fmaf(fdv[u][0], wv[u][w][0], fmaf(fdv[u][1], wv[u][w][1], fmaf(fdv[u][2], wv[u][w][2], fmaf(fdv[u][3], wv[u][w][3], part[w]))));
            }
          }
        }
      }
    }
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      if (w < WY) {
        const float d = wave_sum(part[w]) + (TANGENT ? bld_t[w] : bl_t[w]);
        if (lane == w) mine = d;
      }
    }
  } else {
    for (int w = 0; w < WY; ++w) {
      float d;
      if (!TANGENT) {
        d = wave_dot(f_n, wl_t + (size_t)w * F, F, lane) + bl_t[w];
      } else {
        d = wave_dot(f_n, wld_t + (size_t)w * F, F, lane) + bld_t[w];
        if (fd_n) d += wave_dot(fd_n, wl_t + (size_t)w * F, F, lane);
      }
      if (lane == w) mine = d;
    }
  }
  const size_t o = ((size_t)task * N + n) * WY;
  const bool act = lane < WY;
  const float invn = 1.f / (float)N;
  if (!TANGENT) {
    // softmax over lanes 0..WY-1 (first maximal index like torch.argmax)
    float mx = act ? mine : -INFINITY;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const unsigned long long eq = __ballot(act && mine == mx);
    const int am = __ffsll((long long)eq) - 1;
    const float ex = act ? expf(mine - mx) : 0.f;
    const float se = wave_sum(ex);
    const float ly = __shfl(mine, y, 64);
    if (act) {
      const float p = ex / se;
      const float dl = (p - (lane == y ? 1.f : 0.f)) * invn;
      if (a.prob) a.prob[o + lane] = p;
      if (a.dl) a.dl[o + lane] = dl;
      if (row_copy) row_copy[lane] = dl;
      if (a.logits) a.logits[o + lane] = mine;
    }
    if (lane == 0) {
      const float rl_ = (mx + logf(se)) - ly, rh_ = (am == y) ? 1.f : 0.f;
      a.rowloss[(size_t)task * N + n] = rl_;
      a.rowhit[(size_t)task * N + n] = rh_;
      if (row_copy) { row_copy[WY] = rl_; row_copy[WY + 1] = rh_; }
    }
  } else {
    if (a.ld_out && act) a.ld_out[o + lane] = mine;
    if (a.fixed_dl) {
      if (act) a.rdl[o + lane] = 0.f;
    } else {
      const float pr = act ? a.prob[o + lane] : 0.f;
      const float dot = wave_sum(pr * mine);
      if (act) {
        const float r_ = pr * (mine - dot) * invn;
        a.rdl[o + lane] = r_;
        if (row_copy) row_copy[lane] = r_;
      }
    }
  }
}


template <bool TANGENT>
__device__ __forceinline__ void head_row(const HeadArgs& a, int task, int n, int lane) {
  const int N = a.n, F = a.feat;
  head_row_at<TANGENT>(a, task, n, lane, a.f + ((size_t)task * N + n) * F,
                       (TANGENT && a.fd) ? a.fd + ((size_t)task * N + n) * F : nullptr, a.wl + (size_t)task * a.pstride,
                       TANGENT ? a.wld + (size_t)task * a.vstride : nullptr, a.bl + (size_t)task * a.pstride,
                       TANGENT ? a.bld + (size_t)task * a.vstride : nullptr, TANGENT ? 0 : a.y[(size_t)task * N + n]);
}

// dl (primal) or R{dl} (+ dl, tangent) of one task into LDS: s_a [N][WY], s_b [N][WY] (tangent only).  The caller synchronises.
template <bool TANGENT>
__device__ __forceinline__ void head_stage_dl(const HeadArgs& a, int task, int tid, int nthreads, float* s_a, float* s_b) {
  const int N = a.n, WY = a.ways;
  const float* src_a = (TANGENT ? a.rdl : a.dl) + (size_t)task * N * WY;
  for (int e = tid; e < N * WY; e += nthreads) {
    s_a[e] = src_a[e];
    if (TANGENT) s_b[e] = a.dl[(size_t)task * N * WY + e];
  }
}

// 64 feature columns (chunk) by a group of 256 threads (tid = 0..255 inside the group): dWl[:, i] and df[:, i].  s_red: [3][8][64] floats
// of the group's own; contains ONE workgroup barrier (every thread of the workgroup must pass through the same number of calls).
template <bool TANGENT>
__device__ __forceinline__ void head_grads_chunk(const HeadArgs& a, int task, int chunk, int tid, const float* s_a, const float* s_b, float* s_red) {
  const int N = a.n, F = a.feat, WY = a.ways;
  const float* f_t = a.f + (size_t)task * N * F;
  const float* fd_t = (TANGENT && a.fd) ? a.fd + (size_t)task * N * F : nullptr;
  const float* wl_t = a.wl + (size_t)task * a.pstride;
  const float* wld_t = TANGENT ? a.wld + (size_t)task * a.vstride : nullptr;
  float* dwl_t = a.dwl + (size_t)task * a.gstride;
  // 64 feature columns per workgroup, the N rows dealt to 4 thread groups (rows rg, rg+4, ...): every thread has at most
  // ceil(N/4) rows' features in flight at once (one round trip to memory instead of N/8), partial column sums fold through LDS
  // in group order, and each group writes df for its own rows.  (One thread per column walking all N rows took 18-24 us whatever
  // the task count: 4 sequential rounds of loads and N sequential stores.)
  const int rg = tid >> 6;
  const int i = chunk * 64 + (tid & 63);
  if (WY <= 8) {
    float dw[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) dw[w] = 0.f;
    constexpr int RMAX = 8;                           // rows per thread and round
    if (i < F) {
      for (int n0 = rg; n0 < N; n0 += 4 * RMAX) {
        float fv[RMAX], fdv[RMAX];
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
          const int n = n0 + 4 * u;
          const bool ok = n < N;
          fv[u] = ok ? f_t[(size_t)n * F + i] : 0.f;
          fdv[u] = (TANGENT && fd_t && ok) ? fd_t[(size_t)n * F + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
          const int n = n0 + 4 * u;
          if (n >= N) break;
#pragma unroll
          for (int w = 0; w < 8; ++w) {
            if (w < WY) {
              dw[w] = fmaf(s_a[n * WY + w], fv[u], dw[w]);
              if (TANGENT && fd_t) dw[w] = fmaf(s_b[n * WY + w], fdv[u], dw[w]);
            }
          }
        }
      }
    }
    if (rg > 0) {
#pragma unroll
      for (int w = 0; w < 8; ++w) s_red[((rg - 1) * 8 + w) * 64 + (tid & 63)] = dw[w];
    }
    __syncthreads();
    if (rg == 0 && i < F) {
#pragma unroll
      for (int w = 0; w < 8; ++w)
        if (w < WY) dwl_t[(size_t)w * F + i] = ((dw[w] + s_red[(0 * 8 + w) * 64 + tid]) + s_red[(1 * 8 + w) * 64 + tid]) + s_red[(2 * 8 + w) * 64 + tid];
    }
    if (a.df && i < F) {                              // df[n][i] = sum_w a[n][w] wl[w][i] (+ b[n][w] wld[w][i]): weights in registers
      float wv[8], wdv[8];
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        wv[w] = w < WY ? wl_t[(size_t)w * F + i] : 0.f;
        wdv[w] = (TANGENT && w < WY) ? wld_t[(size_t)w * F + i] : 0.f;
      }
      float* df_t = a.df + (size_t)task * N * F;
      for (int n = rg; n < N; n += 4) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          if (w < WY) {
            s = fmaf(s_a[n * WY + w], wv[w], s);
            if (TANGENT) s = fmaf(s_b[n * WY + w], wdv[w], s);
          }
        }
        df_t[(size_t)n * F + i] = s;
      }
    }
  } else if (i < F && rg == 0) {
    for (int w = 0; w < WY; ++w) {                 // dwl[w][i] = sum_n a[n][w] f[n][i] (+ b[n][w] fd[n][i])
      float s = 0.f;
      for (int n = 0; n < N; ++n) {
        s = fmaf(s_a[n * WY + w], f_t[(size_t)n * F + i], s);
        if (TANGENT && fd_t) s = fmaf(s_b[n * WY + w], fd_t[(size_t)n * F + i], s);
      }
      dwl_t[(size_t)w * F + i] = s;
    }
    if (a.df) {                                    // df[n][i] = sum_w a[n][w] wl[w][i] (+ b[n][w] wld[w][i])
      float* df_t = a.df + (size_t)task * N * F;
      for (int n = 0; n < N; ++n) {
        float s = 0.f;
        for (int w = 0; w < WY; ++w) {
          s = fmaf(s_a[n * WY + w], wl_t[(size_t)w * F + i], s);
          if (TANGENT) s = fmaf(s_b[n * WY + w], wld_t[(size_t)w * F + i], s);
        }
        df_t[(size_t)n * F + i] = s;
      }
    }
  }
}

// dbl, and loss[t] / acc[t] from the row losses / hits (fixed order): once per task, by the threads tid = 0..nthreads-1 (nthreads > 64).
template <bool TANGENT>
__device__ __forceinline__ void head_task_sums(const HeadArgs& a, int task, int tid, int nthreads, const float* s_a) {
  const int N = a.n, WY = a.ways;
  float* dbl_t = a.dbl + (size_t)task * a.gstride;
  for (int w = tid; w < WY; w += nthreads) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += s_a[n * WY + w];
    dbl_t[w] = s;
  }
  // loss[t] = mean_n rowloss, acc[t] = mean_n rowhit (fixed order) -- the job of head_reduce_kernel, folded in here when a
  // gradient launch follows the rows launch anyway
  if (!TANGENT && a.loss && tid == 64) {
    float ls = 0.f, cs = 0.f;
    for (int k = 0; k < N; ++k) { ls += a.rowloss[(size_t)task * N + k]; cs += a.rowhit[(size_t)task * N + k]; }
    a.loss[task] = ls / (float)N;
    a.acc[task] = cs / (float)N;
  }
}

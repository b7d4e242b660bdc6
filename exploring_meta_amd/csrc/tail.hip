// The tail of a pass in ONE launch: the LAST ConvBlock's BatchNorm + ReLU + MaxPool, the classifier head with its cross-entropy and accuracy,
// the head's backward and the last block's BatchNorm-backward sums -- in the Hessian-vector passes their tangents.
//
// Replaces (reference): the last `normalize -> relu -> max_pool` of ConvBase (core_functions/vision_models.py:188-193), `self.linear(x.view(-1,
// 25*hidden))` (:109), `loss(learner(adapt_data), adapt_labels)` / `accuracy` (core_functions/vision.py:11,16-18,21-23) and the autograd backward /
// double-backward of those ops -- four launches per pass before (bn_fwd, head_rows, head_grads, bn_bwd_reduce; tangent: bn_tan_fwd, head_rows<T>,
// head_grads<T>, bn_tan_bwd_reduce), each a chain of dependent memory round trips on a few hundred kilobytes.
//
// Shape: FOUR workgroups per task, workgroup rg owning the sample rows n = rg, rg + 4, ... -- the row groups head_grads_kernel already sums over.
// A workgroup reads its rows' conv outputs ONCE and keeps everything it derives on chip: the pooled features in LDS (the head's dot products, the
// weight-gradient partial and the feature cotangents read them there), zhat at the pooling argmax and the ReLU mask in registers (the BatchNorm-backward
// terms of a pooled element are formed by the thread that pooled it, from the cotangent it has just computed).  What crosses workgroups is sums only
// -- dWl over rows, dbl, loss, accuracy, dgamma / dbeta over images -- and they cross the way every sum in this library does: one partial per
// workgroup, written through (sc1), drained, counted; the workgroup that arrives last folds them in a fixed order (finalize.h).  Nobody WAITS for
// another workgroup: no in-kernel barrier (which costs what a kernel boundary costs on this chip, MI355X_MICROARCH.md price list), no assumption about
// dispatch order, residency or placement.  The BatchNorm-backward APPLY needs the folded sums and stays the next launch.
//
// (Round 6 first built the whole tail, apply included, as ONE workgroup per task: bit-identical and 27 % SLOWER at one task per call, 4 % at 32 -- a
// single CU pulls a task's 320 KB through five dependent stages at ~100 GB/s.  profiles/r6/ab_fuse_last_v1.txt.)
//
// Arithmetic: the stage bodies are the separate kernels' (bn_window.h, head_bodies.h) or the same operations in the same order -- p, logits, prob,
// dlogits, loss, accuracy, dWl (the four row-group partials are folded ((g0 + g1) + g2) + g3 as head_grads_kernel folds them), dbl, df are
// bit-identical to the separate launches.  dgamma / dbeta are the same fp64 terms in a fixed order that does not depend on the tasks per call (the
// separate kernels fold per-workgroup partials whose number does): equal in fp32 except where the fp64 sum lies within ~1e-16 of a rounding boundary.
#include "mi_common.h"
#include "kernels.h"
#include "bn_window.h"
#include "head_bodies.h"

#define TAIL_THREADS 512
#define TAIL_GROUPS 4          // row groups = workgroups per task (head_grads_kernel's grouping: rows rg, rg + 4, ...)
#define TAIL_KMAX 4            // (row, pooling window, channel quad) items per thread

__device__ __forceinline__ float tail_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double tail_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tail_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tail_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Write-through (sc1, aux = 16) 16-byte stores and cache-bypassing loads on a raw buffer: plain intrinsics (the scheduler may batch them freely, unlike
// a sequence of atomics), full lines per wave instruction (a 4-byte sc1 store per lane costs ~6x the time per byte of a 16-byte one, MI355X_MICROARCH.md).
__device__ __forceinline__ mi_rsrc tail_rsrc(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void tail_st16(mi_rsrc r, unsigned byte_off, floatx4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(mi_u32x4, v), r, byte_off, 0, 16);
}
__device__ __forceinline__ floatx4 tail_ld16(mi_rsrc r, unsigned byte_off) {
  return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}
__device__ __forceinline__ float tail_ld4(mi_rsrc r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 16));
}

// global -> LDS copies with every load of a trip issued before the first store waits for one (a plain copy loop is a chain of round trips:
// load, wait, store -- eight of them for the head's weights)
__device__ __forceinline__ void tail_copy(float* dst, const float* src, int count) {
  const int tid = threadIdx.x;
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (count & 3) == 0) {
    const int n4 = count >> 2;
    for (int e0 = tid; e0 < n4; e0 += 4 * TAIL_THREADS) {
      floatx4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int e = e0 + u * TAIL_THREADS; v[u] = reinterpret_cast<const floatx4*>(src)[e < n4 ? e : 0]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int e = e0 + u * TAIL_THREADS; if (e < n4) reinterpret_cast<floatx4*>(dst)[e] = v[u]; }
    }
  } else {
    for (int e0 = tid; e0 < count; e0 += 8 * TAIL_THREADS) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + u * TAIL_THREADS; v[u] = src[e < count ? e : 0]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + u * TAIL_THREADS; if (e < count) dst[e] = v[u]; }
    }
  }
}
// rows rg, rg + 4, ... of src [N][F] -> dst [rows][F]
__device__ __forceinline__ void tail_copy_rows(float* dst, const float* src, int rows, int F, int rg) {
  const int tid = threadIdx.x;
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (F & 3) == 0) {
    const int f4 = F >> 2, n4 = rows * f4;
    for (int e0 = tid; e0 < n4; e0 += 4 * TAIL_THREADS) {
      floatx4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * TAIL_THREADS, ec = e < n4 ? e : 0, rl = ec / f4;
        v[u] = reinterpret_cast<const floatx4*>(src + (size_t)(rg + TAIL_GROUPS * rl) * F)[ec - rl * f4];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int e = e0 + u * TAIL_THREADS; if (e < n4) reinterpret_cast<floatx4*>(dst)[e] = v[u]; }
    }
  } else {
    for (int e = tid; e < rows * F; e += TAIL_THREADS) { const int rl = e / F; dst[e] = src[(size_t)(rg + TAIL_GROUPS * rl) * F + (e - rl * F)]; }
  }
}

struct TailLds {
  double* red;       // [TAIL_THREADS * 8]
  float *f, *fd;     // [RL][F] pooled features of this workgroup's rows (tangent kernel: the stored primal features, and their tangents)
  float *wl, *wld;   // [WY][F] the task's head weights (and the direction's)
  float *a, *b;      // [RL][WY + 2] dlogits, row loss, row hit (tangent kernel: a = R{dlogits}, b = dlogits)
  float *bl, *bld;   // [WY] the head bias (and the direction's)
  int* y;            // [RL] labels of this workgroup's rows (primal)
  int* flag;
};
__host__ __device__ inline size_t tail_lds_bytes(int n, int feat, int ways, int tangent) {
  const size_t rl = (size_t)(n + TAIL_GROUPS - 1) / TAIL_GROUPS;
  size_t floats = (tangent ? 2 : 1) * (rl * feat + (size_t)ways * feat) + 2 * ((rl * (ways + 2) + 3) & ~(size_t)3) + 2 * ((ways + 3) & ~(size_t)3) +
                  ((rl + 3) & ~(size_t)3) + 4;
  return (size_t)TAIL_THREADS * 8 * sizeof(double) + floats * sizeof(float);
}
__device__ __forceinline__ TailLds tail_carve(double* smem, int RL, int F, int WY, bool tangent) {
  TailLds L;
  L.red = smem;
  float* p = reinterpret_cast<float*>(smem + TAIL_THREADS * 8);
  L.f = p; p += (size_t)RL * F;
  L.fd = tangent ? p : nullptr; if (tangent) p += (size_t)RL * F;
  L.wl = p; p += (size_t)WY * F;
  L.wld = tangent ? p : nullptr; if (tangent) p += (size_t)WY * F;
  L.a = p; p += (RL * (WY + 2) + 3) & ~3;
  L.b = p; p += (RL * (WY + 2) + 3) & ~3;
  L.bl = p; p += (WY + 3) & ~3;
  L.bld = p; p += (WY + 3) & ~3;
  L.y = reinterpret_cast<int*>(p); p += (RL + 3) & ~3;
  L.flag = reinterpret_cast<int*>(p);
  return L;
}

// the workgroup's 8 per-thread fp64 accumulators (4 channels x 2 quantities) -> one partial [2][c] of the task, written through
__device__ __forceinline__ void tail_partial(const double* acc0, const double* acc1, double* red, int quads, int c, double* part) {
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
    tail_st(part + tid, s0);
    tail_st(part + c + tid, s1);
  }
}

// finalize.h's arrival: every thread of the workgroup calls it after the workgroup's write-through stores; true in the workgroup that arrived last
__device__ __forceinline__ bool tail_arrive(unsigned* counter, int task, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(counter + task, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (prev + 1u == (unsigned)TAIL_GROUPS);
    if (last) __hip_atomic_store(counter + task, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = last;
  }
  __syncthreads();
  const bool last = *flag != 0;
  if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // (compiler ordering: the fold's sc1 loads stay below the counter read)
  return last;
}

// per-row scalars that the folding workgroup sums in row order: scr[task][rg][rl][WY + 2] = (WY values, row loss, row hit)
__device__ __forceinline__ float* tail_scr_row(const TailArgs& t, int task, int n, int RL, int WY) {
  return t.scr + (((size_t)task * TAIL_GROUPS + (n % TAIL_GROUPS)) * RL + n / TAIL_GROUPS) * (WY + 2);
}

// The folding workgroup (the one that arrived last): per-row scalars -> loss, accuracy (primal), dbl; the four row-group partials -> dWl
// (((g0 + g1) + g2) + g3, head_grads_kernel's fold) and dgamma / dbeta.  All loads are write-through loads that miss every cache (~1 us each):
// they are issued in batches, never one per dependent step.  `buf`: TAIL_THREADS * 8 doubles of LDS the kernel no longer needs.
template <bool TANGENT>
__device__ __forceinline__ void tail_fold(const TailArgs& t, int task, int RL, bool bwd, double* buf) {
  const BnArgs& a = t.bn;
  const HeadArgs& h = t.hd;
  const int tid = threadIdx.x, N = h.n, F = h.feat, WY = h.ways, WS = WY + 2;
  float* rowv = reinterpret_cast<float*>(buf);               // [N][WS]
  const int WF = WY * F, q4n = WF >> 2;                      // (4 | F: tail_supported)
  const mi_rsrc rscr = tail_rsrc(t.scr + (size_t)task * TAIL_GROUPS * RL * WS, (size_t)TAIL_GROUPS * RL * WS * sizeof(float));
  const mi_rsrc rwp = tail_rsrc(t.wpart + (size_t)task * TAIL_GROUPS * WF, (size_t)TAIL_GROUPS * WF * sizeof(float));
  // every load of the fold is requested here, before anything waits
  float rv[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + u * TAIL_THREADS, n = e / WS, w = e - n * WS;
    const bool need = e < N * WS && !(TANGENT && w >= WY);
    rv[u] = need ? tail_ld4(rscr, (unsigned)((((n % TAIL_GROUPS) * RL + n / TAIL_GROUPS) * WS + w) * sizeof(float))) : 0.f;
  }
  floatx4 pv[2][TAIL_GROUPS];
  double bv0[TAIL_GROUPS], bv1[TAIL_GROUPS];
  const int ch = tid - (TAIL_THREADS - a.c);
  if (bwd) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = tid + u * TAIL_THREADS;
#pragma unroll
      for (int r = 0; r < TAIL_GROUPS; ++r) pv[u][r] = tail_ld16(rwp, (unsigned)(((size_t)r * WF + 4 * (size_t)(q < q4n ? q : 0)) * sizeof(float)));
    }
#pragma unroll
    for (int r = 0; r < TAIL_GROUPS; ++r) {
      const int cc = ch >= 0 ? ch : 0;
      // (8-byte agent-scope loads as in finalize.h's fold: the form the statistics partials have always been read with)
      const double* bp = t.bpart + (size_t)task * TAIL_GROUPS * 2 * a.c + (size_t)r * 2 * a.c;
      bv0[r] = tail_ld(bp + cc);
      bv1[r] = tail_ld(bp + a.c + cc);
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + u * TAIL_THREADS;
    if (e < N * WS) rowv[e] = rv[u];
  }
  __syncthreads();
  if (!TANGENT && tid == 64 && h.loss) {
    float ls = 0.f, cs = 0.f;
    for (int n = 0; n < N; ++n) { ls += rowv[n * WS + WY]; cs += rowv[n * WS + WY + 1]; }
    h.loss[task] = ls / (float)N;
    h.acc[task] = cs / (float)N;
  }
  if (!bwd) return;
  if (tid < WY) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += rowv[n * WS + tid];
    h.dbl[(size_t)task * h.gstride + tid] = s;
  }
  {
    float* dwl_t = h.dwl + (size_t)task * h.gstride;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = tid + u * TAIL_THREADS;
      if (q < q4n) {
        floatx4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = ((pv[u][0][c] + pv[u][1][c]) + pv[u][2][c]) + pv[u][3][c];
        *reinterpret_cast<floatx4*>(dwl_t + 4 * (size_t)q) = o;
      }
    }
    for (int q = tid + 2 * TAIL_THREADS; q < q4n; q += TAIL_THREADS) {       // (more than 4096 head weights per task: the rest, one quad at a time)
      floatx4 v[TAIL_GROUPS], o;
#pragma unroll
      for (int r = 0; r < TAIL_GROUPS; ++r) v[r] = tail_ld16(rwp, (unsigned)(((size_t)r * WF + 4 * (size_t)q) * sizeof(float)));
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = ((v[0][c] + v[1][c]) + v[2][c]) + v[3][c];
      *reinterpret_cast<floatx4*>(dwl_t + 4 * (size_t)q) = o;
    }
  }
  if (ch >= 0) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int r = 0; r < TAIL_GROUPS; ++r) { s0 += bv0[r]; s1 += bv1[r]; }
    t.sum0[(size_t)task * t.sum_stride + ch] = (float)s0;
    t.sum1[(size_t)task * t.sum_stride + ch] = (float)s1;
  }
}

// debug stamps (TailArgs::stamps): the constant 100 MHz clock at a stage boundary, thread 0 of the workgroup
#define TAIL_STAMP(i)                                                                             \
  do {                                                                                            \
    if (t.stamps && threadIdx.x == 0) t.stamps[((size_t)blockIdx.y * TAIL_GROUPS + blockIdx.x) * 16 + (i)] = wall_clock64(); \
  } while (0)

#define TAIL_SETUP(POOL, TANGENT)                                                                 \
  extern __shared__ double tail_smem[];                                                           \
  const BnArgs& a = t.bn;                                                                         \
  const HeadArgs& h = t.hd;                                                                       \
  const int tid = threadIdx.x, rg = blockIdx.x, task = blockIdx.y;                                \
  const int N = h.n, F = h.feat, WY = h.ways;                                                     \
  const int RL = (N + TAIL_GROUPS - 1) / TAIL_GROUPS;                                             \
  const int rows = N > rg ? (N - rg + TAIL_GROUPS - 1) / TAIL_GROUPS : 0;                         \
  const TailLds L = tail_carve(tail_smem, RL, F, WY, TANGENT);                                    \
  const int quads = a.c >> 2, quad = tid % quads, c0 = quad * 4;                                  \
  const WinIter<POOL> it(a);                                                                      \
  const int wins = it.hw2 * it.ww2, per_row = wins * quads, items = rows * per_row;               \
  const size_t z_task = (size_t)a.n * a.ho * a.wo * a.c;                                          \
  const size_t p_task = (size_t)a.n * it.hp * it.wp * a.c;                                        \
  const ChanConst k = load_consts(a, task, c0);                                                   \
  const float* z_t = a.z + (size_t)task * z_task;

template <int POOL>
__global__ __launch_bounds__(TAIL_THREADS) void tail_fwd_bwd_kernel(TailArgs t) {
  TAIL_SETUP(POOL, false)
  // the task's head weights into LDS (requested first: in flight under the BatchNorm stage)
  TAIL_STAMP(0);
  // the task's head weights, bias and this workgroup's labels: requested now, stored to LDS after the BatchNorm stage's loads are on their way
  const float* wl_t = h.wl + (size_t)task * h.pstride;
  const int wq = (WY * F) >> 2;
  const bool wl_vec = (reinterpret_cast<uintptr_t>(wl_t) & 15) == 0 && wq <= 4 * TAIL_THREADS;
  floatx4 wlv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { const int e = tid + u * TAIL_THREADS; wlv[u] = (wl_vec && e < wq) ? reinterpret_cast<const floatx4*>(wl_t)[e] : floatx4{0.f, 0.f, 0.f, 0.f}; }
  const float bias_v = tid < WY ? h.bl[(size_t)task * h.pstride + tid] : 0.f;
  const int y_v = (tid >= 64 && tid - 64 < rows) ? h.y[(size_t)task * N + rg + TAIL_GROUPS * (tid - 64)] : 0;
  TAIL_STAMP(1);
  // ---- BatchNorm + ReLU + MaxPool of this workgroup's rows (bn_fwd_kernel's arithmetic): p -> memory and LDS, zhat at the argmax / "ReLU on" kept
  float zh_keep[TAIL_KMAX][4];
  unsigned on_keep[TAIL_KMAX];
  int poff_keep[TAIL_KMAX];             // offset of the item's channel quad inside its row's features, -1: no item
  {
    float* out_t = t.pooled + (size_t)task * p_task;
    constexpr int NP = Window<POOL>::NP;
    floatx4 zv[TAIL_KMAX][NP];
    size_t poff[TAIL_KMAX];
    // every item's loads first (an item that does not exist reads the task's first window: a valid address, never used)
#pragma unroll
    for (int kk = 0; kk < TAIL_KMAX; ++kk) {
      const int item = tid + kk * TAIL_THREADS;
      const bool ok = item < items;
      const int rl = ok ? item / per_row : 0, n = rg + TAIL_GROUPS * rl;
      Window<POOL> w;
      w.locate(a, it, ok ? n * wins + (item - rl * per_row) / quads : 0, c0);
      const bool use = ok && w.pooled;
#pragma unroll
      for (int p = 0; p < NP; ++p) zv[kk][p] = *reinterpret_cast<const floatx4*>(z_t + (use ? w.off[p] : (size_t)c0));
      poff[kk] = w.poff;
      poff_keep[kk] = use ? (int)(w.poff - (size_t)n * F) : -1;
    }
    if (wl_vec) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int e = tid + u * TAIL_THREADS; if (e < wq) reinterpret_cast<floatx4*>(L.wl)[e] = wlv[u]; }
    } else {
      tail_copy(L.wl, wl_t, WY * F);
    }
    if (tid < WY) L.bl[tid] = bias_v;
    if (tid >= 64 && tid - 64 < rows) L.y[tid - 64] = y_v;
#pragma unroll
    for (int kk = 0; kk < TAIL_KMAX; ++kk) {
      on_keep[kk] = 0u;
#pragma unroll
      for (int c = 0; c < 4; ++c) zh_keep[kk][c] = 0.f;
      if (poff_keep[kk] < 0) continue;
      const int rl = (tid + kk * TAIL_THREADS) / per_row;
      floatx4 umax, zh_at, zd_at;
      scan_values<POOL, false>(zv[kk], nullptr, k, umax, zh_at, zd_at);
      floatx4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        o[c] = fmaxf(umax[c], 0.f);
        zh_keep[kk][c] = zh_at[c];
        on_keep[kk] |= (umax[c] > 0.f ? 1u : 0u) << c;
      }
      *reinterpret_cast<floatx4*>(out_t + poff[kk]) = o;
      *reinterpret_cast<floatx4*>(L.f + (size_t)rl * F + poff_keep[kk]) = o;
    }
  }
  __syncthreads();
  TAIL_STAMP(2);
  // ---- Linear + cross-entropy, one wave per row (head_rows_kernel's body on the LDS copies)
  const int WS = WY + 2;                                     // a row's dlogits, loss, hit in LDS
  for (int rl = tid >> 6; rl < rows; rl += TAIL_THREADS / 64)
    head_row_at<false>(h, task, rg + TAIL_GROUPS * rl, tid & 63, L.f + (size_t)rl * F, nullptr, L.wl, nullptr, L.bl, nullptr, L.y[rl], L.a + rl * WS);
  __syncthreads();
  TAIL_STAMP(3);
  // ... and, written through, where the folding workgroup sums them in row order
  const bool bwd = t.with_grad && task < t.bwd_tasks;
  for (int e = tid; e < rows * WS; e += TAIL_THREADS) {
    const int rl = e / WS;
    tail_st(tail_scr_row(t, task, rg + TAIL_GROUPS * rl, RL, WY) + (e - rl * WS), L.a[e]);
  }
  TAIL_STAMP(4);
  if (bwd) {
    // ---- this row group's share of dWl (head_grads_kernel: rows rg, rg + 4, ... in order) -> partial, written through (a column quad per thread)
    {
      const mi_rsrc rwp = tail_rsrc(t.wpart + ((size_t)task * TAIL_GROUPS + rg) * WY * F, (size_t)WY * F * sizeof(float));
      for (int q = tid; q < (F >> 2); q += TAIL_THREADS) {
        floatx4 dw[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) dw[w] = floatx4{0.f, 0.f, 0.f, 0.f};
        for (int rl = 0; rl < rows; ++rl) {
          const floatx4 fv = *reinterpret_cast<const floatx4*>(L.f + (size_t)rl * F + 4 * q);
#pragma unroll
          for (int w = 0; w < 8; ++w)
            if (w < WY) {
              const float dlw = L.a[rl * WS + w];
#pragma unroll
              for (int c = 0; c < 4; ++c) dw[w][c] = fmaf(dlw, fv[c], dw[w][c]);
            }
        }
#pragma unroll
        for (int w = 0; w < 8; ++w)
          if (w < WY) tail_st16(rwp, (unsigned)(((size_t)w * F + 4 * (size_t)q) * sizeof(float)), dw[w]);
      }
    }
    TAIL_STAMP(5);
    // ---- df of this workgroup's rows (= the cotangent of p) and, from it, the BatchNorm-backward terms of the elements this thread pooled
    double dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
    float* df_t = h.df + (size_t)task * N * F;
#pragma unroll
    for (int kk = 0; kk < TAIL_KMAX; ++kk) {
      if (poff_keep[kk] < 0) continue;
      const int item = tid + kk * TAIL_THREADS, rl = item / per_row, n = rg + TAIL_GROUPS * rl, fo = poff_keep[kk];
      floatx4 s = {0.f, 0.f, 0.f, 0.f};
      for (int w = 0; w < WY; ++w) {
        const floatx4 wv = *reinterpret_cast<const floatx4*>(L.wl + (size_t)w * F + fo);
        const float dlw = L.a[rl * WS + w];
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] = fmaf(dlw, wv[c], s[c]);
      }
      *reinterpret_cast<floatx4*>(df_t + (size_t)n * F + fo) = s;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float du = ((on_keep[kk] >> c) & 1u) ? s[c] : 0.f;
        db[c] += (double)du;
        dg[c] += (double)du * (double)zh_keep[kk][c];
      }
    }
    TAIL_STAMP(6);
    tail_partial(dg, db, L.red, quads, a.c, t.bpart + ((size_t)task * TAIL_GROUPS + rg) * 2 * a.c);
  }
  TAIL_STAMP(7);
  const bool last_wg = tail_arrive(t.counter, task, L.flag);
  TAIL_STAMP(8);
  if (!last_wg) return;
  // ---- the workgroup that arrived last: loss, accuracy; dbl; dWl from the four partials; dgamma / dbeta
  tail_fold<false>(t, task, RL, bwd, L.red);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TAIL_STAMP(9);
}

template <int POOL>
__global__ __launch_bounds__(TAIL_THREADS) void tail_tangent_kernel(TailArgs t) {
  TAIL_SETUP(POOL, true)
  TAIL_STAMP(0);
  // the task's head weights and bias, the direction's, and the primal features / dlogits of this workgroup's rows (stored by the primal pass): requested
  // now, stored to LDS once the BatchNorm stage's first loads are on their way
  const float* wl_t = h.wl + (size_t)task * h.pstride;
  const float* wld_t = h.wld + (size_t)task * h.vstride;
  const float* f_t = h.f + (size_t)task * N * F;
  const int wq = (WY * F) >> 2, f4 = F >> 2, fq = rows * f4;
  const bool pre_vec = ((reinterpret_cast<uintptr_t>(wl_t) | reinterpret_cast<uintptr_t>(wld_t) | reinterpret_cast<uintptr_t>(f_t)) & 15) == 0 &&
                       wq <= 2 * TAIL_THREADS && fq <= 4 * TAIL_THREADS;
  floatx4 wlv[2], wldv[2], fpv[4];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + u * TAIL_THREADS;
    const bool ok = pre_vec && e < wq;
    wlv[u] = ok ? reinterpret_cast<const floatx4*>(wl_t)[e] : floatx4{0.f, 0.f, 0.f, 0.f};
    wldv[u] = ok ? reinterpret_cast<const floatx4*>(wld_t)[e] : floatx4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + u * TAIL_THREADS, ec = (pre_vec && e < fq) ? e : 0, rl = ec / f4;
    fpv[u] = (pre_vec && e < fq) ? reinterpret_cast<const floatx4*>(f_t + (size_t)(rg + TAIL_GROUPS * rl) * F)[ec - rl * f4] : floatx4{0.f, 0.f, 0.f, 0.f};
  }
  const float bias_v = tid < WY ? h.bl[(size_t)task * h.pstride + tid] : 0.f;
  const float biasd_v = tid < WY ? h.bld[(size_t)task * h.vstride + tid] : 0.f;
  TAIL_STAMP(1);
  const float* zd_t = a.zd + (size_t)task * z_task;
  float m1[4], m2[4], gd[4], bd[4];
  load4(a.m1 + (size_t)task * a.c + c0, m1);
  load4(a.m2 + (size_t)task * a.c + c0, m2);
  load4(a.gammad + (size_t)task * a.vstride + c0, gd);
  load4(a.betad + (size_t)task * a.vstride + c0, bd);
  // ---- tangent of BatchNorm + ReLU + MaxPool (bn_tan_fwd_kernel's arithmetic): pd -> memory and LDS; zhat, zhat-dot at the argmax and "ReLU on" kept
  float zh_keep[TAIL_KMAX][4], zhd_keep[TAIL_KMAX][4];
  unsigned on_keep[TAIL_KMAX];
  int poff_keep[TAIL_KMAX];
  {
    float* out_t = t.pooled + (size_t)task * p_task;
    constexpr int NP = Window<POOL>::NP;
    // two items' loads (z and z-dot at every window position) in flight at a time
#pragma unroll
    for (int k0 = 0; k0 < TAIL_KMAX; k0 += 2) {
      floatx4 zv[2][NP], zdv[2][NP];
      size_t poff[2];
      if (k0 == 2) {            // (the staged operands: their loads were requested before the first pair's)
        if (pre_vec) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int e = tid + u * TAIL_THREADS;
            if (e < wq) { reinterpret_cast<floatx4*>(L.wl)[e] = wlv[u]; reinterpret_cast<floatx4*>(L.wld)[e] = wldv[u]; }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) { const int e = tid + u * TAIL_THREADS; if (e < fq) reinterpret_cast<floatx4*>(L.f)[e] = fpv[u]; }
        } else {
          tail_copy(L.wl, wl_t, WY * F);
          tail_copy(L.wld, wld_t, WY * F);
          tail_copy_rows(L.f, f_t, rows, F, rg);
        }
        if (tid < WY) { L.bl[tid] = bias_v; L.bld[tid] = biasd_v; }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kk = k0 + j, item = tid + kk * TAIL_THREADS;
        const bool ok = item < items;
        const int rl = ok ? item / per_row : 0, n = rg + TAIL_GROUPS * rl;
        Window<POOL> w;
        w.locate(a, it, ok ? n * wins + (item - rl * per_row) / quads : 0, c0);
        const bool use = ok && w.pooled;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          zv[j][p] = *reinterpret_cast<const floatx4*>(z_t + (use ? w.off[p] : (size_t)c0));
          zdv[j][p] = *reinterpret_cast<const floatx4*>(zd_t + (use ? w.off[p] : (size_t)c0));
        }
        poff[j] = w.poff;
        poff_keep[kk] = use ? (int)(w.poff - (size_t)n * F) : -1;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kk = k0 + j;
        on_keep[kk] = 0u;
#pragma unroll
        for (int c = 0; c < 4; ++c) { zh_keep[kk][c] = 0.f; zhd_keep[kk][c] = 0.f; }
        if (poff_keep[kk] < 0) continue;
        const int rl = (tid + kk * TAIL_THREADS) / per_row;
        floatx4 umax, zh_at, zd_at;
        scan_values<POOL, true>(zv[j], zdv[j], k, umax, zh_at, zd_at);
        floatx4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float zhd = k.r[c] * (zd_at[c] - m1[c] - zh_at[c] * m2[c]);
          const float ud = gd[c] * zh_at[c] + k.g[c] * zhd + bd[c];
          o[c] = (umax[c] > 0.f) ? ud : 0.f;
          zh_keep[kk][c] = zh_at[c];
          zhd_keep[kk][c] = zhd;
          on_keep[kk] |= (umax[c] > 0.f ? 1u : 0u) << c;
        }
        *reinterpret_cast<floatx4*>(out_t + poff[j]) = o;
        *reinterpret_cast<floatx4*>(L.fd + (size_t)rl * F + poff_keep[kk]) = o;
      }
    }
  }
  __syncthreads();
  TAIL_STAMP(2);
  // ---- logit tangents and R{dlogits} (head_rows_kernel<true>'s body on the LDS copies)
  const int WS = WY + 2;
  for (int e = tid; e < rows * WY; e += TAIL_THREADS) {      // the primal dlogits of this workgroup's rows (stored by the primal pass)
    const int rl = e / WY, w = e - rl * WY;
    L.b[rl * WS + w] = h.dl[((size_t)task * N + rg + TAIL_GROUPS * rl) * WY + w];
  }
  for (int rl = tid >> 6; rl < rows; rl += TAIL_THREADS / 64)
    head_row_at<true>(h, task, rg + TAIL_GROUPS * rl, tid & 63, L.f + (size_t)rl * F, L.fd + (size_t)rl * F, L.wl, L.wld, L.bl, L.bld, 0, L.a + rl * WS);
  __syncthreads();
  TAIL_STAMP(3);
  for (int e = tid; e < rows * WY; e += TAIL_THREADS) {
    const int rl = e / WY, w = e - rl * WY;
    tail_st(tail_scr_row(t, task, rg + TAIL_GROUPS * rl, RL, WY) + w, L.a[rl * WS + w]);
  }
  TAIL_STAMP(4);
  // ---- this row group's share of R{dWl} = sum_n R{dl}[n] f[n] + dl[n] fd[n] (head_grads_kernel<true>: per row, in that order)
  {
    const mi_rsrc rwp = tail_rsrc(t.wpart + ((size_t)task * TAIL_GROUPS + rg) * WY * F, (size_t)WY * F * sizeof(float));
    for (int q = tid; q < (F >> 2); q += TAIL_THREADS) {
      floatx4 dw[8];
#pragma unroll
      for (int w = 0; w < 8; ++w) dw[w] = floatx4{0.f, 0.f, 0.f, 0.f};
      for (int rl = 0; rl < rows; ++rl) {
        const floatx4 fv = *reinterpret_cast<const floatx4*>(L.f + (size_t)rl * F + 4 * q);
        const floatx4 fdv = *reinterpret_cast<const floatx4*>(L.fd + (size_t)rl * F + 4 * q);
#pragma unroll
        for (int w = 0; w < 8; ++w)
          if (w < WY) {
            const float ra = L.a[rl * WS + w], rb = L.b[rl * WS + w];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              dw[w][c] = fmaf(ra, fv[c], dw[w][c]);
              dw[w][c] = fmaf(rb, fdv[c], dw[w][c]);
            }
          }
      }
#pragma unroll
      for (int w = 0; w < 8; ++w)
        if (w < WY) tail_st16(rwp, (unsigned)(((size_t)w * F + 4 * (size_t)q) * sizeof(float)), dw[w]);
    }
  }
  TAIL_STAMP(5);
  // ---- R{df} of this workgroup's rows and the tangent BatchNorm-backward terms: R{dbeta} += dud, R{dgamma} += dud zh + du zhd
  {
    double rgm[4] = {0, 0, 0, 0}, rbm[4] = {0, 0, 0, 0};
    float* df_t = h.df + (size_t)task * N * F;
    const float* dp_t = a.dp + (size_t)task * p_task;
    floatx4 dpv[TAIL_KMAX];
#pragma unroll
    for (int kk = 0; kk < TAIL_KMAX; ++kk) {                  // (all items' cotangent loads first)
      const int item = tid + kk * TAIL_THREADS, rl = item / per_row, n = rg + TAIL_GROUPS * rl;
      dpv[kk] = *reinterpret_cast<const floatx4*>(dp_t + (poff_keep[kk] < 0 ? (size_t)0 : (size_t)n * F + poff_keep[kk]));
    }
#pragma unroll
    for (int kk = 0; kk < TAIL_KMAX; ++kk) {
      if (poff_keep[kk] < 0) continue;
      const int item = tid + kk * TAIL_THREADS, rl = item / per_row, n = rg + TAIL_GROUPS * rl, fo = poff_keep[kk];
      const floatx4 d = dpv[kk];
      floatx4 s = {0.f, 0.f, 0.f, 0.f};
      for (int w = 0; w < WY; ++w) {
        const floatx4 wv = *reinterpret_cast<const floatx4*>(L.wl + (size_t)w * F + fo);
        const floatx4 wdv = *reinterpret_cast<const floatx4*>(L.wld + (size_t)w * F + fo);
        const float ra = L.a[rl * WS + w], rb = L.b[rl * WS + w];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          s[c] = fmaf(ra, wv[c], s[c]);
          s[c] = fmaf(rb, wdv[c], s[c]);
        }
      }
      *reinterpret_cast<floatx4*>(df_t + (size_t)n * F + fo) = s;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool on = (on_keep[kk] >> c) & 1u;
        const float du = on ? d[c] : 0.f, dud = on ? s[c] : 0.f;
        rbm[c] += (double)dud;
        rgm[c] += (double)dud * (double)zh_keep[kk][c] + (double)du * (double)zhd_keep[kk][c];
      }
    }
    TAIL_STAMP(6);
    tail_partial(rgm, rbm, L.red, quads, a.c, t.bpart + ((size_t)task * TAIL_GROUPS + rg) * 2 * a.c);
  }
  TAIL_STAMP(7);
  const bool last_wg = tail_arrive(t.counter, task, L.flag);
  TAIL_STAMP(8);
  if (!last_wg) return;
  tail_fold<true>(t, task, RL, true, L.red);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TAIL_STAMP(9);
}

// What the kernels assume: 4 | c, (c / 4) | TAIL_THREADS, c <= TAIL_THREADS / 2; 4 | feat; ways <= 8; a row group's (row, window, quad) items within
// TAIL_KMAX per thread; the LDS of the tangent kernel within the CU's 160 KB.
bool tail_supported(int n, int ho, int wo, int c, int pool, int feat, int ways) {
  if (c < 4 || c % 4 || TAIL_THREADS % (c / 4) || c > TAIL_THREADS / 2 || feat % 4 || ways < 1 || ways > 8 || n < 1) return false;
  const int wins = pool ? ((ho + 1) / 2) * ((wo + 1) / 2) : ho * wo;
  const int hp = pool ? ho / 2 : ho, wp = pool ? wo / 2 : wo;
  if (hp * wp * c != feat) return false;
  const long items = (long)((n + TAIL_GROUPS - 1) / TAIL_GROUPS) * wins * (c / 4);
  if (items > (long)TAIL_KMAX * TAIL_THREADS) return false;
  if ((long)n * (ways + 2) > 2L * TAIL_THREADS) return false;                 // the folding workgroup takes the row scalars two per thread
  return tail_lds_bytes(n, feat, ways, 1) <= 150 * 1024;
}
size_t tail_wpart_floats(int tasks, int feat, int ways) { return (size_t)tasks * TAIL_GROUPS * ways * feat; }
size_t tail_bpart_doubles(int tasks, int c) { return (size_t)tasks * TAIL_GROUPS * 2 * c; }
size_t tail_scr_floats(int tasks, int n, int ways) { return (size_t)tasks * TAIL_GROUPS * ((n + TAIL_GROUPS - 1) / TAIL_GROUPS) * (ways + 2); }

template <class K>
static hipError_t tail_launch(K kern, hipStream_t st, const TailArgs& t, int tasks, size_t lds, unsigned* attr_done) {
  if (lds > 64 * 1024) {
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (!(*attr_done & (1u << (dev & 31)))) {
      if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e != hipSuccess) return e;
      *attr_done |= 1u << (dev & 31);
    }
  }
  hipLaunchKernelGGL(kern, dim3(TAIL_GROUPS, tasks), dim3(TAIL_THREADS), lds, st, t);
  return hipGetLastError();
}

hipError_t launch_tail(hipStream_t st, const TailArgs& t, int tasks, int pool, int tangent) {
  if (!tail_supported(t.bn.n, t.bn.ho, t.bn.wo, t.bn.c, pool, t.hd.feat, t.hd.ways) || !t.counter || !t.wpart || !t.bpart || !t.scr || tasks > 65535)
    return hipErrorInvalidValue;
  const size_t lds = tail_lds_bytes(t.hd.n, t.hd.feat, t.hd.ways, tangent);
  static unsigned done[4] = {0, 0, 0, 0};
  if (tangent) return pool ? tail_launch(tail_tangent_kernel<1>, st, t, tasks, lds, &done[0]) : tail_launch(tail_tangent_kernel<0>, st, t, tasks, lds, &done[1]);
  return pool ? tail_launch(tail_fwd_bwd_kernel<1>, st, t, tasks, lds, &done[2]) : tail_launch(tail_fwd_bwd_kernel<0>, st, t, tasks, lds, &done[3]);
}

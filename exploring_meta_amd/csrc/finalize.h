// "Last workgroup folds": the per-workgroup fp64 partials of a statistics / reduction kernel are folded by whichever workgroup
// of the task finishes last, inside the producing kernel, instead of by a separate bn_finalize launch (88 launches of ~6.5 us
// per cfg2 meta-iteration; 18 of the 109 launches of the launch-bound one-step configuration).  The fold itself walks the
// partials in the same fixed order as bn_finalize_kernel, so the result does not depend on which workgroup happens to be last:
// the arrival counter only decides WHO folds (an atomic on an integer, no floating-point atomics anywhere).
//
// Protocol: with a counter present the partials are written with agent-scope (write-through, "sc1") stores and read back by
// the folding workgroup with agent-scope loads, so no cache-wide operation is needed: a seq_cst __threadfence() on gfx950 is a
// buffer_wbl2 + buffer_inv of the XCD's whole L2, and issuing it from every finishing workgroup while the others are still
// streaming their outputs through that L2 made the cfg2 meta-iteration 1.7x SLOWER (measured: 42.4 vs 24.5 ms).  Ordering:
// each writer waits for its stores to be acknowledged (s_waitcnt vmcnt(0)), the workgroup
// barrier orders them before thread 0's relaxed agent-scope increment of the task's arrival counter; the workgroup that
// observes arrivals - 1 folds.  It also resets the counter, so the buffer is all-zero again when the kernel ends and the next
// launch on the same stream can re-use it.
#pragma once
#include "mi_common.h"

enum { FIN_STATS = 0, FIN_TSTATS = 1, FIN_SUMS = 2 };

struct FinArgs {
  unsigned* counter;   // [tasks], zero on entry; nullptr = the caller launches bn_finalize itself
  float* out0; float* out1;
  size_t stride0, stride1;
  double inv_m;
  int mode;            // FIN_*
};

//  FIN_STATS : (sum z, sum z^2)        -> out0 = mean, out1 = 1/sqrt(biased var + eps)
//  FIN_TSTATS: (sum zd, sum zh zd)     -> out0 = m1,   out1 = m2
//  FIN_SUMS  : (first, second)         -> out0 = first, out1 = second (e.g. dgamma, dbeta)
__device__ __forceinline__ void mi_fin_store(double s, double q, const FinArgs& f, int task, int ch) {
  float o0, o1;
  if (f.mode == FIN_STATS) {
    const double mean = s * f.inv_m;
    double var = q * f.inv_m - mean * mean;
    var = var > 0.0 ? var : 0.0;
    o0 = (float)mean;
    o1 = (float)(1.0 / sqrt(var + MI_BN_EPS));
  } else if (f.mode == FIN_TSTATS) {
    o0 = (float)(s * f.inv_m);
    o1 = (float)(q * f.inv_m);
  } else {
    o0 = (float)s;
    o1 = (float)q;
  }
  f.out0[(size_t)task * f.stride0 + ch] = o0;
  f.out1[(size_t)task * f.stride1 + ch] = o1;
}

// One fp64 partial: plain store without a counter, agent-scope write-through store with one (visible to every XCD once acknowledged).
__device__ __forceinline__ void mi_partial_store(double* p, double v, const FinArgs& f) {
  if (f.counter) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// Fold partial[nblk][2][c] of one task with the first 256 threads of the workgroup: thread (slice, channel) folds every
// `slices`-th partial, the slices are then folded in order.  `red`: 512 doubles of LDS.  COHERENT: agent-scope loads (the
// partials were written by other workgroups of the same kernel, possibly through another XCD's L2).
template <bool COHERENT>
__device__ __forceinline__ void mi_fold_partials(const double* p, int nblk, int c, const FinArgs& f, int task, double* red) {
  const int t = threadIdx.x;
  const int slices = 256 / c, sl = t / c, ch = t - sl * c;
  double s = 0.0, q = 0.0;
  if (t < 256 && sl < slices) {
    // eight partials in flight per thread (the write-through loads miss every cache: ~1 us each if issued one by one; a few-task launch
    // leaves up to ~370 partials per task, 46 per thread), summed in the same order as a one-by-one loop
    const size_t step = (size_t)slices * 2 * c;
    const double* ps = p + (size_t)sl * 2 * c + ch;
    int b = sl;
    for (; b + 7 * slices < nblk; b += 8 * slices, ps += 8 * step) {
      double v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (COHERENT) {
          v0[u] = __hip_atomic_load(ps + u * step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v1[u] = __hip_atomic_load(ps + u * step + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          v0[u] = ps[u * step];
          v1[u] = ps[u * step + c];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s += v0[u]; q += v1[u]; }
    }
    for (; b < nblk; b += slices, ps += step) {
      if (COHERENT) {
        s += __hip_atomic_load(ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q += __hip_atomic_load(ps + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        s += ps[0];
        q += ps[c];
      }
    }
  }
  if (t < 256) {
    red[t] = s;
    red[256 + t] = q;
  }
  __syncthreads();
  if (t >= c) return;
  s = 0.0; q = 0.0;
  for (int k = 0; k < slices; ++k) {
    s += red[k * c + ch];
    q += red[256 + k * c + ch];
  }
  mi_fin_store(s, q, f, task, ch);
}

// Call from EVERY thread of the workgroup after the workgroup's partial has been written with mi_partial_store (by any of its
// threads).  `arrivals` = workgroups that contribute to this task's partials (gridDim.x, times gridDim.z where channel tiles
// share a row).  `red` may alias LDS the kernel no longer needs (>= 512 doubles + 1 int); the function synchronises first.
__device__ __forceinline__ void mi_finalize_last(const FinArgs& f, const double* partial_task, int nblk, int c, int task,
                                                 unsigned arrivals, double* red) {
  if (!f.counter) return;                        // uniform: separate finalize launch
  int* flag = reinterpret_cast<int*>(red + 512);
  // this thread's write-through stores must be ACKNOWLEDGED before the barrier that precedes the counter increment (a
  // workgroup-scope release fence does not wait for vmcnt on gfx950: waves of a workgroup share the CU's L1 / the XCD's L2)
  // inline asm with a memory clobber, not __builtin_amdgcn_s_waitcnt: the builtin is IntrNoMem, so nothing would stop the compiler
  // from sinking the sc1 stores below it, and a builtin wait can make a later pass drop waits it believes redundant
  // (MI355X_MICROARCH.md, "Compiler hazard"); the asm statement is opaque to both
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(f.counter + task, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (prev + 1u == arrivals);
    if (last) __hip_atomic_store(f.counter + task, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return;                            // uniform per workgroup
  // the fold's sc1 loads must not be hoisted above the counter read that made this workgroup the folder (compiler ordering only:
  // they bypass the L1, and the last arriver's fetch_add returned after every other workgroup's stores were acknowledged)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  mi_fold_partials<true>(partial_task, nblk, c, f, task, red);
}

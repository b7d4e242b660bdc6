// Kernels of the fused Fisher-vector-product sweeps: see policy_sweep.h for the design.
#include "policy_sweep.h"

#ifndef SW_EXP
#define SW_EXP 0        // (diagnostic builds, tools/sweep_exp.sh: 1 = the first slab's rows requested AFTER the weights, 2 / 3 = without the direction's / both W2 loads -- wrong results)
#endif
#define SW_STAMP(k) do { if (a.stamps && blockIdx.x == 0 && tid == 0) a.stamps[nstamp++] = ((unsigned long long)(k) << 56) | (__builtin_amdgcn_s_memtime() & 0x00FFFFFFFFFFFFFFull); } while (0)
__device__ __forceinline__ floatx4 lds4(const float* p) { return *reinterpret_cast<const floatx4*>(p); }

// NG groups of 4 reduction steps: operands of the next chunk of 4 groups are issued BEFORE the 16 MFMAs of the current chunk, and the
// scheduling fences keep hipcc from sinking each LDS read next to its use (it otherwise emits read / s_waitcnt / two MFMAs, and the
// matrix pipe idles for one LDS latency in every pair: measured 129 cycles per MFMA instead of 64)
template <int NG, class LA, class LB>
__device__ __forceinline__ void mfma_groups(floatx16& acc, LA la, LB lb) {
  constexpr int CH = 4, NC = (NG + CH - 1) / CH;
  floatx4 av[2][CH], bv[2][CH];
#pragma unroll
  for (int g = 0; g < CH; ++g)
    if (g < NG) { av[0][g] = la(g); bv[0][g] = lb(g); }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int cur = c & 1, nxt = cur ^ 1;
#pragma unroll
    for (int g = 0; g < CH; ++g) {
      const int j = (c + 1) * CH + g;
      if (j < NG) { av[nxt][g] = la(j); bv[nxt][g] = lb(j); }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < CH; ++g) {
      const int j = c * CH + g;
      if (j < NG) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][g][k], bv[cur][g][k], acc, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int H, int MODE>
__global__ __launch_bounds__(256) void policy_sweep_kernel(SweepArgs a) {
  constexpr bool HVP = MODE == SW_HVP, PRIMAL = MODE == SW_PRIMAL;
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
  // A workgroup takes a.spw consecutive VIRTUAL slabs of the stream [task 0: marker, slab 0 .. spt-1][task 1: marker, ...]: the marker in
  // front of every task's slabs stands for what entering a task costs (flushing the previous task's partial, loading the new task's
  // tables: about half a slab's time), so a workgroup whose run crosses a task boundary gets one slab less than the others instead of
  // being the straggler every other workgroup waits for (with plain runs of 5 slabs 16 of cfg5's 252 workgroups crossed a boundary and
  // the launch took their 227k cycles against 210k).
  const int vspt = spt + 1;
  const int v0 = blockIdx.x * a.spw, v1 = min(v0 + a.spw, a.T * vspt);

  // ---- accumulators that live across the slabs of one task
  floatx16 accW2[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) accW2[m][r] = 0.f;
  // vector-stage mapping: thread = (column col of the hidden layer, half of the slab's rows)
  const int col = tid & 127, rb = (tid >> 7) * 16;
  float accW3[SW_MAX_A], accb3r[SW_MAX_A], accb2 = 0.f, accb1 = 0.f, accW1[SW_MAX_S], accrho[SW_MAX_A], accloss = 0.f, acckl = 0.f;
#pragma unroll
  for (int d = 0; d < SW_MAX_A; ++d) { accW3[d] = 0.f; accb3r[d] = 0.f; }
#pragma unroll
  for (int s = 0; s < SW_MAX_S; ++s) accW1[s] = 0.f;
#pragma unroll
  for (int d = 0; d < SW_MAX_A; ++d) accrho[d] = 0.f;

  for (int e = tid; e < 32 * SW_MAX_S; e += 256) xs[e] = 0.f;
  for (int e = tid; e < H * SW_MAX_S; e += 256) W1d[e] = 0.f;
  for (int e = tid; e < SW_MAX_A * H; e += 256) { W3s[e] = 0.f; W3d[e] = 0.f; }
  for (int e = tid; e < 32 * SW_MAX_A; e += 256) { dmus[e] = 0.f; rdmus[e] = 0.f; }
  __syncthreads();
  auto load_weights = [&](int t) {
    const float* th = a.theta + (size_t)t * a.tstride;
    const float* dv = PRIMAL ? th : a.dir + (size_t)t * a.dstride;      // primal sweep: the "direction" tables hold theta's own W1, b1, b2, b3
    // Every global load of the task's tables is requested BEFORE the first LDS store waits for one: the small tables first (into
    // registers), then the two matrices.  Table by table (load, wait, store) the cold start of a sweep paid five memory latencies in a
    // row -- 26k cycles of every launch (tools/sweep_stamps.py), a ninth of a five-slab workgroup's time.
    if constexpr (PRIMAL) {        // (the primal variant sits at the register limit and runs 4 of a step's 38 sweeps: table by table, as before)
#pragma unroll 20
      for (int e = tid; e < HH; e += 256) W2s[e] = th[a.o_w2 + e];
      for (int e = tid; e < H * S; e += 256) W1d[(e / S) * SW_MAX_S + e % S] = dv[a.o_w1 + e];
      for (int e = tid; e < H; e += 256) { b1d[e] = dv[a.o_b1 + e]; b2d[e] = dv[a.o_b2 + e]; }
      for (int e = tid; e < A * H; e += 256) { W3s[e] = th[a.o_w3 + e]; W3d[e] = dv[a.o_w3 + e]; }
      if (tid < A) {
        b3d[tid] = dv[a.o_b3 + tid]; rho[tid] = th[a.o_sigma + tid];
        rhod[tid] = a.surrogate ? a.old_scale[(size_t)t * A + tid] : 1.f;       // the OLD policy's scale
      }
      return;
    }
    constexpr int N1 = (H * SW_MAX_S + 255) / 256, N3 = (SW_MAX_A * H + 255) / 256;
    float w1r[N1], w3sr[N3], w3dr[N3], b1r = 0.f, b2r = 0.f, b3r = 0.f, rhor = 0.f, rhodr = 0.f;
#pragma unroll
    for (int i = 0; i < N1; ++i) { const int e = tid + 256 * i; w1r[i] = e < H * S ? dv[a.o_w1 + e] : 0.f; }
    if (tid < H) { b1r = dv[a.o_b1 + tid]; b2r = dv[a.o_b2 + tid]; }
#pragma unroll
    for (int i = 0; i < N3; ++i) {
      const int e = tid + 256 * i;
      w3sr[i] = e < A * H ? th[a.o_w3 + e] : 0.f;
      w3dr[i] = PRIMAL ? 0.f : (e < A * H ? dv[a.o_w3 + e] : 0.f);      // (primal: the same table)
    }
    if (tid < A) {
      b3r = dv[a.o_b3 + tid]; rhor = th[a.o_sigma + tid];
      rhodr = PRIMAL ? (a.surrogate ? a.old_scale[(size_t)t * A + tid] : 1.f) : dv[a.o_sigma + tid];      // primal: the OLD policy's scale
    }
    auto store_small = [&]() {
#pragma unroll
      for (int i = 0; i < N1; ++i) { const int e = tid + 256 * i; if (e < H * S) W1d[(e / S) * SW_MAX_S + e % S] = w1r[i]; }
      if (tid < H) { b1d[tid] = b1r; b2d[tid] = b2r; }
#pragma unroll
      for (int i = 0; i < N3; ++i) { const int e = tid + 256 * i; if (e < A * H) { W3s[e] = w3sr[i]; W3d[e] = PRIMAL ? w3sr[i] : w3dr[i]; } }
      if (tid < A) { b3d[tid] = b3r; rho[tid] = rhor; rhod[tid] = rhodr; }
    };
    __builtin_amdgcn_sched_barrier(0);
    if (PRIMAL) {
      store_small();                                  // (the primal variant sits at the register limit: its small tables do not stay live under the matrix)
#pragma unroll 20
      for (int e = tid; e < HH; e += 256) W2s[e] = th[a.o_w2 + e];
    } else if ((((size_t)(th + a.o_w2) | (size_t)(dv + a.o_w2)) & 7) == 0) {      // 8-byte loads where both matrices are 8-byte aligned
      // Chunks of W2_CH + W2_CH loads issued together, then their LDS stores.  Written as one loop of load / store pairs hipcc keeps two
      // or three loads in flight (s_waitcnt vmcnt(0) in front of every store: it schedules for register pressure), and the two matrices
      // cost 13k cycles of round trips at the start of every sweep (tools/sweep_exp.sh).  Raw buffer loads: the range check ends the
      // matrix, no predicated load.
      constexpr int W2_CH = 10, W2_N = (HH / 2 + 255) / 256;      // H = 100: 20 pairs of floats per thread and matrix
      typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
      const mi_rsrc rws = __builtin_amdgcn_make_buffer_rsrc((void*)(th + a.o_w2), 0, (unsigned)(HH * 4), 0x00020000);
      const mi_rsrc rwd = __builtin_amdgcn_make_buffer_rsrc((void*)(dv + a.o_w2), 0, (unsigned)(HH * 4), 0x00020000);
#pragma unroll
      for (int c0 = 0; c0 < W2_N; c0 += W2_CH) {
        u32x2 vs[W2_CH], vd[W2_CH];
#pragma unroll
        for (int k = 0; k < W2_CH; ++k) {
          const unsigned off = (unsigned)(8 * tid + 2048 * (c0 + k));
          if (c0 + k < W2_N && SW_EXP != 3) vs[k] = __builtin_amdgcn_raw_buffer_load_b64(rws, off, 0, 0);
          if (c0 + k < W2_N && SW_EXP != 2 && SW_EXP != 3) vd[k] = __builtin_amdgcn_raw_buffer_load_b64(rwd, off, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < W2_CH; ++k) {
          const int e = 2 * tid + 512 * (c0 + k);
          if (c0 + k < W2_N && e < HH) {
            if (SW_EXP != 3) *reinterpret_cast<u32x2*>(W2s + e) = vs[k];
            if (SW_EXP != 2 && SW_EXP != 3) *reinterpret_cast<u32x2*>(W2d + e) = vd[k];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll 16
      for (int e = tid; e < HH; e += 256) { W2s[e] = th[a.o_w2 + e]; W2d[e] = dv[a.o_w2 + e]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!PRIMAL) store_small();
  };

  // one partial [P] per (workgroup, task): slot = this workgroup's position among the workgroups that touch the task
  auto flush = [&](int t) {
    const int slot = blockIdx.x - (t * vspt + 1) / a.spw;       // position among the workgroups that hold slabs of task t
    float* pv = a.partial + ((size_t)t * a.slots + slot) * a.pitch;
    const int icol = 32 * wave + n;
    // the 64 store addresses hang off ONE lane offset that the compiler cannot see through: otherwise it hoists 64 loop-invariant
    // 64-bit offsets out of the slab loop and the kernel spills (measured: 261 spilled registers)
    int wbase = a.o_w2 + 4 * hh * H + icol;
    asm volatile("" : "+v"(wbase));
    float* pw = pv + wbase;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int orel = 32 * m + (r & 3) + 8 * (r >> 2);      // o = orel + 4 hh
        if (orel + 4 * hh < H && icol < H) pw[orel * H] = accW2[m][r];
        accW2[m][r] = 0.f;
      }
    // W1 / b1 partials sit per lane half (16 rows each), rho partials per row-thread: fold through LDS (the slab arrays are free)
    float* t1 = h1s;                                   // [2][H][S + 1]
    if (icol < H) {
#pragma unroll
      for (int s = 0; s < SW_MAX_S; ++s) if (s < S) t1[(hh * H + icol) * (SW_MAX_S + 1) + s] = accW1[s];
      t1[(hh * H + icol) * (SW_MAX_S + 1) + SW_MAX_S] = accb1;
    }
    float* t3 = h2s;                                   // [2][SW_MAX_A + 1][128]: W3 / b2 partials per (row half, column)
    if (col < H) {
#pragma unroll
      for (int d = 0; d < SW_MAX_A; ++d) t3[((tid >> 7) * (SW_MAX_A + 1) + d) * 128 + col] = accW3[d];
      t3[((tid >> 7) * (SW_MAX_A + 1) + SW_MAX_A) * 128 + col] = accb2;
    }
    float* t2 = h1d;                                   // [32][A] rho partials, [32][2] loss / KL partials of the primal sweep, [32][A] b3 partials
    if (tid < 32) {
#pragma unroll
      for (int d = 0; d < SW_MAX_A; ++d) if (d < A) { t2[tid * SW_MAX_A + d] = accrho[d]; t2[32 * SW_MAX_A + 64 + tid * SW_MAX_A + d] = accb3r[d]; }
      if (PRIMAL) { t2[32 * SW_MAX_A + 2 * tid] = accloss; t2[32 * SW_MAX_A + 2 * tid + 1] = acckl; }
    }
    __syncthreads();
    if (tid < H) {
      for (int s = 0; s < S; ++s) pv[a.o_w1 + tid * S + s] = t1[tid * (SW_MAX_S + 1) + s] + t1[(H + tid) * (SW_MAX_S + 1) + s];
      pv[a.o_b1 + tid] = t1[tid * (SW_MAX_S + 1) + SW_MAX_S] + t1[(H + tid) * (SW_MAX_S + 1) + SW_MAX_S];
    }
    if (tid < A) {
      float s = 0.f;
      for (int r = 0; r < 32; ++r) s += t2[r * SW_MAX_A + tid];
      pv[a.o_sigma + tid] = s;
      float b = 0.f;
      for (int r = 0; r < 32; ++r) b += t2[32 * SW_MAX_A + 64 + r * SW_MAX_A + tid];
      pv[a.o_b3 + tid] = b;
    }
    if (PRIMAL && tid >= 64 && tid < 66) {
      float s = 0.f;
      for (int r = 0; r < 32; ++r) s += t2[32 * SW_MAX_A + 2 * r + (tid - 64)];
      pv[a.P + (tid - 64)] = s;
    }
    for (int idx = tid; idx < A * H; idx += 256) {
      const int d = idx / H, k = idx - d * H;
      pv[a.o_w3 + idx] = t3[d * 128 + k] + t3[((SW_MAX_A + 1) + d) * 128 + k];
    }
    if (tid < H) pv[a.o_b2 + tid] = t3[SW_MAX_A * 128 + tid] + t3[((SW_MAX_A + 1) + SW_MAX_A) * 128 + tid];
    __syncthreads();
    accb1 = 0.f; accb2 = 0.f; accloss = 0.f; acckl = 0.f;
#pragma unroll
    for (int d = 0; d < SW_MAX_A; ++d) { accW3[d] = 0.f; accb3r[d] = 0.f; }
#pragma unroll
    for (int s = 0; s < SW_MAX_S; ++s) accW1[s] = 0.f;
#pragma unroll
    for (int d = 0; d < SW_MAX_A; ++d) accrho[d] = 0.f;
  };

  // the NEXT slab's global data travels in registers under this slab's last two matrix stages
  constexpr int NPF = (SL + 1023) / 1024;
  struct Prefetch { floatx4 v1[NPF], v2[NPF], v3[NPF]; float x, act, mu, dmu, coef; } pf;
  auto fetch = [&](int t, int row0, Prefetch& f) {
    const int nv = min(32, B - row0);
    const size_t rbase = (size_t)t * B + row0;
    if (!PRIMAL) {
      // raw buffer loads: one descriptor per tensor and slab whose record count is the slab's valid bytes -- rows past the batch read 0
      // from the hardware range check, no predication (a predicated global load is an exec-mask branch region each)
      const unsigned bytes = (unsigned)(nv * H * 4);
      const mi_rsrc r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.h1 + rbase * H), 0, bytes, 0x00020000);
      const mi_rsrc r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.h2 + rbase * H), 0, bytes, 0x00020000);
      const mi_rsrc r3 = __builtin_amdgcn_make_buffer_rsrc((void*)((HVP ? a.d2 : a.h2) + rbase * H), 0, bytes, 0x00020000);
#pragma unroll
      for (int i = 0; i < NPF; ++i) {
        const unsigned off = (unsigned)(tid * 16 + 4096 * i);
        f.v1[i] = buf_ld16(r1, off);
        f.v2[i] = buf_ld16(r2, off);
        if (HVP) f.v3[i] = buf_ld16(r3, off);
      }
    }
    f.x = (tid < nv * S) ? a.x[rbase * S + tid] : 0.f;
    f.act = f.mu = f.dmu = f.coef = 0.f;
    if (HVP) {
      if (tid < nv * A) { f.act = a.act[rbase * A + tid]; f.mu = a.mu[rbase * A + tid]; f.dmu = a.dmu[rbase * A + tid]; }
      if (tid < nv) f.coef = a.coef[rbase + tid];           // rows past count[t] are masked at staging (no wait on count here)
    }
    if (PRIMAL) {                                           // actions, advantages, the old policy's mean
      if (tid < nv * A) { f.act = a.act[rbase * A + tid]; if (a.surrogate) f.mu = a.old_loc[rbase * A + tid]; }
      if (tid < nv) f.coef = a.adv[rbase + tid];
    }
  };

  int cur = -1;
  int nstamp = 0;
  SW_STAMP(0);
  int t = v0 / vspt, vr = v0 - t * vspt;               // (task, position in the task's virtual slabs) of v, carried along: no division per slab
  for (int v = v0; v < v1; ++v, ++vr) {
    if (vr == vspt) { vr = 0; ++t; }
    if (vr == 0) continue;                             // a task's marker
    const int row0 = (vr - 1) * 32;
    if (t != cur) {
      if (cur >= 0) flush(cur);
      if (cur < 0 && SW_EXP != 1) fetch(t, row0, pf);  // the first slab's rows fly under the weight load
      if (cur < 0 || a.tstride != 0 || a.dstride != 0) load_weights(t);
      if (cur < 0 && SW_EXP == 1) fetch(t, row0, pf);
      cur = t;
    }
    SW_STAMP(1);
    const int cnt = a.count ? a.count[t] : B;
    // ---- stage the slab from the registers its data was fetched into (16-byte coalesced copies of the stored activations; rows past
    // the batch are zero; the per-row scalars)
    if (!PRIMAL) {
#pragma unroll
      for (int i = 0; i < NPF; ++i) {
        const int e = tid * 4 + 1024 * i;
        if (e < SL) {
          *reinterpret_cast<floatx4*>(h1s + e) = pf.v1[i];
          *reinterpret_cast<floatx4*>(h2s + e) = pf.v2[i];
          if (HVP) *reinterpret_cast<floatx4*>(d2s + e) = pf.v3[i];
        }
      }
    }
    if (tid < 32 * S) xs[(tid / S) * SW_MAX_S + tid % S] = pf.x;
    if (HVP || PRIMAL) {
      if (tid < 32 * A) { const int q = (tid / A) * SW_MAX_A + tid % A; acts[q] = pf.act; mus[q] = pf.mu; if (HVP) dmus[q] = pf.dmu; }
      if (tid < 32) coefs[tid] = (row0 + tid < cnt) ? pf.coef : 0.f;
    }
    const size_t rbase = (size_t)t * B + row0;
    const int nv = min(32, B - row0);                 // rows of this slab inside the padded batch
    __syncthreads();
    SW_STAMP(2);
    // ---- tangent of the first hidden layer (S-wide: vector work): h1d = [h1 > 0] (x W1d^T + b1d)
    if (col < H) {
      const floatx4 w1 = lds4(W1d + col * SW_MAX_S);   // rows past S are zero
      const float bb = b1d[col];
#pragma unroll
      for (int jb = 0; jb < 16; jb += 8) {             // 8 rows' operands in flight, then the arithmetic
        floatx4 xv[8];
        float hv0[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xv[j] = lds4(xs + (rb + jb + j) * SW_MAX_S);
          if (!PRIMAL) hv0[j] = h1s[(rb + jb + j) * H + col];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = rb + jb + j;
          const float z = fmaf(xv[j][3], w1[3], fmaf(xv[j][2], w1[2], fmaf(xv[j][1], w1[1], fmaf(xv[j][0], w1[0], bb))));
          if (PRIMAL) {                                // the layer itself: h1 = relu(x W1^T + b1), kept for the later sweeps
            const float hv = fmaxf(z, 0.f);
            h1s[r * H + col] = hv;
            if (r < nv) a.h1_out[(rbase + r) * H + col] = hv;
          } else {
            h1d[r * H + col] = hv0[j] > 0.f ? z : 0.f;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    SW_STAMP(3);
    // ---- tangent of the second hidden layer on the matrix pipe: z2d = h1 W2d^T + h1d W2^T + b2d, h2d = [h2 > 0] z2d
    {
      floatx16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int ocol = min(32 * wave + n, H - 1);
#pragma unroll
      for (int term = 0; term < (PRIMAL ? 1 : 2); ++term) {
        const float* arow = (term == 0 ? h1s : h1d) + n * H + hh * KH0;
        const float* brow = ((term == 0 && !PRIMAL) ? W2d : W2s) + ocol * H + hh * KH0;
        const floatx4 z4 = {0.f, 0.f, 0.f, 0.f};
        mfma_groups<NG>(acc,
                        [&](int j) { return (KH0 + 4 * j >= H && hh) ? z4 : lds4(arow + 4 * j); },     // upper half's k past the end
                        [&](int j) { return (KH0 + 4 * j >= H && hh) ? z4 : lds4(brow + 4 * j); });
      }
      const int o = 32 * wave + n;
      if (o < H) {
        const float bb = b2d[o];
#pragma unroll
        for (int r = 0; r < 16; ++r) {                 // the ReLU gate [h2 > 0] is applied by the two readers of h2d (they hold h2 anyway)
          const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (PRIMAL) {                                // h2 = relu(h1 W2^T + b2)
            const float hv = fmaxf(acc[r] + bb, 0.f);
            h2s[row * H + o] = hv;
            if (row < nv) a.h2_out[(rbase + row) * H + o] = hv;
          } else {
            h2d[row * H + o] = acc[r] + bb;
          }
        }
      }
    }
    __syncthreads();
    SW_STAMP(4);
    // ---- head tangent (A-wide): mud = h2 W3d^T + h2d W3^T + b3d; partial dot products per (row, eighth of the columns)
    // Two action dimensions (the reference's tasks: 2D navigation, rl/maml_trpo.py) take a copy of the stage with the dimension count as a
    // constant: with the run-time count every dimension is a branch region, and the stage was a chain of 16 LDS latencies (2.8k cycles).
    if (!PRIMAL && A == 2) {                         // (the primal variant sits at the register limit: it keeps the general form)
      constexpr int AA = 2;
      const int r = tid & 31, q = tid >> 5;
      float sacc[AA] = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < (H / 4 + 7) / 8; ++j) {
        const int kq = q + 8 * j;
        const bool okq = kq < H / 4;
        const int kc = okq ? kq : H / 4 - 1;          // (a chunk past the row reads the row's last chunk and is zeroed: no branch)
        floatx4 hv = lds4(h2s + r * H + 4 * kc);
        floatx4 hd = PRIMAL ? hv : lds4(h2d + r * H + 4 * kc);
        floatx4 ws[AA], wd[AA];
#pragma unroll
        for (int d = 0; d < AA; ++d) {
          ws[d] = lds4(W3s + d * H + 4 * kc);
          wd[d] = PRIMAL ? ws[d] : lds4(W3d + d * H + 4 * kc);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          hd[c] = (okq && hv[c] > 0.f) ? hd[c] : 0.f;
          hv[c] = okq ? hv[c] : 0.f;
        }
#pragma unroll
        for (int d = 0; d < AA; ++d) {
          if (PRIMAL) {                                // mu = h2 W3^T + b3
#pragma unroll
            for (int c = 0; c < 4; ++c) sacc[d] = fmaf(hv[c], ws[d][c], sacc[d]);
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) sacc[d] = fmaf(hv[c], wd[d][c], fmaf(hd[c], ws[d][c], sacc[d]));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int d = 0; d < AA; ++d) red[(d * 8 + q) * 32 + r] = sacc[d];
    } else
    {
      const int r = tid & 31, q = tid >> 5;
      float sacc[SW_MAX_A];
#pragma unroll
      for (int d = 0; d < SW_MAX_A; ++d) sacc[d] = 0.f;
#pragma unroll
      for (int j = 0; j < (H / 4 + 7) / 8; ++j) {
        // no branch on the chunk index (it differs between the lane halves): a chunk past the row reads the row's last chunk and is
        // zeroed -- unconditional loads can all be in flight together
        const int kq = q + 8 * j;
        const bool okq = kq < H / 4;
        const int kc = okq ? kq : H / 4 - 1;
        floatx4 hv = lds4(h2s + r * H + 4 * kc);
        floatx4 hd = PRIMAL ? hv : lds4(h2d + r * H + 4 * kc);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          hd[c] = (okq && hv[c] > 0.f) ? hd[c] : 0.f;
          hv[c] = okq ? hv[c] : 0.f;
        }
#pragma unroll
        for (int d = 0; d < SW_MAX_A; ++d) {
          if (d >= A) break;
          const floatx4 ws = lds4(W3s + d * H + 4 * kc);
          if (PRIMAL) {                                // mu = h2 W3^T + b3
#pragma unroll
            for (int c = 0; c < 4; ++c) sacc[d] = fmaf(hv[c], ws[c], sacc[d]);
          } else {
            const floatx4 wd = lds4(W3d + d * H + 4 * kc);
#pragma unroll
            for (int c = 0; c < 4; ++c) sacc[d] = fmaf(hv[c], wd[c], fmaf(hd[c], ws[c], sacc[d]));
          }
        }
      }
#pragma unroll
      for (int d = 0; d < SW_MAX_A; ++d) if (d < A) red[(d * 8 + q) * 32 + r] = sacc[d];
    }
    __syncthreads();
    if (tid < 32 * A) {
      const int r = tid & 31, d = tid >> 5;
      float m = b3d[d];
      for (int q = 0; q < 8; ++q) m += red[(d * 8 + q) * 32 + r];
      muds[r * SW_MAX_A + d] = m;
      if (PRIMAL && a.mu_out && r < nv) a.mu_out[(rbase + r) * A + d] = m;
    }
    __syncthreads();
    SW_STAMP(5);
    // ---- Gaussian part, one thread per row: the cotangent of mu that the tangent backward starts from (and the sigma slots)
    if (tid < 32) {
      const bool valid = tid < nv && row0 + tid < cnt;
      const float invD = 1.f / (float)A;
      if (PRIMAL) {
        // the loss of the pass per row (reference rl.py:358 / 459-469): its value, dL/dlogp (kept for the Hessian-vector sweeps), the
        // cotangent of mu the backward starts from, and the sigma gradient
        const float invB = 1.f / (float)cnt;
        float lp = 0.f, lpo = 0.f, klb = 0.f;
#pragma unroll
        for (int d = 0; d < SW_MAX_A; ++d) {
          if (d >= A) break;
          const float rr = fmaxf(rho[d], LOG_EPS), sg = expf(rr);
          const float df = acts[tid * SW_MAX_A + d] - muds[tid * SW_MAX_A + d];
          lp += -(df * df) / (2.f * sg * sg) - rr - HALF_LOG_2PI;
          if (a.surrogate) {
            const float so = rhod[d], lo = mus[tid * SW_MAX_A + d];
            const float dfo = acts[tid * SW_MAX_A + d] - lo;
            lpo += -(dfo * dfo) / (2.f * so * so) - logf(so) - HALF_LOG_2PI;
            const float vr = (sg / so) * (sg / so), t1 = (muds[tid * SW_MAX_A + d] - lo) / so;
            klb += 0.5f * (vr + t1 * t1 - 1.f - logf(vr));
          }
        }
        lp *= invD;
        lpo *= invD;
        float c = 0.f;
        if (valid) {
          const float ad = coefs[tid];
          if (!a.surrogate) { c = -ad * invB; accloss += c * lp; }
          else { const float ratio = expf(lp - lpo); c = -ratio * ad * invB; accloss += c; acckl += klb * invB * invD; }
        }
        if (a.coef_out && tid < nv) a.coef_out[rbase + tid] = c;
#pragma unroll
        for (int d = 0; d < SW_MAX_A; ++d) {
          if (d >= A) break;
          const float rp = rho[d];
          const float rr = fmaxf(rp, LOG_EPS), sg = expf(rr), iv = 1.f / (sg * sg);
          const float df = acts[tid * SW_MAX_A + d] - muds[tid * SW_MAX_A + d];
          const float dm = c * invD * df * iv;
          rdmus[tid * SW_MAX_A + d] = dm;
          if (a.dmu_out && tid < nv) a.dmu_out[(rbase + tid) * A + d] = dm;
          if (rp > LOG_EPS) accrho[d] += c * invD * (df * df * iv - 1.f);
        }
      } else if (HVP) {
        const float c = coefs[tid];                    // 0 on padding rows
#pragma unroll
        for (int d = 0; d < SW_MAX_A; ++d) {
          if (d >= A) break;
          const float rp = rho[d];
          const bool live = rp > LOG_EPS;
          const float rr = fmaxf(rp, LOG_EPS), sg = expf(rr), iv = 1.f / (sg * sg);
          const float rd = live ? rhod[d] : 0.f;
          const float df = acts[tid * SW_MAX_A + d] - mus[tid * SW_MAX_A + d];
          const float md = muds[tid * SW_MAX_A + d];
          rdmus[tid * SW_MAX_A + d] = c * invD * (-md * iv - 2.f * df * rd * iv);
          if (live) accrho[d] += c * invD * (-2.f * df * md * iv - 2.f * df * df * rd * iv);
        }
      } else {
        const float invB = 1.f / (float)cnt;
        for (int d = 0; d < A; ++d) {
          const float rr = fmaxf(rho[d], LOG_EPS), sg = expf(rr);
          rdmus[tid * SW_MAX_A + d] = valid ? muds[tid * SW_MAX_A + d] * invB * invD / (sg * sg) : 0.f;
        }
      }
    }
    __syncthreads();
    SW_STAMP(6);
    // ---- head weight gradient, then r2 = [h2 > 0] (rdmu W3 + dmu W3d) in place of h2d: a thread owns one column and 16 rows, reads
    // its own h2d element before overwriting it (no barrier in between), and sums its rows of r2 for the bias gradient
    const bool bwd = !(PRIMAL && a.fwd_only);        // uniform: a forward-only primal sweep stops after the loss
    // (the head bias gradient: every row thread of the Gaussian stage keeps the sum of its row's cotangents, folded at the flush like the
    // rho partials -- one thread per action dimension summing the slab's 32 rows here was 32 dependent LDS reads, 2k cycles of wave 0 in
    // front of the stage's barrier)
    if (bwd && tid < 32) {
#pragma unroll
      for (int d = 0; d < SW_MAX_A; ++d) if (d < A) accb3r[d] += rdmus[tid * SW_MAX_A + d];
    }
    if (bwd && col < H) {
      float w3[SW_MAX_A], w3d[SW_MAX_A];
#pragma unroll
      for (int d = 0; d < SW_MAX_A; ++d) { w3[d] = W3s[d * H + col]; w3d[d] = W3d[d * H + col]; }     // rows past A are zero
      if (!PRIMAL && A == 2) {                       // (two action dimensions: one pair, no branch per pair -- as the head-tangent stage)
        typedef float floatx2 __attribute__((ext_vector_type(2)));
#pragma unroll 8
        for (int j = 0; j < 16; ++j) {
          const int r = rb + j;
          const float h2v = h2s[r * H + col], h2raw = h2d[r * H + col];
          const floatx2 rm = *reinterpret_cast<const floatx2*>(rdmus + r * SW_MAX_A);
          floatx2 dm = {0.f, 0.f};
          if (HVP) dm = *reinterpret_cast<const floatx2*>(dmus + r * SW_MAX_A);
          const float h2dv = h2v > 0.f ? h2raw : 0.f;
          float v = 0.f;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            accW3[e] = fmaf(rm[e], h2v, accW3[e]);
            v = fmaf(rm[e], w3[e], v);
            if (HVP) { accW3[e] = fmaf(dm[e], h2dv, accW3[e]); v = fmaf(dm[e], w3d[e], v); }
          }
          v = h2v > 0.f ? v : 0.f;
          h2d[r * H + col] = v;
          accb2 += v;
        }
      } else {
#pragma unroll 8
      for (int j = 0; j < 16; ++j) {
        const int r = rb + j;
        const float h2v = h2s[r * H + col], h2raw = h2d[r * H + col];
        const float h2dv = h2v > 0.f ? h2raw : 0.f;
        float v = 0.f;
        typedef float floatx2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int dp = 0; dp < SW_MAX_A / 2; ++dp) {    // action dimensions in pairs: one 8-byte broadcast read per table and pair
          if (2 * dp >= A) break;
          const floatx2 rm = *reinterpret_cast<const floatx2*>(rdmus + r * SW_MAX_A + 2 * dp);
          floatx2 dm = {0.f, 0.f};
          if (HVP) dm = *reinterpret_cast<const floatx2*>(dmus + r * SW_MAX_A + 2 * dp);
#pragma unroll
          for (int e = 0; e < 2; ++e) {                // a dimension past A has zero table entries and zero weights: contributes 0
            accW3[2 * dp + e] = fmaf(rm[e], h2v, accW3[2 * dp + e]);
            v = fmaf(rm[e], w3[2 * dp + e], v);
            if (HVP) { accW3[2 * dp + e] = fmaf(dm[e], h2dv, accW3[2 * dp + e]); v = fmaf(dm[e], w3d[2 * dp + e], v); }
          }
        }
        v = h2v > 0.f ? v : 0.f;
        h2d[r * H + col] = v;
        if (PRIMAL && a.d2_out && r < nv) a.d2_out[(rbase + r) * H + col] = v;      // dz2 of the pass, for its Hessian-vector sweeps
        accb2 += v;
      }
      }
    }
    __syncthreads();
    SW_STAMP(8);
    {                                                 // the next slab's data flies under the two matrix stages below
      const bool last_of_task = vr + 1 == vspt;        // then the next slab is the next task's first, behind its marker
      if (v + (last_of_task ? 2 : 1) < v1) fetch(last_of_task ? t + 1 : t, last_of_task ? 0 : row0 + 32, pf);
    }
    const float* r2 = h2d;
    // ---- r1 = [h1 > 0] (r2 W2 + d2 W2d) on the matrix pipe; its products with the states (W1, b1 gradients) from the accumulators
    if (bwd) {
      floatx16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int icol = min(32 * wave + n, H - 1);
      float gate[16];                                  // [h1 > 0] of this lane's 16 output elements, in flight under the products
#pragma unroll
      for (int r = 0; r < 16; ++r) gate[r] = h1s[((r & 3) + 8 * (r >> 2) + 4 * hh) * H + icol];
#pragma unroll
      for (int term = 0; term < (HVP ? 2 : 1); ++term) {
        const float* arow = (term == 0 ? r2 : d2s) + n * H + hh * KH0;
        const float* bcol = (term == 0 ? W2s : W2d) + icol + hh * KH0 * H;
        const floatx4 z4 = {0.f, 0.f, 0.f, 0.f};
        mfma_groups<NG>(acc,
                        [&](int j) { return (KH0 + 4 * j >= H && hh) ? z4 : lds4(arow + 4 * j); },
                        [&](int j) {                                  // the upper half's last group does not exist: stay in bounds
                          const float* bp = (KH0 + 4 * j >= H && hh) ? bcol : bcol + (4 * j) * H;
                          floatx4 b;
#pragma unroll
                          for (int c = 0; c < 4; ++c) b[c] = bp[c * H];
                          return b;
                        });
      }
      const int i = 32 * wave + n;
      if (i < H) {
#pragma unroll
        for (int rb8 = 0; rb8 < 16; rb8 += 8) {        // the states of 8 rows in flight (one 16-byte broadcast read per row), then the FMAs
          floatx4 xv[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = rb8 + q;
            xv[q] = lds4(xs + ((r & 3) + 8 * (r >> 2) + 4 * hh) * SW_MAX_S);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = rb8 + q;
            const float v = gate[r] > 0.f ? acc[r] : 0.f;
            accb1 += v;
#pragma unroll
            for (int s = 0; s < SW_MAX_S; ++s) accW1[s] = fmaf(v, xv[q][s], accW1[s]);       // states past S are zero
          }
        }
      }
    }
    SW_STAMP(9);
    // ---- dW2[o][i] += sum_rows r2[row][o] h1[row][i] + d2[row][o] h1d[row][i]: M = o (MT tiles), N = this wave's columns, K = rows
    if (bwd) {
      const int icol = 32 * wave + n;                  // columns past H read the next row: finite, never stored
#pragma unroll
      for (int term = 0; term < (HVP ? 2 : 1); ++term) {
        const float* am = (term == 0 ? r2 : d2s) + (16 * hh) * H + n;
        const float* bm = (term == 0 ? h1s : h1d) + (16 * hh) * H + icol;
#pragma unroll
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
    SW_STAMP(10);
    __syncthreads();                                   // the next slab's staging overwrites the slab arrays
  }
  SW_STAMP(11);
  if (cur >= 0) flush(cur);
  SW_STAMP(12);
}

__device__ __forceinline__ float sweep_fold_sum(const FoldArgs& f, int t, int p) {
  const int vspt = f.spt + 1;                          // (the sweep's virtual slab stream: one marker in front of every task's slabs)
  const int first = (t * vspt + 1) / f.spw, last = (t * vspt + f.spt) / f.spw;
  const float* pp = f.partial + (size_t)t * f.slots * f.pitch + p;
  float s = 0.f;
  const int ns = last - first + 1;
  int k = 0;
  for (; k + 4 <= ns; k += 4) {                       // four slots in flight, summed in slot order
    const float v0 = pp[(size_t)k * f.pitch], v1 = pp[(size_t)(k + 1) * f.pitch], v2 = pp[(size_t)(k + 2) * f.pitch], v3 = pp[(size_t)(k + 3) * f.pitch];
    s += v0; s += v1; s += v2; s += v3;
  }
  for (; k < ns; ++k) s += pp[(size_t)k * f.pitch];
  return s;
}
// mode 2 with the mean over tasks folded in (FoldArgs::counter): the arrival protocol of finalize.h -- write-through stores, acknowledged,
// workgroup barrier, one relaxed agent-scope increment per workgroup; the workgroup that arrives last reads the T values back with
// agent-scope loads, in task order
__global__ __launch_bounds__(256) void policy_sweep_fold_mean_kernel(FoldArgs f) {
  __shared__ int last_s;
  const int p = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y;
  const bool act = p < f.P;
  if (act) {
    const float s = sweep_fold_sum(f, t, p);
    __hip_atomic_store(f.out + (size_t)t * f.P + p, f.w[(size_t)t * f.P + p] - f.lr * s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(f.counter + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = prev + 1u == (unsigned)f.T;
    if (last) __hip_atomic_store(f.counter + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_s = last;
  }
  __syncthreads();
  if (!last_s || !act) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float s = 0.f;
  int k = 0;
  for (; k + 4 <= f.T; k += 4) {                        // four tasks in flight, summed in task order (as mean_tasks_kernel does)
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __hip_atomic_load(f.out + (size_t)(k + u) * f.P + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s += v[0]; s += v[1]; s += v[2]; s += v[3];
  }
  for (; k < f.T; ++k) s += __hip_atomic_load(f.out + (size_t)k * f.P + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  f.mean_out[p] = s * f.inv_T + f.damping * f.v[p];
}

__global__ __launch_bounds__(256) void policy_sweep_fold_kernel(FoldArgs f) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= f.P) {
    if (f.mode == 3 && p < f.P + 2) {                  // the primal sweep's loss / KL slots
      const float s = sweep_fold_sum(f, blockIdx.y, p);
      if (p == f.P && f.loss_t) f.loss_t[blockIdx.y] = s;
      if (p == f.P + 1 && f.kl_t) f.kl_t[blockIdx.y] = s;
    }
    return;
  }
  const int t = blockIdx.y;
  const float s = sweep_fold_sum(f, t, p);
  if (f.mode == 0) {
    f.out[(size_t)t * f.P + p] = f.v[p] - f.lr * s;
  } else if (f.mode == 2) {
    f.out[(size_t)t * f.P + p] = f.w[(size_t)t * f.P + p] - f.lr * s;
  } else if (f.mode == 3) {
    if (f.out) f.out[(size_t)t * f.P + p] = s;
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

bool policy_sweep_supported(int act_relu, int h1, int h2, int s, int a) {
  return act_relu && h1 == 100 && h2 == 100 && s <= SW_MAX_S && a <= SW_MAX_A;
}

template <int MODE>
static hipError_t launch_sweep_t(hipStream_t st, const SweepArgs& a, int grid) {
  static bool attr_set = false;
  const size_t lds = policy_sweep_lds_bytes<100>();
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&policy_sweep_kernel<100, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((policy_sweep_kernel<100, MODE>), dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}
hipError_t launch_policy_sweep(hipStream_t st, const SweepArgs& a, int grid, int mode) {
  if (mode == SW_HVP) return launch_sweep_t<SW_HVP>(st, a, grid);
  if (mode == SW_FISHER) return launch_sweep_t<SW_FISHER>(st, a, grid);
  if (mode == SW_PRIMAL) return launch_sweep_t<SW_PRIMAL>(st, a, grid);
  return hipErrorInvalidValue;
}
hipError_t launch_policy_sweep_fold(hipStream_t st, const FoldArgs& f, int tasks) {
  if (f.mode == 2 && f.counter) {
    if (!f.mean_out || !f.v || f.T != tasks) return hipErrorInvalidValue;
    hipLaunchKernelGGL(policy_sweep_fold_mean_kernel, dim3(ceil_div(f.P, 256), tasks), dim3(256), 0, st, f);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(policy_sweep_fold_kernel, dim3(ceil_div(f.P + 2, 256), tasks), dim3(256), 0, st, f);
  return hipGetLastError();
}

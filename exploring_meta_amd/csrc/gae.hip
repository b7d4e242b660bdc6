// Generalised advantage estimation with cherry's LinearValue baseline, for a whole list of replays in one launch.
// Replaces (reference): core_functions/rl.py:95-110 `compute_advantages` -- ch.td.discount, baseline.fit (cherry.models.robotics
// LinearValue: features [s, s^2, t, t^2, t^3, 1], t = row / 100; ridge normal equations), baseline(states), baseline(next_states),
// bootstraps, cherry.pg.generalized_advantage -- and the ch.normalize applied to its result at every call site (rl.py:355,
// trpo_a2c_loss / vpg_a2c_loss / ppo_update).  In the reference this is torch-CPU work per task and replay, re-done at every
// evaluation of the meta-surrogate; here it is one workgroup per replay, all arithmetic in fp64 (the reference's fp32 baseline
// fit is the least conditioned step of the whole RL path), the replay staged in LDS.
//
//   returns_t = r_t + gamma (1 - d_t) returns_{t+1}                         (reverse scan, cut at every done)
//   w         = (F^T F + reg I)^-1 F^T returns                              (D = 2 S + 4 features)
//   boot_t    = V(s_t) (1 - d_t) + V(s'_t) d_t,   V(s) = f(s, t) . w
//   delta_t   = r_t + gamma (1 - d_t) boot_{t+1} - boot_t   (boot_n = 0)
//   adv_t     = delta_t + gamma tau (1 - d_t) adv_{t+1};   normalised: (adv - mean) / (std_unbiased + 1e-8)
#include "mi_common.h"
#include "../../include/mi_maml.h"
#include <string>

int mi_internal_fail(int code, const char* msg);   // engine.hip: sets the global error string

#define GAE_MAX_D 20        // 2 * state_dim + 4 with state_dim <= 8

struct GaeArgs {
  const float *states, *next_states, *rewards, *dones;   // [R][B][S], [R][B][S], [R][B], [R][B]
  const int32_t* count;                                  // [R] rows in use (<= B), or null = B
  float* adv;                                            // [R][B]
  double* weight;                                        // [R][D] baseline weights (may be null)
  const double* weight_in;                               // [R][D] given weights: no fit (update_vf = False), or null
  int B, S;
  double gamma, tau, reg;
  int normalize;
};

#define GAE_THREADS 1024

// t / 100.0 without the division sequence: q = RN(t * 0.01), r = t - 100 q (exact in an FMA), q + r * 0.01 -- the correctly rounded quotient
// (0.01 is the correctly rounded reciprocal of 100 and q is within one ulp: Markstein's final division step), so the time features are
// bit for bit those of `t / 100.0` (checked for every t below 2^24 on the host: tests/test_host_surface.py).
__device__ __forceinline__ double gae_time(int t) {
  const double td = (double)t, q = td * 0.01;
  return fma(fma(-q, 100.0, td), 0.01, q);
}
__device__ __forceinline__ double gae_feature(const float* st, int S, int k, int t) {
  if (k < S) return (double)st[k];
  if (k < 2 * S) { const double v = (double)st[k - S]; return v * v; }
  const double al = gae_time(t);
  if (k == 2 * S) return al;
  if (k == 2 * S + 1) return al * al;
  if (k == 2 * S + 2) return al * al * al;
  return 1.0;
}

// x_t = y_t + c (1 - d_t) x_{t+1}, cut at every done: the thread that owns the LAST row of an episode walks it backwards, four rows'
// operands requested at a time (one LDS latency per four steps of the recurrence instead of one per step)
__device__ __forceinline__ void gae_scan(const double* y, const float* dn, double* x, int n, double c, int tid) {
  for (int i = tid; i < n; i += GAE_THREADS) {
    if (dn[i] != 0.f || i == n - 1) {
      double carry = 0.0;
      int t = i;
      bool head = true, more = true;
      while (more) {
        double yv[4];
        float dv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int tt = max(t - k, 0); yv[k] = y[tt]; dv[k] = dn[tt]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (more) {
            if (t - k < 0 || (!(head && k == 0) && dv[k] != 0.f)) { more = false; }       // the row before the episode's first one
            else { carry = yv[k] + (c * (1.0 - (double)dv[k])) * carry; x[t - k] = carry; }
          }
        }
        head = false;
        t -= 4;
      }
    }
  }
}

__device__ __forceinline__ double gae_block_sum(double v, double* red, int tid) {   // fixed order: wave shuffle tree, then the waves' partials
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < GAE_THREADS / 64; ++k) s += red[k];
  return s;
}

__global__ __launch_bounds__(GAE_THREADS) void gae_kernel(GaeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int r = blockIdx.x, tid = threadIdx.x, B = a.B, S = a.S, D = 2 * S + 4;
  const int n = a.count ? min(max(a.count[r], 0), B) : B;
  double* rw = reinterpret_cast<double*>(smem);           // [B] rewards, later delta
  double* x = rw + B;                                     // [B] returns, later boot, later advantages
  double* mat = x + B;                                    // [D][D + 1] normal equations
  double* red = mat + GAE_MAX_D * (GAE_MAX_D + 1);        // [GAE_THREADS] partial sums of the pair products | block reductions | the scaling
  double* w = red + GAE_THREADS;                          // [D]
  float* dn = reinterpret_cast<float*>(w + GAE_MAX_D);    // [B]
  float* st = dn + B;                                     // [B][S]
  const float* st_g = a.states + (size_t)r * B * S;
  const float* ns_g = a.next_states + (size_t)r * B * S;
  float* out = a.adv + (size_t)r * B;
  for (int t = tid; t < B; t += GAE_THREADS) out[t] = 0.f;        // rows past the replay's length carry no advantage
  if (n == 0) return;
  for (int t = tid; t < n; t += GAE_THREADS) {
    rw[t] = (double)a.rewards[(size_t)r * B + t];
    dn[t] = a.dones[(size_t)r * B + t];
  }
  for (int e = tid; e < n * S; e += GAE_THREADS) st[e] = st_g[e];
  __syncthreads();
  if (!a.weight_in) {
  gae_scan(rw, dn, x, n, a.gamma, tid);
  __syncthreads();
  // normal equations: one (i <= j) entry of F^T F or one entry of F^T returns per thread "pair", the rows split over G groups
  const int npair = D * (D + 1) / 2 + D;
  const int G = max(1, GAE_THREADS / npair);
  for (int p0 = 0; p0 < npair; p0 += GAE_THREADS / G) {   // one sweep (npair <= 230 for state_dim <= 8)
    const int p = p0 + tid / G, g = tid % G;
    double s = 0.0;
    int i = 0, j = 0;
    const bool act = tid < (GAE_THREADS / G) * G && p < npair;
    if (act) {
      if (p < D * (D + 1) / 2) { int q = p; while (q >= D - i) { q -= D - i; ++i; } j = i + q; }
      else { i = p - D * (D + 1) / 2; j = -1; }
      for (int t = g; t < n; t += G) {
        const double fi = gae_feature(st + t * S, S, i, t);
        s += fi * (j < 0 ? x[t] : gae_feature(st + t * S, S, j, t));
      }
    }
    red[tid] = s;
    __syncthreads();
    if (act && g == 0) {
      double tot = 0.0;
      for (int k = 0; k < G; ++k) tot += red[tid + k];
      if (j < 0) mat[i * (D + 1) + D] = tot;
      else { mat[i * (D + 1) + j] = tot; mat[j * (D + 1) + i] = tot; }
    }
    __syncthreads();
  }
  // (F^T F + reg I) w = F^T returns: symmetric diagonal scaling (the polynomial time features span 12 orders of magnitude), then Gaussian
  // elimination with partial pivoting -- one thread per matrix entry, every entry through exactly the operations of the one-thread loop
  // (an LDS latency per entry and pivot step there: 36 us of a 2000-row replay's 320)
  {
    const int W = D + 1, ei = tid / W, ej = tid - ei * W;
    const bool ent = tid < D * W;
    double* sc = red;
    if (tid < D) mat[tid * W + tid] += a.reg;
    __syncthreads();
    if (tid < D) { const double dg = mat[tid * W + tid]; sc[tid] = dg > 0.0 ? 1.0 / sqrt(dg) : 1.0; }
    __syncthreads();
    if (ent) mat[tid] *= ej < D ? sc[ei] * sc[ej] : sc[ei];
    __syncthreads();
    for (int c = 0; c < D; ++c) {
      int piv = c;
      double best = fabs(mat[c * W + c]);
      for (int i = c + 1; i < D; ++i) { const double v = fabs(mat[i * W + c]); if (v > best) { best = v; piv = i; } }
      __syncthreads();
      if (piv != c && tid >= c && tid <= D) { const double tmp = mat[c * W + tid]; mat[c * W + tid] = mat[piv * W + tid]; mat[piv * W + tid] = tmp; }
      __syncthreads();
      const double pv = mat[c * W + c];
      const double inv = pv != 0.0 ? 1.0 / pv : 0.0;        // a zero pivot (all-zero feature column, reg = 0) leaves that weight at 0: the minimum-norm choice
      const bool upd = ent && ei > c && ej >= c;
      double f = 0.0, mij = 0.0, mcj = 0.0;
      if (upd) { f = mat[ei * W + c] * inv; mij = mat[tid]; mcj = mat[c * W + ej]; }
      __syncthreads();
      if (upd && f != 0.0) mat[tid] = mij - f * mcj;
      __syncthreads();
    }
    if (tid == 0) {
      for (int c = D - 1; c >= 0; --c) {
        double s = mat[c * W + D];
        for (int j = c + 1; j < D; ++j) s -= mat[c * W + j] * w[j];
        const double pv = mat[c * W + c];
        w[c] = pv != 0.0 ? s / pv : 0.0;
      }
      for (int i = 0; i < D; ++i) {
        w[i] *= sc[i];
        if (a.weight) a.weight[(size_t)r * D + i] = w[i];
      }
    }
  }
  } else if (tid < D) {
    w[tid] = a.weight_in[(size_t)r * D + tid];
    if (a.weight) a.weight[(size_t)r * D + tid] = w[tid];
  }
  __syncthreads();
  // bootstraps -> x, deltas -> rw
  for (int t = tid; t < n; t += GAE_THREADS) {
    double v = 0.0, nv = 0.0;
    for (int k = 0; k < D; ++k) {
      v += gae_feature(st + t * S, S, k, t) * w[k];
      nv += gae_feature(ns_g + (size_t)t * S, S, k, t) * w[k];
    }
    const double d = (double)dn[t];
    x[t] = v * (1.0 - d) + nv * d;
  }
  __syncthreads();
  for (int t = tid; t < n; t += GAE_THREADS) {
    const double nxt = t + 1 < n ? x[t + 1] : 0.0;
    rw[t] = rw[t] + (a.gamma * (1.0 - (double)dn[t])) * nxt - x[t];
  }
  __syncthreads();
  gae_scan(rw, dn, x, n, a.gamma * a.tau, tid);
  __syncthreads();
  double mean = 0.0, istd = 1.0;
  if (a.normalize && n > 1) {
    double s = 0.0;
    for (int t = tid; t < n; t += GAE_THREADS) s += x[t];
    mean = gae_block_sum(s, red, tid) / (double)n;
    double q = 0.0;
    for (int t = tid; t < n; t += GAE_THREADS) { const double dv = x[t] - mean; q += dv * dv; }
    const double var = gae_block_sum(q, red, tid) / (double)(n - 1);
    istd = 1.0 / (sqrt(var) + 1e-8);
  }
  for (int t = tid; t < n; t += GAE_THREADS) out[t] = (float)((x[t] - mean) * istd);
}

static size_t gae_smem_bytes(int B, int S) {
  return (size_t)B * (8 + 8 + 4 + 4 * (size_t)S) + (GAE_MAX_D * (GAE_MAX_D + 1) + GAE_THREADS + GAE_MAX_D) * sizeof(double) + 64;
}

extern "C" int mi_gae_max_rows(int state_dim) {
  if (state_dim < 1 || 2 * state_dim + 4 > GAE_MAX_D) return 0;
  const size_t fixed = (GAE_MAX_D * (GAE_MAX_D + 1) + GAE_THREADS + GAE_MAX_D) * sizeof(double) + 64;
  return (int)((160 * 1024 - fixed) / (8 + 8 + 4 + 4 * (size_t)state_dim));
}

extern "C" int mi_gae_advantages(void* stream, const float* states, const float* next_states, const float* rewards, const float* dones,
                                 const int32_t* count, const double* weight_in, int replays, int rows, int state_dim, double gamma,
                                 double tau, double reg, int normalize, float* adv_out, double* weight_out) {
  if (!states || !next_states || !rewards || !dones || !adv_out || replays < 1 || rows < 1)
    return mi_internal_fail(MI_ERR_ARG, "mi_gae_advantages: null pointer or empty batch");
  if (state_dim < 1 || 2 * state_dim + 4 > GAE_MAX_D)
    return mi_internal_fail(MI_ERR_ARG, "mi_gae_advantages: state_dim must be in 1..8");
  if (rows > mi_gae_max_rows(state_dim))
    return mi_internal_fail(MI_ERR_ARG, ("mi_gae_advantages: a replay of " + std::to_string(rows) + " rows does not fit in LDS (max " +
                                         std::to_string(mi_gae_max_rows(state_dim)) + ")").c_str());
  GaeArgs a{states, next_states, rewards, dones, count, adv_out, weight_out, weight_in, rows, state_dim, gamma, tau, reg, normalize};
  const size_t smem = gae_smem_bytes(rows, state_dim);
  hipError_t s = hipSuccess;
  if (smem > 64 * 1024) s = hipFuncSetAttribute(reinterpret_cast<const void*>(gae_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (s == hipSuccess) {
    hipLaunchKernelGGL(gae_kernel, dim3(replays), dim3(GAE_THREADS), smem, reinterpret_cast<hipStream_t>(stream), a);
    s = hipGetLastError();
  }
  return s == hipSuccess ? MI_OK : mi_internal_fail(MI_ERR_HIP, hipGetErrorString(s));
}

// ---------------------------------------------------------------------------------------------------------------------
// Replay assembly: a list of contiguous fp32 device arrays (the fields of the replays of a meta-iteration, the parameters of the stored old
// policies) into padded batch tensors, one launch per 128 arrays.  The pointers travel in the kernel arguments (3 KB): no pointer table is
// uploaded, and the host does one ctypes call where the tensor library did a conversion chain per array and a concatenate / pad / gather
// per field (reference core_functions/rl.py:444-465 walks the replays one by one on the CPU).
#define SEG_MAX 128
struct SegArgs {
  const float* src[SEG_MAX];
  float* dst[SEG_MAX];
  unsigned n[SEG_MAX];         // floats copied
  unsigned npad[SEG_MAX];      // floats written in all (n .. npad: zeros)
};
__global__ __launch_bounds__(256) void copy_segments_kernel(SegArgs a) {
  const int sgi = blockIdx.y;
  const float* __restrict__ src = a.src[sgi];
  float* __restrict__ dst = a.dst[sgi];
  const unsigned n = a.n[sgi], np = a.npad[sgi];
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < np; e += gridDim.x * 256u) dst[e] = e < n ? src[e] : 0.f;
}

extern "C" int mi_copy_segments(void* stream, const void* const* src, void* const* dst, const uint32_t* nfloat, const uint32_t* npad, int nseg) {
  if (nseg < 0 || (nseg > 0 && (!src || !dst || !nfloat || !npad))) return mi_internal_fail(MI_ERR_ARG, "mi_copy_segments: null argument");
  for (int s0 = 0; s0 < nseg; s0 += SEG_MAX) {
    SegArgs a;
    const int m = nseg - s0 < SEG_MAX ? nseg - s0 : SEG_MAX;
    unsigned longest = 1;
    for (int k = 0; k < m; ++k) {
      if (npad[s0 + k] < nfloat[s0 + k] || (npad[s0 + k] && !dst[s0 + k]) || (nfloat[s0 + k] && !src[s0 + k]))
        return mi_internal_fail(MI_ERR_ARG, "mi_copy_segments: a segment without pointer, or padded to less than its length");
      a.src[k] = static_cast<const float*>(src[s0 + k]);
      a.dst[k] = static_cast<float*>(dst[s0 + k]);
      a.n[k] = nfloat[s0 + k];
      a.npad[k] = npad[s0 + k];
      if (npad[s0 + k] > longest) longest = npad[s0 + k];
    }
    for (int k = m; k < SEG_MAX; ++k) { a.src[k] = nullptr; a.dst[k] = nullptr; a.n[k] = 0; a.npad[k] = 0; }
    unsigned bx = (longest + 1023) / 1024;                // four elements per thread at the longest segment
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(copy_segments_kernel, dim3(bx, m), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mi_internal_fail(MI_ERR_HIP, hipGetErrorString(e));
  }
  return MI_OK;
}

struct I32Args { int32_t v[512]; };
__global__ __launch_bounds__(256) void upload_i32_kernel(I32Args a, int32_t* dst, int n) {
  for (int e = threadIdx.x; e < n; e += 256) dst[e] = a.v[e];
}
extern "C" int mi_upload_i32(void* stream, int32_t* dst, const int32_t* host_values, int n) {
  if (n < 0 || (n > 0 && (!dst || !host_values))) return mi_internal_fail(MI_ERR_ARG, "mi_upload_i32: null argument");
  for (int o = 0; o < n; o += 512) {
    I32Args a;
    const int m = n - o < 512 ? n - o : 512;
    for (int k = 0; k < m; ++k) a.v[k] = host_values[o + k];
    for (int k = m; k < 512; ++k) a.v[k] = 0;
    hipLaunchKernelGGL(upload_i32_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, dst + o, m);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mi_internal_fail(MI_ERR_HIP, hipGetErrorString(e));
  }
  return MI_OK;
}

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

__device__ __forceinline__ double gae_feature(const float* st, int S, int k, int t) {
  if (k < S) return (double)st[k];
  if (k < 2 * S) { const double v = (double)st[k - S]; return v * v; }
  const double al = (double)t / 100.0;
  if (k == 2 * S) return al;
  if (k == 2 * S + 1) return al * al;
  if (k == 2 * S + 2) return al * al * al;
  return 1.0;
}

// x_t = y_t + c (1 - d_t) x_{t+1}, cut at every done: the thread that owns the LAST row of an episode walks it backwards
__device__ __forceinline__ void gae_scan(const double* y, const float* dn, double* x, int n, double c, int tid) {
  for (int i = tid; i < n; i += 256) {
    if (dn[i] != 0.f || i == n - 1) {
      double carry = 0.0;
      int t = i;
      do {
        carry = y[t] + (c * (1.0 - (double)dn[t])) * carry;
        x[t] = carry;
        --t;
      } while (t >= 0 && dn[t] == 0.f);
    }
  }
}

__device__ __forceinline__ double gae_block_sum(double v, double* red, int tid) {   // fixed order: wave shuffle tree, then 4 partials
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void gae_kernel(GaeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int r = blockIdx.x, tid = threadIdx.x, B = a.B, S = a.S, D = 2 * S + 4;
  const int n = a.count ? min(max(a.count[r], 0), B) : B;
  double* rw = reinterpret_cast<double*>(smem);           // [B] rewards, later delta
  double* x = rw + B;                                     // [B] returns, later boot, later advantages
  double* mat = x + B;                                    // [D][D + 1] normal equations | scratch for the partial sums
  double* red = mat + GAE_MAX_D * (GAE_MAX_D + 1);        // [256] partial sums of the pair products, [4] block reductions
  double* w = red + 256;                                  // [D]
  float* dn = reinterpret_cast<float*>(w + GAE_MAX_D);    // [B]
  float* st = dn + B;                                     // [B][S]
  const float* st_g = a.states + (size_t)r * B * S;
  const float* ns_g = a.next_states + (size_t)r * B * S;
  float* out = a.adv + (size_t)r * B;
  for (int t = tid; t < B; t += 256) out[t] = 0.f;        // rows past the replay's length carry no advantage
  if (n == 0) return;
  for (int t = tid; t < n; t += 256) {
    rw[t] = (double)a.rewards[(size_t)r * B + t];
    dn[t] = a.dones[(size_t)r * B + t];
  }
  for (int e = tid; e < n * S; e += 256) st[e] = st_g[e];
  __syncthreads();
  if (!a.weight_in) {
  gae_scan(rw, dn, x, n, a.gamma, tid);
  __syncthreads();
  // normal equations: one (i <= j) entry of F^T F or one entry of F^T returns per thread "pair", the rows split over G groups
  const int npair = D * (D + 1) / 2 + D;
  const int G = max(1, 256 / npair);
  for (int p0 = 0; p0 < npair; p0 += 256 / G) {           // one sweep when npair * G <= 256 (always for state_dim <= 6)
    const int p = p0 + tid / G, g = tid % G;
    double s = 0.0;
    int i = 0, j = 0;
    const bool act = tid < (256 / G) * G && p < npair;
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
  if (tid == 0) {
    // (F^T F + reg I) w = F^T returns: symmetric diagonal scaling (the polynomial time features span 12 orders of magnitude),
    // then Gaussian elimination with partial pivoting
    double sc[GAE_MAX_D];
    for (int i = 0; i < D; ++i) { mat[i * (D + 1) + i] += a.reg; }
    for (int i = 0; i < D; ++i) { const double dg = mat[i * (D + 1) + i]; sc[i] = dg > 0.0 ? 1.0 / sqrt(dg) : 1.0; }
    for (int i = 0; i < D; ++i) {
      for (int j = 0; j < D; ++j) mat[i * (D + 1) + j] *= sc[i] * sc[j];
      mat[i * (D + 1) + D] *= sc[i];
    }
    for (int c = 0; c < D; ++c) {
      int piv = c;
      double best = fabs(mat[c * (D + 1) + c]);
      for (int i = c + 1; i < D; ++i) { const double v = fabs(mat[i * (D + 1) + c]); if (v > best) { best = v; piv = i; } }
      if (piv != c) for (int j = c; j <= D; ++j) { const double tmp = mat[c * (D + 1) + j]; mat[c * (D + 1) + j] = mat[piv * (D + 1) + j]; mat[piv * (D + 1) + j] = tmp; }
      const double pv = mat[c * (D + 1) + c];
      const double inv = pv != 0.0 ? 1.0 / pv : 0.0;        // a zero pivot (all-zero feature column, reg = 0) leaves that weight at 0: the minimum-norm choice
      for (int i = c + 1; i < D; ++i) {
        const double f = mat[i * (D + 1) + c] * inv;
        if (f != 0.0) for (int j = c; j <= D; ++j) mat[i * (D + 1) + j] -= f * mat[c * (D + 1) + j];
      }
    }
    for (int c = D - 1; c >= 0; --c) {
      double s = mat[c * (D + 1) + D];
      for (int j = c + 1; j < D; ++j) s -= mat[c * (D + 1) + j] * w[j];
      const double pv = mat[c * (D + 1) + c];
      w[c] = pv != 0.0 ? s / pv : 0.0;
    }
    for (int i = 0; i < D; ++i) {
      w[i] *= sc[i];
      if (a.weight) a.weight[(size_t)r * D + i] = w[i];
    }
  }
  } else if (tid < D) {
    w[tid] = a.weight_in[(size_t)r * D + tid];
    if (a.weight) a.weight[(size_t)r * D + tid] = w[tid];
  }
  __syncthreads();
  // bootstraps -> x, deltas -> rw
  for (int t = tid; t < n; t += 256) {
    double v = 0.0, nv = 0.0;
    for (int k = 0; k < D; ++k) {
      v += gae_feature(st + t * S, S, k, t) * w[k];
      nv += gae_feature(ns_g + (size_t)t * S, S, k, t) * w[k];
    }
    const double d = (double)dn[t];
    x[t] = v * (1.0 - d) + nv * d;
  }
  __syncthreads();
  for (int t = tid; t < n; t += 256) {
    const double nxt = t + 1 < n ? x[t + 1] : 0.0;
    rw[t] = rw[t] + (a.gamma * (1.0 - (double)dn[t])) * nxt - x[t];
  }
  __syncthreads();
  gae_scan(rw, dn, x, n, a.gamma * a.tau, tid);
  __syncthreads();
  double mean = 0.0, istd = 1.0;
  if (a.normalize && n > 1) {
    double s = 0.0;
    for (int t = tid; t < n; t += 256) s += x[t];
    mean = gae_block_sum(s, red, tid) / (double)n;
    double q = 0.0;
    for (int t = tid; t < n; t += 256) { const double dv = x[t] - mean; q += dv * dv; }
    const double var = gae_block_sum(q, red, tid) / (double)(n - 1);
    istd = 1.0 / (sqrt(var) + 1e-8);
  }
  for (int t = tid; t < n; t += 256) out[t] = (float)((x[t] - mean) * istd);
}

static size_t gae_smem_bytes(int B, int S) {
  return (size_t)B * (8 + 8 + 4 + 4 * (size_t)S) + (GAE_MAX_D * (GAE_MAX_D + 1) + 256 + GAE_MAX_D) * sizeof(double) + 64;
}

extern "C" int mi_gae_max_rows(int state_dim) {
  if (state_dim < 1 || 2 * state_dim + 4 > GAE_MAX_D) return 0;
  const size_t fixed = (GAE_MAX_D * (GAE_MAX_D + 1) + 256 + GAE_MAX_D) * sizeof(double) + 64;
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
    hipLaunchKernelGGL(gae_kernel, dim3(replays), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), a);
    s = hipGetLastError();
  }
  return s == hipSuccess ? MI_OK : mi_internal_fail(MI_ERR_HIP, hipGetErrorString(s));
}

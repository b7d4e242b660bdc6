// MAML-TRPO policy path (BASELINE config 5) for a whole meta-batch of tasks: DiagNormalPolicy MLP (reference
// core_functions/policies.py:30-67, ReLU default) log-prob / gradient / Hessian-vector products, the inner `trpo_update`
// (core_functions/rl.py:361-374), the meta surrogate loss + KL (rl.py:441-473), its gradient and the Fisher-vector product
// used by conjugate gradient (rl.py:413-418) -- replacing per-task PyTorch autograd graphs (create_graph=True) with
// batched-over-tasks kernels and forward-over-reverse tangents.  The dense products (41.6 MFLOP per 2000-row batch forward,
// SURVEY.md a13) run on the fp32 matrix pipe, one launch per layer for all tasks, deterministic reductions.
//
// Parameter vector (reference named_parameters() order): sigma[A], W1[H1][S], b1[H1], W2[H2][H1], b2[H2], W3[A][H2], b3[A].
//
// Mathematics (K = 1 inner step, the reference default): theta'_t = theta - lr g_t(theta), g_t = grad of -mean(logp A) on the
// support replay.  surrogate S_t(theta') = -mean(exp(logp_new - logp_old) A), KL_t(theta') = mean KL(new || old) on the query
// replay.   grad_theta mean_t S_t = mean_t (I - lr H_t) grad S_t(theta'_t)          (H_t v: tangent sweep over the support pass)
// At theta_old the adapted policy equals the stored old policy, KL's gradient is exactly zero, so
//   Fvp(v) = mean_t (I - lr H_t) F_t (I - lr H_t) v + damping v,   F_t = J^T diag(1/(B D sigma^2), 2/D) J  (Gaussian Fisher).
#include <string>
#include <vector>
#include <cmath>
#include "mi_common.h"
#include "../../include/mi_maml.h"

#define HALF_LOG_2PI 0.9189385332046727f

#include "policy_sweep.h"

#define MI_DENSE_TERMS 4   // products summed into one output (2 for first tangents, up to 4 for second tangents)
struct DenseArgs {
  const float* x[MI_DENSE_TERMS];     // [T][B][I]
  const float* w[MI_DENSE_TERMS];     // [O][I] per task (stride wstride[k] floats, 0 = shared)
  size_t wstride[MI_DENSE_TERMS];
  const float* bias;     // [O] (term 0 only), may be null
  size_t bstride;
  const float* mask;     // optional [T][B][O]: stored activations h; output multiplied by phi'(z) written in terms of h
  const float* hd;       // tanh tangent-backward only: tangent activations  [T][B][.]
  const float* dpre;     // tanh tangent-backward only: primal cotangent w.r.t. h (before the phi' factor)
  float* y;              // [T][B][O]
  float* ypre;           // optional: the value before the phi' factor (primal backward of a tanh layer keeps it for the HVP)
  int B, I, O, nterms;
  int act;               // forward: activation applied to the output (ACT_*); mask users: which phi' to form from h
};
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };
// phi'(z) from the stored activation h = phi(z): ReLU -> [h > 0], tanh -> 1 - h^2
__device__ __forceinline__ float act_gate(float s, float h, int act) {
  return act == ACT_TANH ? s * (1.f - h * h) : (h > 0.f ? s : 0.f);
}

// The three dense products of the MLP for ALL tasks in one launch each, on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32 is
// bitwise an fmaf chain, so the parity budget of the fp32 path is untouched).  One wave owns a 32 x 32 output tile; lane half
// h = lane >> 5 takes the upper / lower half of the reduction index, so a lane's A (and, for the forward product, B) operands of
// consecutive MFMAs are consecutive floats of one row.  Rows / columns / reduction indices past the end read zeros.
//   forward   y[b][o]  = sum_k x_k[b][:] . w_k[o][:]          M = B, N = O, K = I     (TRANS_W: B operand = w[n][k])
//   backward  dx[b][i] = sum_k dy_k[b][:] . w_k[:][i]         M = B, N = I, K = O     (B operand = w[k][n])
// followed by the same elementwise epilogues as before (bias / activation / phi' gate / tanh curvature term).
template <bool TRANS_W>
__global__ __launch_bounds__(256) void dense_mfma_kernel(DenseArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int t = blockIdx.y;
  const int N = TRANS_W ? a.O : a.I, Kd = TRANS_W ? a.I : a.O;
  const int nt = (N + 31) >> 5, mt = (a.B + 31) >> 5;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= mt * nt) return;
  const int m0 = (tile / nt) * 32, n0 = (tile % nt) * 32;
  const int Kh = (Kd + 1) >> 1, kbase = h * Kh;
  const int row = m0 + j, col = n0 + j;
  const bool rok = row < a.B, cok = col < N;
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int term = 0; term < a.nterms; ++term) {
    const float* xr = a.x[term] + ((size_t)t * a.B + (rok ? row : 0)) * Kd;
    const float* wt = a.w[term] + (size_t)t * a.wstride[term];
    if (Kd % 4 == 0 && Kd >= 16) {
      // 16-byte operand loads: the reduction index is dealt to the two lane halves in groups of four (half h takes k = 8c + 4h .. + 3),
      // so a lane's four consecutive MFMAs read four consecutive floats of its row.  An operand load touches 32 different rows
      // (64 cache lines) whatever its width: the address unit, not the matrix pipe, bounds these products, and twice the width is
      // half the loads.  (Rows are multiples of 16 bytes; weights inside the parameter vector are only 8-byte aligned: dword-aligned
      // vector type.)
      typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
      const int nch = (Kd + 7) >> 3;
#ifndef MI_DENSE_CU
#define MI_DENSE_CU 2
#endif
      constexpr int CU = MI_DENSE_CU;                         // chunks (of 4 MFMAs) in flight
      for (int c0 = 0; c0 < nch; c0 += CU) {
        f4u av[CU], bv[CU];
#pragma unroll
        for (int u = 0; u < CU; ++u) {
          const int k = 8 * (c0 + u) + 4 * h;
          const bool kok = (c0 + u < nch) && (k < Kd);
          av[u] = (f4u){0.f, 0.f, 0.f, 0.f}; bv[u] = (f4u){0.f, 0.f, 0.f, 0.f};
          if (kok && rok) av[u] = *reinterpret_cast<const f4u*>(xr + k);
          if (TRANS_W) {
            if (kok && cok) bv[u] = *reinterpret_cast<const f4u*>(wt + (size_t)col * Kd + k);
          } else if (kok && cok) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[u][q] = wt[(size_t)(k + q) * N + col];
          }
        }
#pragma unroll
        for (int u = 0; u < CU; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][q], bv[u][q], acc, 0, 0, 0);
      }
      continue;
    }
    constexpr int CH = 10;
    // a lane's consecutive reduction indices are consecutive floats of its row: 8-byte loads where the half-row split keeps them
    // 8-byte aligned (the 100-wide hidden layers: rows of 400 B, halves of 200 B)
    const bool vec2 = (Kd % 2 == 0) && (Kh % 2 == 0) &&
                      ((reinterpret_cast<uintptr_t>(xr + kbase) | (TRANS_W ? reinterpret_cast<uintptr_t>(wt + (size_t)(cok ? col : 0) * Kd + kbase) : 0)) & 7) == 0;
    for (int s0 = 0; s0 < Kh; s0 += CH) {
      float av[CH], bv[CH];
      if (vec2) {
#pragma unroll
        for (int c = 0; c < CH; c += 2) {
          const int k = kbase + s0 + c;
          const bool kok = (s0 + c < Kh) && (k < Kd);               // pairs never straddle Kh or Kd (both even)
          float2 a2 = make_float2(0.f, 0.f), b2 = make_float2(0.f, 0.f);
          if (kok && rok) a2 = *reinterpret_cast<const float2*>(xr + k);
          if (TRANS_W) {
            if (kok && cok) b2 = *reinterpret_cast<const float2*>(wt + (size_t)col * Kd + k);
          } else if (kok && cok) {
            b2.x = wt[(size_t)k * N + col];
            b2.y = wt[(size_t)(k + 1) * N + col];
          }
          av[c] = a2.x; av[c + 1] = a2.y; bv[c] = b2.x; bv[c + 1] = b2.y;
        }
      } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int k = kbase + s0 + c;
          const bool kok = (s0 + c < Kh) && (k < Kd);
          av[c] = (kok && rok) ? xr[k] : 0.f;
          if (TRANS_W) bv[c] = (kok && cok) ? wt[(size_t)col * Kd + k] : 0.f;
          else bv[c] = (kok && cok) ? wt[(size_t)k * N + col] : 0.f;
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bv[c], acc, 0, 0, 0);
    }
  }
  if (!cok) return;
  const float bias = (TRANS_W && a.bias) ? a.bias[(size_t)t * a.bstride + col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (m >= a.B) continue;
    const size_t oi = ((size_t)t * a.B + m) * N + col;
    float s = acc[r] + bias;
    if (TRANS_W) {
      if (a.mask) s = act_gate(s, a.mask[oi], a.act);        // tangent forward: hdot = phi'(z) zdot
      else if (a.act == ACT_RELU) s = fmaxf(s, 0.f);
      else if (a.act == ACT_TANH) s = tanhf(s);
    } else {
      if (a.ypre) a.ypre[oi] = s;
      if (a.mask) {
        const float hh = a.mask[oi];
        s = act_gate(s, hh, a.act);
        // R{dz} = phi' R{dh} + phi'' zdot dh, and for tanh phi'' zdot = -2 h hdot
        if (a.hd) s = fmaf(-2.f * hh * a.hd[oi], a.dpre[oi], s);
      }
    }
    a.y[oi] = s;
  }
}

template <bool TRANS_W>
static void launch_dense(hipStream_t st, int T, const DenseArgs& a) {
  // (an LDS-tiled variant -- 128 rows x all columns per workgroup, 32-wide reduction chunks staged with coalesced loads -- was built
  // and measured: 1.09 ms per Fisher-vector product against 1.01 ms for this kernel with 8-byte operand loads; two barriers per
  // chunk and the staging cost outweigh the better coalescing on these 100-wide layers.  A second variant -- every wave copies its
  // own contiguous 32-row operand blocks into a private LDS slice with 16-byte loads, no barriers, 2 waves per workgroup -- ran the
  // 100x100 forward product in 116 us against 87 us: 25.6 KB of LDS per wave leaves 6 waves per CU and the copy's latency is no
  // longer hidden.  The product is address-unit bound (32 rows per operand load) but short of a kernel that keeps the whole 2x100
  // MLP of a row tile on chip, occupancy is worth more than coalescing here.)
  const int N = TRANS_W ? a.O : a.I;
  hipLaunchKernelGGL(dense_mfma_kernel<TRANS_W>, dim3(ceil_div(ceil_div(a.B, 32) * ceil_div(N, 32), 4), T), dim3(256), 0, st, a);
}

struct DenseWArgs {
  const float* dy[MI_DENSE_TERMS];    // [T][B][O]
  const float* x[MI_DENSE_TERMS];     // [T][B][I]
  float* dw;             // [T][gstride] at offset: [O][I]
  float* db;             // [T][gstride] at offset: [O]  (from dy[0])
  size_t gstride;
  int B, I, O, nterms;
};
// dW[o][i] = sum_k sum_b dy_k[b][o] x_k[b][i] ; db[o] = sum_b dy_0[b][o]:  M = O, N = I + 1 (column I = the bias: a column of
// ones next to x_0), K = B.  One 8-wave workgroup per 32 x 32 tile; wave w takes the row pairs w, w + 8, ... of the batch (both
// operands are coalesced 128-B rows), the 8 partial tiles are folded through LDS in wave order => deterministic.
__global__ __launch_bounds__(512) void dense_wgrad_mfma_kernel(DenseWArgs a) {
  __shared__ float red[8 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int t = blockIdx.y;
  const int N = a.I + 1, nt = (N + 31) >> 5;
  const int m0 = (blockIdx.x / nt) * 32, n0 = (blockIdx.x % nt) * 32;
  const int orow = m0 + j, col = n0 + j;
  const bool ook = orow < a.O, xok = col < a.I, one = col == a.I;
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int term = 0; term < a.nterms; ++term) {
    const float* dy = a.dy[term] + (size_t)t * a.B * a.O + (ook ? orow : 0);
    const float* x = a.x[term] + (size_t)t * a.B * a.I + (xok ? col : 0);
    const float onev = (one && term == 0) ? 1.f : 0.f;
    constexpr int CH = 8;
    // (prefetching the next chunk into a second register set was measured: no gain, 1.05 vs 1.01 ms per Fisher-vector product)
    for (int p0 = wave * 2 * CH; p0 < a.B; p0 += 8 * 2 * CH) {      // this wave's chunk of CH row pairs
      float av[CH], bv[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int b = p0 + 2 * c + h;
        const bool bok = b < a.B;
        av[c] = (bok && ook) ? dy[(size_t)b * a.O] : 0.f;
        bv[c] = bok ? (xok ? x[(size_t)b * a.I] : onev) : 0.f;
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bv[c], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[r];
  __syncthreads();
  for (int e = threadIdx.x; e < 1024; e += 512) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += red[w * 1024 + e];
    const int r = e >> 6, l = e & 63;
    const int o = m0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), c = n0 + (l & 31);
    if (o >= a.O) continue;
    if (c < a.I) a.dw[(size_t)t * a.gstride + (size_t)o * a.I + c] = v;
    else if (c == a.I) a.db[(size_t)t * a.gstride + o] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
struct GaussArgs {
  const float* mu;        // [T][B][A]
  const float* mud;       // tangent of mu (tangent / fisher modes)
  const float* rho;       // sigma parameter, per task stride rstride (0 = shared)
  const float* rhod;      // tangent of rho (stride vstride)
  size_t rstride, vstride;
  const float* act;       // [T][B][A]
  const float* adv;       // [T][B]
  const float* old_loc;   // [T][B][A]
  const float* old_scale; // [T][A]
  const int32_t* count;   // [T] valid samples (rows >= count are padding)
  float* coef;            // [T][B]: dL/dlogp per sample (written in modes 0/1, read in mode 2)
  float* dmu;             // [T][B][A]
  float* drho;            // [T][gstride] at offset: [A]
  size_t gstride;
  float* loss;            // [T]
  float* kl;              // [T]
  int B, A, mode;
};
enum { G_A2C = 0, G_SURROGATE = 1, G_TANGENT = 2, G_FISHER = 3 };

// One workgroup per task.
//  G_A2C       L = -mean(logp*adv)                 (rl.py:358)  -> coef, dmu, drho, loss
//  G_SURROGATE L = -mean(exp(logp-logp_old)*adv)   (rl.py:469)  -> coef, dmu, drho, loss ; kl = mean KL(new||old) (rl.py:459-461)
//  G_TANGENT   R{dmu}, R{drho} of the coef-weighted log-prob gradient for tangents (mud, rhod)
//  G_FISHER    cotangent of the KL Hessian at new == old: dmu = mud/(B D sigma^2), drho = 2 rhod / D
__global__ __launch_bounds__(256) void gauss_kernel(GaussArgs a) {
  __shared__ float red[256];
  const int t = blockIdx.x, tid = threadIdx.x;
  const int B = a.B, A = a.A, cnt = a.count ? a.count[t] : B;
  const float invB = 1.f / (float)cnt, invD = 1.f / (float)A;
  float acc[8];   // loss, kl, drho[0..5]  (A <= 6)
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  for (int b = tid; b < B; b += 256) {
    const bool valid = b < cnt;
    const size_t ob = (size_t)t * B + b;
    float lp = 0.f, lpo = 0.f, klb = 0.f;
    if (a.mode == G_A2C || a.mode == G_SURROGATE) {
      for (int d = 0; d < A; ++d) {
        const float rp = a.rho[(size_t)t * a.rstride + d];
        const float r = fmaxf(rp, LOG_EPS), sg = expf(r);
        const float df = a.act[ob * A + d] - a.mu[ob * A + d];
        lp += -(df * df) / (2.f * sg * sg) - r - HALF_LOG_2PI;
        if (a.mode == G_SURROGATE) {
          const float so = a.old_scale[(size_t)t * A + d], lo = a.old_loc[ob * A + d];
          const float dfo = a.act[ob * A + d] - lo;
          lpo += -(dfo * dfo) / (2.f * so * so) - logf(so) - HALF_LOG_2PI;
          const float vr = (sg / so) * (sg / so), t1 = (a.mu[ob * A + d] - lo) / so;
          klb += 0.5f * (vr + t1 * t1 - 1.f - logf(vr));
        }
      }
      lp *= invD;
      lpo *= invD;
      float c = 0.f;
      if (valid) {
        const float ad = a.adv[ob];
        if (a.mode == G_A2C) { c = -ad * invB; acc[0] += c * lp; }
        else { const float ratio = expf(lp - lpo); c = -ratio * ad * invB; acc[0] += -ratio * ad * invB; acc[1] += klb * invB * invD; }
      }
      if (a.coef) a.coef[ob] = c;
      for (int d = 0; d < A; ++d) {
        const float rp = a.rho[(size_t)t * a.rstride + d];
        const float r = fmaxf(rp, LOG_EPS), sg = expf(r), iv = 1.f / (sg * sg);
        const float df = a.act[ob * A + d] - a.mu[ob * A + d];
        if (a.dmu) a.dmu[ob * A + d] = c * invD * df * iv;
        if (rp > LOG_EPS) acc[2 + d] += c * invD * (df * df * iv - 1.f);
      }
    } else if (a.mode == G_TANGENT) {
      const float c = valid ? a.coef[ob] : 0.f;
      for (int d = 0; d < A; ++d) {
        const float rp = a.rho[(size_t)t * a.rstride + d];
        const bool live = rp > LOG_EPS;
        const float r = fmaxf(rp, LOG_EPS), sg = expf(r), iv = 1.f / (sg * sg);
        const float rd = live ? a.rhod[(size_t)t * a.vstride + d] : 0.f;
        const float df = a.act[ob * A + d] - a.mu[ob * A + d];
        const float md = a.mud[ob * A + d];
        a.dmu[ob * A + d] = c * invD * (-md * iv - 2.f * df * rd * iv);
        if (live) acc[2 + d] += c * invD * (-2.f * df * md * iv - 2.f * df * df * rd * iv);
      }
    } else {  // G_FISHER
      for (int d = 0; d < A; ++d) {
        const float rp = a.rho[(size_t)t * a.rstride + d];
        const float r = fmaxf(rp, LOG_EPS), sg = expf(r);
        a.dmu[ob * A + d] = valid ? a.mud[ob * A + d] * invB * invD / (sg * sg) : 0.f;
      }
    }
  }
  // deterministic block reductions
  for (int k = 0; k < 2 + A; ++k) {
    red[tid] = acc[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) red[tid] += red[tid + s];
      __syncthreads();
    }
    if (tid == 0) {
      if (k == 0 && a.loss) a.loss[t] = red[0];
      if (k == 1 && a.kl) a.kl[t] = red[0];
      if (k >= 2 && a.drho) {
        float v = red[0];
        if (a.mode == G_FISHER) {
          const float rp = a.rho[(size_t)t * a.rstride + (k - 2)];
          v = rp > LOG_EPS ? 2.f * a.rhod[(size_t)t * a.vstride + (k - 2)] * invD : 0.f;
        }
        a.drho[(size_t)t * a.gstride + (k - 2)] = v;
      }
    }
    __syncthreads();
  }
}

__global__ void axpy_bcast_kernel(const float* __restrict__ a, size_t astride, const float* __restrict__ b, float alpha, int p,
                                  float* __restrict__ out) {   // out[t][i] = a[t*astride + i] - alpha b[t][i]
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p) return;
  const size_t t = blockIdx.y;
  out[t * p + i] = alpha == 0.f ? a[t * astride + i] : a[t * astride + i] - alpha * b[t * p + i];   // alpha 0: pure broadcast
}
__global__ void mean_tasks_kernel(const float* __restrict__ x, int tasks, int p, float scale, const float* __restrict__ add,
                                  float add_scale, float* __restrict__ out) {   // out[i] = scale*sum_t x[t][i] + add_scale*add[i]
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p) return;
  float s = 0.f;
  for (int t = 0; t < tasks; ++t) s += x[(size_t)t * p + i];
  out[i] = s * scale + (add ? add_scale * add[i] : 0.f);
}

// ---------------------------------------------------------------------------------------------------------------------
struct mi_policy {
  mi_policy_desc d;
  int device;
  int S, A, H1, H2;
  int act;   // ACT_RELU / ACT_TANH between the dense layers (policies.py:32-37,76)
  size_t o_sigma, o_w1, o_b1, o_w2, o_b2, o_w3, o_b3, P;
  std::string err;
  unsigned* fold_counters = nullptr;   // device, one per 256-parameter block: arrival counters of the fold that also takes the mean over tasks
  bool fold_dirty = false;             // a counted fold was issued and not seen to launch cleanly: re-zero the counters before the next one
                                       // (policy_sweep.h FoldArgs::counter; zero between launches).  Allocated at the first fused product.
};
static thread_local std::string g_perr;
static int pfail(mi_policy* p, int code, const std::string& m) {
  if (p) p->err = m;
  g_perr = m;
  return code;
}
#define PCHK(p, call)                                                                                 \
  do {                                                                                                \
    hipError_t _s = (call);                                                                           \
    if (_s != hipSuccess) return pfail(p, MI_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_s)); \
  } while (0)

extern "C" const char* mi_policy_last_error(const mi_policy* p) { return p ? p->err.c_str() : g_perr.c_str(); }

extern "C" int mi_policy_create(const mi_policy_desc* d, int device, mi_policy** out) {
  if (!d || !out) return pfail(nullptr, MI_ERR_ARG, "null argument");
  if (d->state_size < 1 || d->action_size < 1 || d->action_size > 6 || d->hidden1 < 1 || d->hidden2 < 1)
    return pfail(nullptr, MI_ERR_ARG, "unsupported policy sizes (action_size must be 1..6)");
  if (d->activation != 0 && d->activation != 1) return pfail(nullptr, MI_ERR_ARG, "activation must be 0 (ReLU) or 1 (tanh)");
  mi_policy* p = new mi_policy();
  p->d = *d; p->device = device;
  p->act = d->activation == 1 ? ACT_TANH : ACT_RELU;
  p->S = d->state_size; p->A = d->action_size; p->H1 = d->hidden1; p->H2 = d->hidden2;
  size_t o = 0;
  p->o_sigma = o; o += p->A;
  p->o_w1 = o; o += (size_t)p->H1 * p->S;
  p->o_b1 = o; o += p->H1;
  p->o_w2 = o; o += (size_t)p->H2 * p->H1;
  p->o_b2 = o; o += p->H2;
  p->o_w3 = o; o += (size_t)p->A * p->H2;
  p->o_b3 = o; o += p->A;
  p->P = o;
  *out = p;
  return MI_OK;
}
extern "C" void mi_policy_destroy(mi_policy* p) {
  if (p && p->fold_counters) (void)hipFree(p->fold_counters);
  delete p;
}
extern "C" int mi_policy_param_count(const mi_policy* p, size_t* n) {
  if (!p || !n) return MI_ERR_ARG;
  *n = p->P;
  return MI_OK;
}

struct Acts { float *h1, *h2, *mu; };        // post-ReLU hidden activations (mask = h > 0) and the mean
struct PBump {
  char* base; size_t off;
  float* f(size_t n) { off = align_up(off, 256); float* r = base ? reinterpret_cast<float*>(base + off) : nullptr; off += n * 4; return r; }
};

static hipError_t dense_fwd(hipStream_t st, int T, int B, int I, int O, const float* x0, const float* w0, size_t ws0,
                            const float* x1, const float* w1, size_t ws1, const float* bias, size_t bs, const float* mask,
                            int act, float* y) {
  DenseArgs a{};
  a.x[0] = x0; a.w[0] = w0; a.wstride[0] = ws0; a.x[1] = x1; a.w[1] = w1; a.wstride[1] = ws1;
  a.bias = bias; a.bstride = bs; a.mask = mask; a.y = y; a.B = B; a.I = I; a.O = O; a.nterms = x1 ? 2 : 1; a.act = act;
  launch_dense<true>(st, T, a);
  return hipGetLastError();
}
static hipError_t dense_bwd_x(hipStream_t st, int T, int B, int I, int O, const float* dy0, const float* w0, size_t ws0,
                              const float* dy1, const float* w1, size_t ws1, const float* mask, int act, float* dx,
                              float* dx_pre = nullptr, const float* hd = nullptr, const float* dpre = nullptr) {
  DenseArgs a{};
  a.x[0] = dy0; a.w[0] = w0; a.wstride[0] = ws0; a.x[1] = dy1; a.w[1] = w1; a.wstride[1] = ws1;
  a.mask = mask; a.act = act; a.y = dx; a.ypre = dx_pre; a.hd = hd; a.dpre = dpre;
  a.B = B; a.I = I; a.O = O; a.nterms = dy1 ? 2 : 1;
  launch_dense<false>(st, T, a);
  return hipGetLastError();
}
static hipError_t dense_bwd_w(hipStream_t st, int T, int B, int I, int O, const float* dy0, const float* x0, const float* dy1,
                              const float* x1, float* dw, float* db, size_t gs) {
  DenseWArgs a{};
  a.dy[0] = dy0; a.x[0] = x0; a.dy[1] = dy1; a.x[1] = x1; a.dw = dw; a.db = db; a.gstride = gs;
  a.B = B; a.I = I; a.O = O; a.nterms = dy1 ? 2 : 1;
  hipLaunchKernelGGL(dense_wgrad_mfma_kernel, dim3(ceil_div(O, 32) * ceil_div(I + 1, 32), T), dim3(512), 0, st, a);
  return hipGetLastError();
}

// MLP forward on [T][B][S]; theta with per-task stride ts (0 = shared)
static int mlp_forward(mi_policy* p, hipStream_t st, int T, int B, const float* x, const float* th, size_t ts, Acts& a) {
  PCHK(p, dense_fwd(st, T, B, p->S, p->H1, x, th + p->o_w1, ts, nullptr, nullptr, 0, th + p->o_b1, ts, nullptr, p->act, a.h1));
  PCHK(p, dense_fwd(st, T, B, p->H1, p->H2, a.h1, th + p->o_w2, ts, nullptr, nullptr, 0, th + p->o_b2, ts, nullptr, p->act, a.h2));
  PCHK(p, dense_fwd(st, T, B, p->H2, p->A, a.h2, th + p->o_w3, ts, nullptr, nullptr, 0, th + p->o_b3, ts, nullptr, ACT_NONE, a.mu));
  return MI_OK;
}
// MLP backward from dmu: grads into g [T][P] (sigma slot untouched); scratch d2 [T][B][H2], d1 [T][B][H1] keep dz2 / dz1;
// pre2 / pre1 (optional) keep the cotangents w.r.t. h2 / h1 before the phi' factor (a tanh HVP needs them).
// head_only: only W3 / b3 get gradients (ANIL inner loop with the body under no_grad, rl.py:381-382, policies.py:100-106).
static int mlp_backward(mi_policy* p, hipStream_t st, int T, int B, const float* x, const float* th, size_t ts, const Acts& a,
                        const float* dmu, float* d2, float* d1, float* g, float* pre2 = nullptr, float* pre1 = nullptr,
                        bool head_only = false) {
  const size_t P = p->P;
  PCHK(p, dense_bwd_w(st, T, B, p->H2, p->A, dmu, a.h2, nullptr, nullptr, g + p->o_w3, g + p->o_b3, P));
  if (head_only) return MI_OK;
  PCHK(p, dense_bwd_x(st, T, B, p->H2, p->A, dmu, th + p->o_w3, ts, nullptr, nullptr, 0, a.h2, p->act, d2, pre2));
  PCHK(p, dense_bwd_w(st, T, B, p->H1, p->H2, d2, a.h1, nullptr, nullptr, g + p->o_w2, g + p->o_b2, P));
  PCHK(p, dense_bwd_x(st, T, B, p->H1, p->H2, d2, th + p->o_w2, ts, nullptr, nullptr, 0, a.h1, p->act, d1, pre1));
  PCHK(p, dense_bwd_w(st, T, B, p->S, p->H1, d1, x, nullptr, nullptr, g + p->o_w1, g + p->o_b1, P));
  return MI_OK;
}
// tangent forward: direction v [T][P] (per task), primal acts a -> tangent acts ad (h1d, h2d, mud)
static int mlp_tangent_forward(mi_policy* p, hipStream_t st, int T, int B, const float* x, const float* th, size_t ts,
                               const Acts& a, const float* v, Acts& ad) {
  const size_t P = p->P;
  PCHK(p, dense_fwd(st, T, B, p->S, p->H1, x, v + p->o_w1, P, nullptr, nullptr, 0, v + p->o_b1, P, a.h1, p->act, ad.h1));
  PCHK(p, dense_fwd(st, T, B, p->H1, p->H2, a.h1, v + p->o_w2, P, ad.h1, th + p->o_w2, ts, v + p->o_b2, P, a.h2, p->act, ad.h2));
  PCHK(p, dense_fwd(st, T, B, p->H2, p->A, a.h2, v + p->o_w3, P, ad.h2, th + p->o_w3, ts, v + p->o_b3, P, nullptr, ACT_NONE, ad.mu));
  return MI_OK;
}
// tangent backward: R{grads} into hv [T][P]; needs primal cotangents dmu, da2 (d2), da1 (d1), tangent acts ad, R{dmu} = rdmu.
static int mlp_tangent_backward(mi_policy* p, hipStream_t st, int T, int B, const float* x, const float* th, size_t ts,
                                const Acts& a, const Acts& ad, const float* v, const float* dmu, const float* d2,
                                const float* d1, const float* pre2, const float* pre1, const float* rdmu, float* r2, float* r1,
                                float* hv) {
  const size_t P = p->P;
  const bool th2 = p->act == ACT_TANH;      // the curvature of tanh adds  -2 h hdot dh  to R{dz}
  PCHK(p, dense_bwd_w(st, T, B, p->H2, p->A, rdmu, a.h2, dmu, ad.h2, hv + p->o_w3, hv + p->o_b3, P));
  PCHK(p, dense_bwd_x(st, T, B, p->H2, p->A, rdmu, th + p->o_w3, ts, dmu, v + p->o_w3, P, a.h2, p->act, r2, nullptr,
                      th2 ? ad.h2 : nullptr, th2 ? pre2 : nullptr));
  PCHK(p, dense_bwd_w(st, T, B, p->H1, p->H2, r2, a.h1, d2, ad.h1, hv + p->o_w2, hv + p->o_b2, P));
  PCHK(p, dense_bwd_x(st, T, B, p->H1, p->H2, r2, th + p->o_w2, ts, d2, v + p->o_w2, P, a.h1, p->act, r1, nullptr,
                      th2 ? ad.h1 : nullptr, th2 ? pre1 : nullptr));
  PCHK(p, dense_bwd_w(st, T, B, p->S, p->H1, r1, x, nullptr, nullptr, hv + p->o_w1, hv + p->o_b1, P));
  return MI_OK;
}

static hipError_t gauss(hipStream_t st, int T, GaussArgs& a) {
  hipLaunchKernelGGL(gauss_kernel, dim3(T), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---- workspace of a TRPO context: everything mi_trpo_fvp re-uses after mi_trpo_surrogate
struct TrpoPlan {
  Acts sa, qa, ta;                      // support acts at theta, query acts at theta', tangent acts (scratch)
  float *s_dmu, *s_d2, *s_d1, *s_coef;  // support primal cotangents
  float *s_pre2, *s_pre1;               // ... before the phi' factor (tanh HVP)
  float *q_dmu, *q_d2, *q_d1, *q_coef;
  float *rdmu, *r2, *r1;                // tangent scratch
  float *g, *thetap, *q, *hv, *u, *w, *tmpP;   // [T][P]
  float *loss_t, *kl_t;
  float* partial;                       // fused sweeps (policy_sweep.h): [T][slots][P] per-workgroup gradient partials
  int spt, spw, slots, sweep_grid;
  size_t bytes;
};
// geometry of the fused sweeps: slabs of 32 rows, a contiguous run of slabs per workgroup, one round of workgroups on 256 CUs
static void sweep_geometry(int T, int B, int& spt, int& spw, int& slots, int& grid) {
  spt = ceil_div(B, 32);
  const int total = T * (spt + 1);      // virtual slabs: a marker in front of every task's slabs (policy_sweep.hip: what entering a task costs)
  spw = ceil_div(total, 256);
  grid = ceil_div(total, spw);
  slots = ceil_div(spt + 1, spw) + 1;
}
static bool sweep_supported(const mi_policy* p) {
  return policy_sweep_supported(p->act == ACT_RELU, p->H1, p->H2, p->S, p->A);
}
static void trpo_plan(const mi_policy* p, void* ws, int T, int B, TrpoPlan& pl) {
  PBump b{reinterpret_cast<char*>(ws), 0};
  const size_t TB = (size_t)T * B, TP = (size_t)T * p->P;
  auto acts = [&](Acts& a) { a.h1 = b.f(TB * p->H1); a.h2 = b.f(TB * p->H2); a.mu = b.f(TB * p->A); };
  acts(pl.sa); acts(pl.qa); acts(pl.ta);
  pl.s_dmu = b.f(TB * p->A); pl.s_d2 = b.f(TB * p->H2); pl.s_d1 = b.f(TB * p->H1); pl.s_coef = b.f(TB);
  pl.s_pre2 = b.f(TB * p->H2); pl.s_pre1 = b.f(TB * p->H1);
  pl.q_dmu = b.f(TB * p->A); pl.q_d2 = b.f(TB * p->H2); pl.q_d1 = b.f(TB * p->H1); pl.q_coef = b.f(TB);
  pl.rdmu = b.f(TB * p->A); pl.r2 = b.f(TB * p->H2); pl.r1 = b.f(TB * p->H1);
  pl.g = b.f(TP); pl.thetap = b.f(TP); pl.q = b.f(TP); pl.hv = b.f(TP); pl.u = b.f(TP); pl.w = b.f(TP); pl.tmpP = b.f(TP);
  pl.loss_t = b.f(T); pl.kl_t = b.f(T);
  sweep_geometry(T, B, pl.spt, pl.spw, pl.slots, pl.sweep_grid);
  pl.partial = sweep_supported(p) ? b.f((size_t)T * pl.slots * (p->P + 2)) : nullptr;
  pl.bytes = align_up(b.off, 256);
}
extern "C" int mi_trpo_workspace_bytes(const mi_policy* p, int tasks, int batch, size_t* bytes) {
  if (!p || !bytes || tasks < 1 || batch < 1) return MI_ERR_ARG;
  TrpoPlan pl;
  trpo_plan(p, nullptr, tasks, batch, pl);
  *bytes = pl.bytes;
  return MI_OK;
}

// loc = MLP(state) for acting / densities (policies.py:49-52); theta [P] shared (tstride 0) or per task (tstride = P).
extern "C" int mi_policy_forward(mi_policy* p, void* stream, const float* theta, size_t tstride, const float* states, int tasks,
                                 int batch, float* loc_out, void* workspace, size_t workspace_bytes) {
  if (!p || !theta || !states || !loc_out || !workspace) return pfail(p, MI_ERR_ARG, "null argument");
  PBump b{reinterpret_cast<char*>(workspace), 0};
  Acts a;
  a.h1 = b.f((size_t)tasks * batch * p->H1); a.h2 = b.f((size_t)tasks * batch * p->H2); a.mu = loc_out;
  if (b.off > workspace_bytes) return pfail(p, MI_ERR_WORKSPACE, "workspace too small");
  return mlp_forward(p, reinterpret_cast<hipStream_t>(stream), tasks, batch, states, theta, tstride, a);
}

static int g_policy_fused_fvp = 1;
static unsigned long long* g_sweep_stamps = nullptr;
extern "C" int mi_debug_policy_sweep_stamps(void* buf) { g_sweep_stamps = reinterpret_cast<unsigned long long*>(buf); return MI_OK; }
// 1 (default): the Fisher-vector product of a supported policy runs as three fused sweeps + three folds (policy_sweep.h);
// 0: the per-layer path (ablation / tests).
extern "C" int mi_policy_set_fused_fvp(int on) { g_policy_fused_fvp = on ? 1 : 0; return MI_OK; }

static SweepArgs sweep_base(const mi_policy* p, const TrpoPlan& pl, int T, int B) {
  SweepArgs a{};
  a.T = T; a.B = B; a.S = p->S; a.A = p->A; a.spt = pl.spt; a.spw = pl.spw; a.slots = pl.slots; a.partial = pl.partial;
  a.pitch = (int)p->P + 2;
  a.o_sigma = (int)p->o_sigma; a.o_w1 = (int)p->o_w1; a.o_b1 = (int)p->o_b1; a.o_w2 = (int)p->o_w2; a.o_b2 = (int)p->o_b2;
  a.o_w3 = (int)p->o_w3; a.o_b3 = (int)p->o_b3; a.P = (int)p->P;
  a.stamps = g_sweep_stamps;
  return a;
}

// inner-loss HVP on the cached support pass: hv = H_t v for every task (v, hv: [T][P])
static int support_hvp(mi_policy* p, hipStream_t st, TrpoPlan& pl, int T, int B, const float* theta, const float* s_states,
                       const float* s_actions, const int32_t* s_count, const float* v, float* hv) {
  PCHK(p, hipMemsetAsync(hv, 0, (size_t)T * p->P * sizeof(float), st));
  int rc = mlp_tangent_forward(p, st, T, B, s_states, theta, 0, pl.sa, v, pl.ta);
  if (rc) return rc;
  GaussArgs ga{};
  ga.mu = pl.sa.mu; ga.mud = pl.ta.mu; ga.rho = theta + p->o_sigma; ga.rstride = 0; ga.rhod = v + p->o_sigma; ga.vstride = p->P;
  ga.act = s_actions; ga.count = s_count; ga.coef = pl.s_coef; ga.dmu = pl.rdmu; ga.drho = hv + p->o_sigma; ga.gstride = p->P;
  ga.B = B; ga.A = p->A; ga.mode = G_TANGENT;
  PCHK(p, gauss(st, T, ga));
  return mlp_tangent_backward(p, st, T, B, s_states, theta, 0, pl.sa, pl.ta, v, pl.s_dmu, pl.s_d2, pl.s_d1, pl.s_pre2, pl.s_pre1, pl.rdmu,
                              pl.r2, pl.r1, hv);
}

// trpo_update (rl.py:361-374) for a meta-batch: theta_out[t] = theta[t] - lr * grad_t(-mean(logp * adv)).  loss_out [T].
extern "C" int mi_policy_adapt(mi_policy* p, void* stream, const float* theta, size_t tstride, const float* states,
                               const float* actions, const float* adv, const int32_t* count, int tasks, int batch, float lr,
                               int head_only, float* theta_out, float* loss_out, void* workspace, size_t workspace_bytes) {
  if (!p || !theta || !states || !actions || !adv || !theta_out || !workspace) return pfail(p, MI_ERR_ARG, "null argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  TrpoPlan pl;
  trpo_plan(p, workspace, tasks, batch, pl);
  if (pl.bytes > workspace_bytes) return pfail(p, MI_ERR_WORKSPACE, "workspace too small: need " + std::to_string(pl.bytes));
  int rc = mlp_forward(p, st, tasks, batch, states, theta, tstride, pl.sa);
  if (rc) return rc;
  PCHK(p, hipMemsetAsync(pl.g, 0, (size_t)tasks * p->P * sizeof(float), st));
  GaussArgs ga{};
  ga.mu = pl.sa.mu; ga.rho = theta + p->o_sigma; ga.rstride = tstride; ga.act = actions; ga.adv = adv; ga.count = count;
  ga.coef = pl.s_coef; ga.dmu = pl.s_dmu; ga.drho = pl.g + p->o_sigma; ga.gstride = p->P; ga.loss = loss_out ? loss_out : pl.loss_t;
  ga.B = batch; ga.A = p->A; ga.mode = G_A2C;
  PCHK(p, gauss(st, tasks, ga));
  rc = mlp_backward(p, st, tasks, batch, states, theta, tstride, pl.sa, pl.s_dmu, pl.s_d2, pl.s_d1, pl.g, nullptr, nullptr, head_only != 0);
  if (rc) return rc;
  hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)p->P, 256), tasks), dim3(256), 0, st, theta, tstride, pl.g, lr,
                     (int)p->P, theta_out);
  PCHK(p, hipGetLastError());
  return MI_OK;
}

// meta_surrogate_loss (rl.py:441-473) at theta for `tasks` tasks with ONE second-order inner step on the support replay, and
// optionally its gradient (rl.py:413-416).  Leaves everything mi_trpo_fvp needs in `workspace`.
extern "C" int mi_trpo_surrogate(mi_policy* p, void* stream, const float* theta, const float* s_states, const float* s_actions,
                                 const float* s_adv, const int32_t* s_count, const float* q_states, const float* q_actions,
                                 const float* q_adv, const int32_t* q_count, const float* old_loc, const float* old_scale,
                                 int tasks, int batch, float inner_lr, float* loss_out, float* kl_out, float* grad_out,
                                 void* workspace, size_t workspace_bytes) {
  if (!p || !theta || !s_states || !s_actions || !s_adv || !q_states || !q_actions || !q_adv || !old_loc || !old_scale ||
      !loss_out || !kl_out || !workspace)
    return pfail(p, MI_ERR_ARG, "null argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int T = tasks, B = batch;
  const size_t P = p->P;
  TrpoPlan pl;
  trpo_plan(p, workspace, T, B, pl);
  if (pl.bytes > workspace_bytes) return pfail(p, MI_ERR_WORKSPACE, "workspace too small: need " + std::to_string(pl.bytes));
  int rc = MI_OK;
  const bool fused = g_policy_fused_fvp && sweep_supported(p) && pl.partial;
  if (fused) {
    // inner step and query pass as two PRIMAL sweeps (forward + loss + backward of a pass in one launch, leaving the activations and
    // cotangents the Hessian-vector / Fisher sweeps read) instead of ~22 per-layer launches
    SweepArgs sp = sweep_base(p, pl, T, B);
    sp.x = s_states; sp.act = s_actions; sp.adv = s_adv; sp.count = s_count; sp.theta = theta; sp.tstride = 0; sp.surrogate = 0;
    sp.h1_out = pl.sa.h1; sp.h2_out = pl.sa.h2; sp.mu_out = pl.sa.mu; sp.dmu_out = pl.s_dmu; sp.d2_out = pl.s_d2; sp.coef_out = pl.s_coef;
    PCHK(p, launch_policy_sweep(st, sp, pl.sweep_grid, SW_PRIMAL));
    FoldArgs f{};
    f.partial = pl.partial; f.slots = pl.slots; f.spt = pl.spt; f.spw = pl.spw; f.T = T; f.P = (int)P; f.pitch = (int)P + 2; f.lr = inner_lr;
    f.o_sigma = (int)p->o_sigma; f.A = p->A;
    f.mode = 0; f.v = theta; f.out = pl.thetap;                     // theta'_t = theta - lr g_t
    PCHK(p, launch_policy_sweep_fold(st, f, T));
    SweepArgs sq = sweep_base(p, pl, T, B);
    sq.x = q_states; sq.act = q_actions; sq.adv = q_adv; sq.count = q_count; sq.theta = pl.thetap; sq.tstride = P; sq.surrogate = 1;
    sq.old_loc = old_loc; sq.old_scale = old_scale; sq.fwd_only = grad_out ? 0 : 1;
    sq.h1_out = pl.qa.h1; sq.h2_out = pl.qa.h2; sq.mu_out = pl.qa.mu;
    PCHK(p, launch_policy_sweep(st, sq, pl.sweep_grid, SW_PRIMAL));
    f.mode = 3; f.out = grad_out ? pl.q : nullptr; f.loss_t = pl.loss_t; f.kl_t = pl.kl_t;      // q_t = grad S_t(theta'_t); per-task loss / KL
    PCHK(p, launch_policy_sweep_fold(st, f, T));
    hipLaunchKernelGGL(mean_tasks_kernel, dim3(1), dim3(64), 0, st, pl.loss_t, T, 1, 1.f / (float)T, (const float*)nullptr, 0.f, loss_out);
    hipLaunchKernelGGL(mean_tasks_kernel, dim3(1), dim3(64), 0, st, pl.kl_t, T, 1, 1.f / (float)T, (const float*)nullptr, 0.f, kl_out);
    PCHK(p, hipGetLastError());
    if (!grad_out) return MI_OK;
  } else {
    // inner step on support at theta (shared)
    rc = mlp_forward(p, st, T, B, s_states, theta, 0, pl.sa);
    if (rc) return rc;
    PCHK(p, hipMemsetAsync(pl.g, 0, (size_t)T * P * sizeof(float), st));
    GaussArgs ga{};
    ga.mu = pl.sa.mu; ga.rho = theta + p->o_sigma; ga.rstride = 0; ga.act = s_actions; ga.adv = s_adv; ga.count = s_count;
    ga.coef = pl.s_coef; ga.dmu = pl.s_dmu; ga.drho = pl.g + p->o_sigma; ga.gstride = P; ga.loss = pl.loss_t;
    ga.B = B; ga.A = p->A; ga.mode = G_A2C;
    PCHK(p, gauss(st, T, ga));
    rc = mlp_backward(p, st, T, B, s_states, theta, 0, pl.sa, pl.s_dmu, pl.s_d2, pl.s_d1, pl.g, pl.s_pre2, pl.s_pre1);
    if (rc) return rc;
    hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, theta, (size_t)0, pl.g, inner_lr, (int)P,
                       pl.thetap);
    PCHK(p, hipGetLastError());
    // query at theta'
    rc = mlp_forward(p, st, T, B, q_states, pl.thetap, P, pl.qa);
    if (rc) return rc;
    PCHK(p, hipMemsetAsync(pl.q, 0, (size_t)T * P * sizeof(float), st));
    GaussArgs gq{};
    gq.mu = pl.qa.mu; gq.rho = pl.thetap + p->o_sigma; gq.rstride = P; gq.act = q_actions; gq.adv = q_adv; gq.count = q_count;
    gq.old_loc = old_loc; gq.old_scale = old_scale; gq.coef = pl.q_coef; gq.dmu = pl.q_dmu; gq.drho = pl.q + p->o_sigma;
    gq.gstride = P; gq.loss = pl.loss_t; gq.kl = pl.kl_t; gq.B = B; gq.A = p->A; gq.mode = G_SURROGATE;
    PCHK(p, gauss(st, T, gq));
    hipLaunchKernelGGL(mean_tasks_kernel, dim3(1), dim3(64), 0, st, pl.loss_t, T, 1, 1.f / (float)T, (const float*)nullptr, 0.f, loss_out);
    hipLaunchKernelGGL(mean_tasks_kernel, dim3(1), dim3(64), 0, st, pl.kl_t, T, 1, 1.f / (float)T, (const float*)nullptr, 0.f, kl_out);
    PCHK(p, hipGetLastError());
    if (!grad_out) return MI_OK;
    rc = mlp_backward(p, st, T, B, q_states, pl.thetap, P, pl.qa, pl.q_dmu, pl.q_d2, pl.q_d1, pl.q);   // q_t = grad S_t(theta'_t)
    if (rc) return rc;
  }
  if (g_policy_fused_fvp && sweep_supported(p) && pl.partial) {
    // (I - lr H_t) q_t as ONE fused sweep over the support pass (direction q_t per task) + fold, instead of ~10 per-layer launches
    SweepArgs hs = sweep_base(p, pl, T, B);
    hs.x = s_states; hs.act = s_actions; hs.h1 = pl.sa.h1; hs.h2 = pl.sa.h2; hs.mu = pl.sa.mu; hs.coef = pl.s_coef; hs.dmu = pl.s_dmu;
    hs.d2 = pl.s_d2; hs.count = s_count; hs.theta = theta; hs.tstride = 0; hs.dir = pl.q; hs.dstride = P;
    PCHK(p, launch_policy_sweep(st, hs, pl.sweep_grid, SW_HVP));
    FoldArgs f{};
    f.partial = pl.partial; f.slots = pl.slots; f.spt = pl.spt; f.spw = pl.spw; f.T = T; f.P = (int)P; f.pitch = (int)P + 2; f.lr = inner_lr;
    f.o_sigma = (int)p->o_sigma; f.A = p->A; f.mode = 2; f.out = pl.tmpP; f.w = pl.q;
    PCHK(p, launch_policy_sweep_fold(st, f, T));
  } else {
    rc = support_hvp(p, st, pl, T, B, theta, s_states, s_actions, s_count, pl.q, pl.hv);
    if (rc) return rc;
    hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, pl.q, P, pl.hv, inner_lr, (int)P, pl.tmpP);
  }
  hipLaunchKernelGGL(mean_tasks_kernel, dim3(ceil_div((int)P, 256)), dim3(256), 0, st, pl.tmpP, T, (int)P, 1.f / (float)T,
                     (const float*)nullptr, 0.f, grad_out);
  PCHK(p, hipGetLastError());
  return MI_OK;
}

// Fvp(v) = mean_t (I - lr H_t) F_t (I - lr H_t) v + damping v   (cherry trpo.hessian_vector_product of the mean KL at the point
// where the adapted policy equals the old policy, rl.py:417).  Must follow mi_trpo_surrogate on the same workspace/arguments.
// One recurrence of cherry.algorithms.trpo.conjugate_gradient (reference rl.py:418) on device vectors, in fp64, as ONE launch:
//   alpha = rr_old / (p . Ap + eps);  x += alpha p;  r -= alpha Ap;  rr_new = r . r;  p = r + (rr_new / rr_old) p
// (a dozen ATen launches and two rocBLAS dots per iteration otherwise: ~0.1 ms of each ~1 ms iteration).  One workgroup: the vectors
// hold the policy's ~10^4 parameters; both dot products are block reductions in a fixed order.
__global__ __launch_bounds__(1024) void cg_update_kernel(double* __restrict__ x, double* __restrict__ r, double* __restrict__ p,
                                                         const float* __restrict__ ap, double* __restrict__ rr, float* __restrict__ p32,
                                                         size_t n, double eps, double tol) {
  // tol >= 0: the reference's `if r_dot_new < tol: break` is taken ON THE DEVICE -- rr[2] latches "converged" and every later
  // recurrence of this solve is a no-op, so the host loop needs no synchronisation per iteration (x is exactly the x of the break)
  if (tol >= 0.0 && rr[2] != 0.0) return;
  __shared__ double red[16];
  __shared__ double bcast;
  const int tid = threadIdx.x;
  auto block_sum = [&](double v) -> double {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int k = 0; k < 16; ++k) s += red[k];
      bcast = s;
    }
    __syncthreads();
    return bcast;
  };
  const double rr_old = rr[0];
  double s = 0.0;
  for (size_t i = tid; i < n; i += 1024) s += p[i] * (double)ap[i];
  const double alpha = rr_old / (block_sum(s) + eps);
  double q = 0.0;
  for (size_t i = tid; i < n; i += 1024) {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * (double)ap[i];
    r[i] = ri;
    q += ri * ri;
  }
  const double rr_new = block_sum(q);
  const double beta = rr_new / rr_old;
  for (size_t i = tid; i < n; i += 1024) {
    const double pi = r[i] + beta * p[i];
    p[i] = pi;
    p32[i] = (float)pi;
  }
  if (tid == 0) { rr[0] = rr_new; rr[1] = alpha; if (tol >= 0.0 && rr_new < tol) rr[2] = 1.0; }
}

// The start of a solve as one launch: r = p = b (fp64), x = 0, p32 = b, rr = (b . b, 0, 0) -- seven tensor-library launches otherwise.
__global__ __launch_bounds__(1024) void cg_init_kernel(const float* __restrict__ b, double* __restrict__ x, double* __restrict__ r,
                                                       double* __restrict__ p, float* __restrict__ p32, double* __restrict__ rr, size_t n) {
  __shared__ double red[16];
  const int tid = threadIdx.x;
  double q = 0.0;
  for (size_t i = tid; i < n; i += 1024) {
    const float bf = b[i];
    const double bi = (double)bf;
    x[i] = 0.0; r[i] = bi; p[i] = bi; p32[i] = bf;
    q += bi * bi;
  }
  q = wave_sum(q);
  if ((tid & 63) == 0) red[tid >> 6] = q;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int k = 0; k < 16; ++k) s += red[k];
    rr[0] = s; rr[1] = 0.0; rr[2] = 0.0;
  }
}
extern "C" int mi_cg_init(void* stream, const float* b, double* x, double* r, double* p, float* p32, double* rr, size_t n) {
  if (!b || !x || !r || !p || !p32 || !rr || n == 0) return MI_ERR_ARG;
  hipLaunchKernelGGL(cg_init_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), b, x, r, p, p32, rr, n);
  return hipGetLastError() == hipSuccess ? MI_OK : MI_ERR_HIP;
}

// The step of a trust-region update from the solve's direction s and F s (reference rl.py:419-421): shs = 0.5 s . F s (summed in fp64),
// lagrange = sqrt(shs / max_kl), out = s / lagrange -- five tensor-library launches otherwise.  A negative shs gives NaN, as there.
__global__ __launch_bounds__(1024) void trpo_scale_step_kernel(const float* __restrict__ s, const float* __restrict__ fs, size_t n, float max_kl,
                                                               float* __restrict__ out, float* __restrict__ lagrange) {
  __shared__ double red[16];
  __shared__ float lm_s;
  const int tid = threadIdx.x;
  double q = 0.0;
  for (size_t i = tid; i < n; i += 1024) q += (double)s[i] * (double)fs[i];
  q = wave_sum(q);
  if ((tid & 63) == 0) red[tid >> 6] = q;
  __syncthreads();
  if (tid == 0) {
    double d = 0.0;
    for (int k = 0; k < 16; ++k) d += red[k];
    const float shs = 0.5f * (float)d;
    lm_s = sqrtf(shs / max_kl);
    if (lagrange) *lagrange = lm_s;
  }
  __syncthreads();
  const float lm = lm_s;
  for (size_t i = tid; i < n; i += 1024) out[i] = s[i] / lm;
}
extern "C" int mi_trpo_scale_step(void* stream, const float* step, const float* fstep, size_t n, float max_kl, float* out, float* lagrange_out) {
  if (!step || !fstep || !out || n == 0 || !(max_kl > 0.f)) return MI_ERR_ARG;
  hipLaunchKernelGGL(trpo_scale_step_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), step, fstep, n, max_kl, out, lagrange_out);
  return hipGetLastError() == hipSuccess ? MI_OK : MI_ERR_HIP;
}

extern "C" int mi_cg_update(void* stream, double* x, double* r, double* p, const float* ap, double* rr, float* p32, size_t n, double eps) {
  if (!x || !r || !p || !ap || !rr || !p32 || n == 0) return MI_ERR_ARG;
  hipLaunchKernelGGL(cg_update_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), x, r, p, ap, rr, p32, n, eps, -1.0);
  return hipGetLastError() == hipSuccess ? MI_OK : MI_ERR_HIP;
}
extern "C" int mi_cg_update_checked(void* stream, double* x, double* r, double* p, const float* ap, double* rr, float* p32, size_t n, double eps,
                                    double tol) {
  if (!x || !r || !p || !ap || !rr || !p32 || n == 0 || tol < 0.0) return MI_ERR_ARG;
  hipLaunchKernelGGL(cg_update_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), x, r, p, ap, rr, p32, n, eps, tol);
  return hipGetLastError() == hipSuccess ? MI_OK : MI_ERR_HIP;
}

static int fused_fvp(mi_policy* p, hipStream_t st, TrpoPlan& pl, int T, int B, const float* theta, const float* s_states,
                     const float* s_actions, const int32_t* s_count, const float* q_states, const int32_t* q_count, float inner_lr,
                     float damping, const float* v, float* out) {
  const int P = (int)p->P;
  SweepArgs hs = sweep_base(p, pl, T, B);              // H_t over the support pass at theta
  hs.x = s_states; hs.act = s_actions; hs.h1 = pl.sa.h1; hs.h2 = pl.sa.h2; hs.mu = pl.sa.mu; hs.coef = pl.s_coef; hs.dmu = pl.s_dmu;
  hs.d2 = pl.s_d2; hs.count = s_count; hs.theta = theta; hs.tstride = 0;
  FoldArgs f{};
  f.partial = pl.partial; f.slots = pl.slots; f.spt = pl.spt; f.spw = pl.spw; f.T = T; f.P = P; f.pitch = P + 2; f.v = v; f.lr = inner_lr;
  f.damping = damping; f.o_sigma = (int)p->o_sigma; f.A = p->A;
  // A: u_t = v - lr H_t v
  hs.dir = v; hs.dstride = 0;
  PCHK(p, launch_policy_sweep(st, hs, pl.sweep_grid, SW_HVP));
  f.mode = 0; f.out = pl.u;
  PCHK(p, launch_policy_sweep_fold(st, f, T));
  // B: w_t = F_t u_t over the query pass at theta'_t
  SweepArgs fs = sweep_base(p, pl, T, B);
  fs.x = q_states; fs.h1 = pl.qa.h1; fs.h2 = pl.qa.h2; fs.count = q_count; fs.theta = pl.thetap; fs.tstride = P; fs.dir = pl.u; fs.dstride = P;
  PCHK(p, launch_policy_sweep(st, fs, pl.sweep_grid, SW_FISHER));
  f.mode = 1; f.out = pl.w; f.thetap = pl.thetap; f.u = pl.u;
  PCHK(p, launch_policy_sweep_fold(st, f, T));
  // C: out = mean_t (w_t - lr H_t w_t) + damping v
  hs.dir = pl.w; hs.dstride = P;
  PCHK(p, launch_policy_sweep(st, hs, pl.sweep_grid, SW_HVP));
  f.mode = 2; f.out = pl.tmpP; f.w = pl.w;
  if (!p->fold_counters) {               // (once per policy object: the fold leaves the counters zero again)
    const size_t nb = (size_t)ceil_div(P, 256) * sizeof(unsigned);
    if (hipMalloc(reinterpret_cast<void**>(&p->fold_counters), nb) != hipSuccess) { p->fold_counters = nullptr; (void)hipGetLastError(); }
    else PCHK(p, hipMemsetAsync(p->fold_counters, 0, nb, st));
  }
  if (p->fold_counters) {                // the mean over tasks + damping v by the last workgroup of every parameter block: no launch of its own
    // The protocol needs the counters at zero when the fold starts; its last workgroup leaves them at zero.  A fold whose launch was
    // refused leaves the flag set, and the next product re-zeroes (stream-ordered, 4 * ceil(P / 256) bytes) instead of waiting for a
    // "last" workgroup that can no longer exist.  One stream per mi_policy at a time (include/mi_maml.h).
    if (p->fold_dirty) PCHK(p, hipMemsetAsync(p->fold_counters, 0, (size_t)ceil_div(P, 256) * sizeof(unsigned), st));
    p->fold_dirty = true;
    f.counter = p->fold_counters; f.mean_out = out; f.inv_T = 1.f / (float)T;
    PCHK(p, launch_policy_sweep_fold(st, f, T));
    p->fold_dirty = false;
    return MI_OK;
  }
  PCHK(p, launch_policy_sweep_fold(st, f, T));
  hipLaunchKernelGGL(mean_tasks_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, st, pl.tmpP, T, P, 1.f / (float)T, v, damping, out);
  PCHK(p, hipGetLastError());
  return MI_OK;
}

extern "C" int mi_trpo_fvp(mi_policy* p, void* stream, const float* theta, const float* s_states, const float* s_actions,
                           const int32_t* s_count, const float* q_states, const int32_t* q_count, int tasks, int batch,
                           float inner_lr, float damping, const float* v, float* out, void* workspace, size_t workspace_bytes) {
  if (!p || !theta || !v || !out || !workspace) return pfail(p, MI_ERR_ARG, "null argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int T = tasks, B = batch;
  const size_t P = p->P;
  TrpoPlan pl;
  trpo_plan(p, workspace, T, B, pl);
  if (pl.bytes > workspace_bytes) return pfail(p, MI_ERR_WORKSPACE, "workspace too small");
  if (g_policy_fused_fvp && sweep_supported(p) && pl.partial)
    return fused_fvp(p, st, pl, T, B, theta, s_states, s_actions, s_count, q_states, q_count, inner_lr, damping, v, out);
  // u_t = v - lr H_t v   (v broadcast to every task)
  hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, v, (size_t)0, pl.tmpP, 0.f, (int)P, pl.u);
  PCHK(p, hipGetLastError());
  int rc = support_hvp(p, st, pl, T, B, theta, s_states, s_actions, s_count, pl.u, pl.hv);
  if (rc) return rc;
  hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, pl.u, P, pl.hv, inner_lr, (int)P, pl.u);
  PCHK(p, hipGetLastError());
  // w_t = F_t u_t : JVP on the query pass at theta', Gaussian Fisher cotangent, VJP
  rc = mlp_tangent_forward(p, st, T, B, q_states, pl.thetap, P, pl.qa, pl.u, pl.ta);
  if (rc) return rc;
  PCHK(p, hipMemsetAsync(pl.w, 0, (size_t)T * P * sizeof(float), st));
  GaussArgs gf{};
  gf.mu = pl.qa.mu; gf.mud = pl.ta.mu; gf.rho = pl.thetap + p->o_sigma; gf.rstride = P; gf.rhod = pl.u + p->o_sigma; gf.vstride = P;
  gf.count = q_count; gf.dmu = pl.rdmu; gf.drho = pl.w + p->o_sigma; gf.gstride = P; gf.B = B; gf.A = p->A; gf.mode = G_FISHER;
  PCHK(p, gauss(st, T, gf));
  rc = mlp_backward(p, st, T, B, q_states, pl.thetap, P, pl.qa, pl.rdmu, pl.r2, pl.r1, pl.w);
  if (rc) return rc;
  // out = mean_t (w_t - lr H_t w_t) + damping v
  rc = support_hvp(p, st, pl, T, B, theta, s_states, s_actions, s_count, pl.w, pl.hv);
  if (rc) return rc;
  hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, pl.w, P, pl.hv, inner_lr, (int)P, pl.tmpP);
  hipLaunchKernelGGL(mean_tasks_kernel, dim3(ceil_div((int)P, 256)), dim3(256), 0, st, pl.tmpP, T, (int)P, 1.f / (float)T, v, damping, out);
  PCHK(p, hipGetLastError());
  return MI_OK;
}

// =====================================================================================================================
// MAML inner loop of the policy with a general number of updates and the VPG / PPO losses (reference core_functions/rl.py:
// vpg_a2c_loss :209-228 (dice=False), fast_adapt_vpg :231-255, fast_adapt_ppo :267-318, single_ppo_update :321-337; drivers
// rl/maml_ppo.py, rl/anil_ppo.py): per task  theta_{k+1} = theta_k - lr * [head mask] grad L_k(theta_k)  for k < K on replayed
// support batches, a query loss at theta_K, and -- what `av_loss.backward()` computes through learn2learn's second-order
// `learner.adapt` -- its gradient w.r.t. theta via the adjoint recursion  lam_k = lam_{k+1} - lr H_k [mask] lam_{k+1},
// every H_k v as a forward-over-reverse sweep over the saved step-k pass (as in engine.hip for the classifier).
//
// Losses as functions of the per-sample mean log-prob lp_i:   A2C  f_i = -A_i lp_i / B                    (a2c.policy_loss)
//   PPO  f_i = -min(r_i A_i, clamp(r_i, 1-c, 1+c) A_i) / B,  r_i = exp(lp_i - lp_old_i)                    (ppo.policy_loss)
// f' feeds the backward pass, f'' (0 for A2C, = f' for an active PPO sample) the Hessian-vector product:
//   R{dL/dx} = f'' (dlp/dtheta . v) dlp/dx + f' R{dlp/dx}.   torch.min splits the gradient of ties evenly between its arguments;
// inside the clip range both arguments ARE r A, so the sum is again r A -- active = inside the range, or outside with r A the
// smaller argument.
struct Gauss2Args {
  const float* mu; const float* mud;
  const float* rho; const float* rhod; size_t rstride, vstride;
  const float* act; const float* adv; const int32_t* count;
  const float* oldlp;     // [T][B] PPO: log-prob under the parameters the epoch group started from
  float* lp_out;          // [T][B] (P_LOGP)
  float* coef;            // [T][B] f'  (written by P_PRIMAL, read by P_TANGENT)
  float* coef2;           // [T][B] f''
  float* dmu; float* drho; size_t gstride;
  float* loss;            // [T]
  float clip;
  int B, A, kind, mode, value_ratio_one;
  const float* done;      // [T][B] DiCE: 1 at the last step of every episode (cherry `dones`)
  float* scratch;         // [T][B] DiCE tangent: per-sample scratch
};
enum { P_PRIMAL = 0, P_TANGENT = 1, P_LOGP = 2 };

// DiCE objective (reference core_functions/rl.py:219-226, vpg_a2c_loss(dice=True)):
//   weights_i = (1 - done_{i-1}) / E  (weights_0 = 1 / E,  E = number of episodes = sum of dones)
//   c = weighted_cumsum(log_probs, weights):  c_i = lp_i + weights_i c_{i-1},  with the reference's Python wrap-around at i = 0
//       (values[-1] is the LAST element, still unmodified): c_0 = lp_0 + weights_0 lp_{N-1}
//   loss = a2c.policy_loss(magic_box(c), A) = -mean(exp(c - stop_gradient(c)) A):  value -mean(A), d loss / d c_i = a_i = -A_i / N,
//   d^2 loss / d c_i^2 = a_i as well.  c = M lp with M fixed by the episode boundaries, so
//       d loss / d lp = M^T a,        R{d loss / d lp} = M^T (a . (M lp_dot)).
// Episodes are independent segments of the recurrence (weights_i = 0 at an episode start): whichever thread meets a segment's
// first (last) sample runs the forward (adjoint) recurrence along it; the wrap term is a one-element fix-up.
__device__ __forceinline__ bool dice_start(const float* dn, int i) { return i == 0 || dn[i - 1] != 0.f; }
// out = M x (forward) for one task; x, out: [cnt] (may not alias)
__device__ void dice_forward(const float* dn, int cnt, float invE, const float* x, float* out) {
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
    if (!dice_start(dn, i)) continue;
    float c = x[i] + (i == 0 ? invE * x[cnt - 1] : 0.f);
    out[i] = c;
    for (int k = i + 1; k < cnt && !dice_start(dn, k); ++k) { c = x[k] + invE * c; out[k] = c; }
  }
  __syncthreads();
}
// out = M^T x (adjoint) for one task
__device__ void dice_adjoint(const float* dn, int cnt, float invE, const float* x, float* out) {
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
    if (!(i == cnt - 1 || dn[i] != 0.f)) continue;           // last sample of a segment
    float g = x[i];
    out[i] = g;
    for (int k = i; k > 0 && !dice_start(dn, k); --k) { g = x[k - 1] + invE * g; out[k - 1] = g; }
  }
  __syncthreads();
  if (threadIdx.x == 0 && cnt > 0) out[cnt - 1] += invE * out[0];          // c_0 also reads lp_{N-1}
  __syncthreads();
}

__global__ __launch_bounds__(256) void gauss2_kernel(Gauss2Args a) {
  __shared__ float red[256];
  const int t = blockIdx.x, tid = threadIdx.x;
  const int B = a.B, A = a.A, cnt = a.count ? a.count[t] : B;
  const float invB = 1.f / (float)cnt, invD = 1.f / (float)A;
  float acc[7];   // loss, drho[0..5]
#pragma unroll
  for (int k = 0; k < 7; ++k) acc[k] = 0.f;
  const bool dice = a.kind == MI_PLOSS_DICE && a.mode != P_LOGP;
  float invE = 0.f;
  if (dice) {
    // per-sample coefficients through the episode recurrences BEFORE the per-sample loop below (which then reads coef / scratch)
    const float* dn = a.done + (size_t)t * B;
    float e = 0.f;
    for (int b = tid; b < cnt; b += 256) e += dn[b] != 0.f ? 1.f : 0.f;
    red[tid] = e;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
      if (tid < s2) red[tid] += red[tid + s2];
      __syncthreads();
    }
    invE = 1.f / red[0];
    __syncthreads();
    float* cf = a.coef + (size_t)t * B;
    float* c2 = a.coef2 + (size_t)t * B;
    if (a.mode == P_PRIMAL) {
      for (int b = tid; b < B; b += 256) { c2[b] = b < cnt ? -a.adv[(size_t)t * B + b] * invB : 0.f; cf[b] = 0.f; }
      __syncthreads();
      dice_adjoint(dn, cnt, invE, c2, cf);                    // coef = M^T a
    } else {
      float* sc = a.scratch + (size_t)t * B;                  // sc <- lp_dot per sample
      for (int b = tid; b < B; b += 256) {
        const size_t ob = (size_t)t * B + b;
        float lpd = 0.f;
        for (int d = 0; d < A; ++d) {
          const float rp = a.rho[(size_t)t * a.rstride + d];
          const bool live = rp > LOG_EPS;
          const float r = fmaxf(rp, LOG_EPS), sg = expf(r), iv = 1.f / (sg * sg);
          const float rd = live ? a.rhod[(size_t)t * a.vstride + d] : 0.f;
          const float df = a.act[ob * A + d] - a.mu[ob * A + d];
          lpd += invD * (df * iv * a.mud[ob * A + d] + (df * df * iv - 1.f) * rd);
        }
        sc[b] = b < cnt ? lpd : 0.f;
      }
      __syncthreads();
      float* tmp = a.dmu + (size_t)t * B * A;                 // [B*A] >= [B]: free until the per-sample loop writes it
      dice_forward(dn, cnt, invE, sc, tmp);                   // tmp = M lp_dot
      for (int b = tid; b < cnt; b += 256) tmp[b] *= c2[b];   // a . (M lp_dot)
      __syncthreads();
      dice_adjoint(dn, cnt, invE, tmp, sc);                   // sc = R{coef}
    }
  }
  for (int b = tid; b < B; b += 256) {
    const bool valid = b < cnt;
    const size_t ob = (size_t)t * B + b;
    float lp = 0.f;
    for (int d = 0; d < A; ++d) {
      const float r = fmaxf(a.rho[(size_t)t * a.rstride + d], LOG_EPS), sg = expf(r);
      const float df = a.act[ob * A + d] - a.mu[ob * A + d];
      lp += -(df * df) / (2.f * sg * sg) - r - HALF_LOG_2PI;
    }
    lp *= invD;
    if (a.mode == P_LOGP) { a.lp_out[ob] = lp; continue; }
    if (a.mode == P_PRIMAL) {
      float c = 0.f, c2 = 0.f;
      if (valid) {
        const float ad = a.adv[ob];
        if (a.kind == MI_PLOSS_A2C) {
          c = -ad * invB;
          acc[0] += a.value_ratio_one ? c : c * lp;          // PPO's validation loss: ratio == 1 exactly, value -mean(A)
        } else if (a.kind == MI_PLOSS_DICE) {
          c = a.coef[ob];                                     // (M^T a)_b, computed above
          c2 = a.coef2[ob];
          acc[0] += -ad * invB;                               // magic_box == 1: value -mean(A)
        } else {
          const float ratio = expf(lp - a.oldlp[ob]);
          const float o1 = ratio * ad, o2 = fminf(fmaxf(ratio, 1.f - a.clip), 1.f + a.clip) * ad;
          acc[0] += -fminf(o1, o2) * invB;
          const bool inside = ratio >= 1.f - a.clip && ratio <= 1.f + a.clip;
          const bool active = inside || o1 < o2;
          c = active ? -o1 * invB : 0.f;
          c2 = c;
        }
      }
      a.coef[ob] = c;
      a.coef2[ob] = c2;
      for (int d = 0; d < A; ++d) {
        const float rp = a.rho[(size_t)t * a.rstride + d];
        const float r = fmaxf(rp, LOG_EPS), sg = expf(r), iv = 1.f / (sg * sg);
        const float df = a.act[ob * A + d] - a.mu[ob * A + d];
        a.dmu[ob * A + d] = c * invD * df * iv;
        if (rp > LOG_EPS) acc[1 + d] += c * invD * (df * df * iv - 1.f);
      }
    } else {   // P_TANGENT
      const float c = valid ? a.coef[ob] : 0.f, c2 = valid ? a.coef2[ob] : 0.f;
      float lpd = 0.f;                                       // tangent of lp along (mud, rhod)
      for (int d = 0; d < A; ++d) {
        const float rp = a.rho[(size_t)t * a.rstride + d];
        const bool live = rp > LOG_EPS;
        const float r = fmaxf(rp, LOG_EPS), sg = expf(r), iv = 1.f / (sg * sg);
        const float rd = live ? a.rhod[(size_t)t * a.vstride + d] : 0.f;
        const float df = a.act[ob * A + d] - a.mu[ob * A + d];
        lpd += invD * (df * iv * a.mud[ob * A + d] + (df * df * iv - 1.f) * rd);
      }
      const float cd = dice ? (valid ? a.scratch[ob] : 0.f) : c2 * lpd;   // R{f'} (DiCE: through the episode recurrences, above)
      for (int d = 0; d < A; ++d) {
        const float rp = a.rho[(size_t)t * a.rstride + d];
        const bool live = rp > LOG_EPS;
        const float r = fmaxf(rp, LOG_EPS), sg = expf(r), iv = 1.f / (sg * sg);
        const float rd = live ? a.rhod[(size_t)t * a.vstride + d] : 0.f;
        const float df = a.act[ob * A + d] - a.mu[ob * A + d];
        const float md = a.mud[ob * A + d];
        a.dmu[ob * A + d] = cd * invD * df * iv + c * invD * (-md * iv - 2.f * df * rd * iv);
        if (live) acc[1 + d] += cd * invD * (df * df * iv - 1.f) + c * invD * (-2.f * df * md * iv - 2.f * df * df * rd * iv);
      }
    }
  }
  if (a.mode == P_LOGP) return;
  for (int k = 0; k < 1 + A; ++k) {
    red[tid] = acc[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) red[tid] += red[tid + s];
      __syncthreads();
    }
    if (tid == 0) {
      if (k == 0) { if (a.loss && a.mode == P_PRIMAL) a.loss[t] = red[0]; }
      else if (a.drho) a.drho[(size_t)t * a.gstride + (k - 1)] = red[0];
    }
    __syncthreads();
  }
}

// keep only sigma and the last Linear of a [T][P] vector (ANIL: body under no_grad)
__global__ void head_mask_kernel(float* __restrict__ v, int p, int body_lo, int body_hi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= body_lo && i < body_hi) v[(size_t)blockIdx.y * p + i] = 0.f;
}

struct StepSet { Acts a; float *dmu, *d2, *d1, *pre2, *pre1, *coef, *coef2; };
struct MetaPlan {
  std::vector<StepSet> st;          // per inner update (second order) or one shared set
  StepSet q;                        // query pass
  Acts ta;                          // tangent activations (scratch)
  float *rdmu, *r2, *r1;            // tangent cotangent scratch
  float *theta, *g, *lam, *hv, *vmask;      // [K+1][T][P], [T][P] ...
  float* oldlp;                     // [n_batches][T][B]
  size_t bytes;
};
static void meta_plan(const mi_policy* p, void* ws, int T, int B, int K, int nb, bool keep_all, MetaPlan& pl) {
  PBump b{reinterpret_cast<char*>(ws), 0};
  const size_t TB = (size_t)T * B, TP = (size_t)T * p->P;
  auto set = [&](StepSet& s) {
    s.a.h1 = b.f(TB * p->H1); s.a.h2 = b.f(TB * p->H2); s.a.mu = b.f(TB * p->A);
    s.dmu = b.f(TB * p->A); s.d2 = b.f(TB * p->H2); s.d1 = b.f(TB * p->H1); s.pre2 = b.f(TB * p->H2); s.pre1 = b.f(TB * p->H1);
    s.coef = b.f(TB); s.coef2 = b.f(TB);
  };
  pl.st.resize(keep_all ? (K > 0 ? K : 1) : 1);
  for (auto& s : pl.st) set(s);
  set(pl.q);
  pl.ta.h1 = b.f(TB * p->H1); pl.ta.h2 = b.f(TB * p->H2); pl.ta.mu = b.f(TB * p->A);
  pl.rdmu = b.f(TB * p->A); pl.r2 = b.f(TB * p->H2); pl.r1 = b.f(TB * p->H1);
  pl.theta = b.f(TP * (K + 1)); pl.g = b.f(TP); pl.lam = b.f(TP); pl.hv = b.f(TP); pl.vmask = b.f(TP);
  pl.oldlp = b.f(TB * (nb > 0 ? nb : 1));
  pl.bytes = align_up(b.off, 256);
}

extern "C" int mi_policy_meta_workspace_bytes(const mi_policy* p, int tasks, int batch, int steps, int n_batches, int second_order,
                                              size_t* bytes) {
  if (!p || !bytes || tasks < 1 || batch < 1 || steps < 0 || n_batches < 0) return MI_ERR_ARG;
  MetaPlan pl;
  meta_plan(p, nullptr, tasks, batch, steps, n_batches, second_order != 0, pl);
  *bytes = pl.bytes;
  return MI_OK;
}

static int policy_meta_batch_impl(mi_policy* p, void* stream, const float* theta, int steps, const int32_t* step_batch,
                                  const int32_t* step_new_old, int n_batches, const float* s_states, const float* s_actions,
                                  const float* s_adv, const int32_t* s_count, const float* s_done, const float* q_states,
                                  const float* q_actions, const float* q_adv, const int32_t* q_count, const float* q_done, int tasks,
                                  int batch, int loss_kind, float clip, float inner_lr, int head_only, int second_order,
                                  int with_grad, float* loss_out, float* theta_out, float* grad_out, void* workspace,
                                  size_t workspace_bytes) {
  if (!p || !theta || !q_states || !q_actions || !q_adv || !loss_out || !workspace) return pfail(p, MI_ERR_ARG, "null argument");
  if (steps > 0 && (!step_batch || !s_states || !s_actions || !s_adv || n_batches < 1)) return pfail(p, MI_ERR_ARG, "support batches missing");
  if (loss_kind != MI_PLOSS_A2C && loss_kind != MI_PLOSS_PPO && loss_kind != MI_PLOSS_DICE)
    return pfail(p, MI_ERR_ARG, "loss_kind must be MI_PLOSS_A2C, MI_PLOSS_PPO or MI_PLOSS_DICE");
  if (loss_kind == MI_PLOSS_DICE && (!q_done || (steps > 0 && !s_done)))
    return pfail(p, MI_ERR_ARG, "the DiCE objective needs the episode-end flags of every replay (s_done / q_done)");
  if (loss_kind == MI_PLOSS_PPO && steps > 0 && !step_new_old) return pfail(p, MI_ERR_ARG, "PPO needs step_new_old");
  if (with_grad && !grad_out) return pfail(p, MI_ERR_ARG, "grad_out is NULL but with_grad != 0");
  for (int k = 0; k < steps; ++k)
    if (step_batch[k] < 0 || step_batch[k] >= n_batches) return pfail(p, MI_ERR_ARG, "step_batch entry out of range");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int T = tasks, B = batch, K = steps;
  const size_t P = p->P, TB = (size_t)T * B, TP = (size_t)T * P;
  const bool so = second_order && with_grad;
  MetaPlan pl;
  meta_plan(p, workspace, T, B, K, n_batches, so, pl);
  if (pl.bytes > workspace_bytes) return pfail(p, MI_ERR_WORKSPACE, "workspace too small: need " + std::to_string(pl.bytes));
  const int body_lo = (int)p->o_w1, body_hi = (int)p->o_w3;          // W1, b1, W2, b2
  auto mask = [&](float* v) {
    if (head_only) hipLaunchKernelGGL(head_mask_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, v, (int)P, body_lo, body_hi);
  };
  hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, theta, (size_t)0, pl.g, 0.f, (int)P, pl.theta);
  PCHK(p, hipGetLastError());
  auto primal = [&](StepSet& s, const float* th, const float* states, const float* actions, const float* adv, const int32_t* count,
                    int kind, const float* oldlp, int value_ratio_one, float* g, float* loss, const float* done) -> int {
    int rc = mlp_forward(p, st, T, B, states, th, P, s.a);
    if (rc) return rc;
    PCHK(p, hipMemsetAsync(g, 0, TP * sizeof(float), st));
    Gauss2Args ga{};
    ga.mu = s.a.mu; ga.rho = th + p->o_sigma; ga.rstride = P; ga.act = actions; ga.adv = adv; ga.count = count; ga.oldlp = oldlp;
    ga.coef = s.coef; ga.coef2 = s.coef2; ga.dmu = s.dmu; ga.drho = g + p->o_sigma; ga.gstride = P; ga.loss = loss; ga.clip = clip;
    ga.B = B; ga.A = p->A; ga.kind = kind; ga.mode = P_PRIMAL; ga.value_ratio_one = value_ratio_one; ga.done = done;
    hipLaunchKernelGGL(gauss2_kernel, dim3(T), dim3(256), 0, st, ga);
    PCHK(p, hipGetLastError());
    return mlp_backward(p, st, T, B, states, th, P, s.a, s.dmu, s.d2, s.d1, g, s.pre2, s.pre1, false);
  };
  // ---- inner updates
  for (int k = 0; k < K; ++k) {
    StepSet& s = so ? pl.st[k] : pl.st[0];
    float* th = pl.theta + (size_t)k * TP;
    const int bi = step_batch[k];
    const float* xs = s_states + (size_t)bi * TB * p->S;
    const float* as = s_actions + (size_t)bi * TB * p->A;
    const float* ad = s_adv + (size_t)bi * TB;
    const int32_t* cn = s_count ? s_count + (size_t)bi * T : nullptr;
    float* olp = pl.oldlp + (size_t)bi * TB;
    if (loss_kind == MI_PLOSS_PPO && step_new_old[k]) {       // old_log_probs = learner.log_prob(...) under no_grad (rl.py:282-283)
      int rc = mlp_forward(p, st, T, B, xs, th, P, s.a);
      if (rc) return rc;
      Gauss2Args gl{};
      gl.mu = s.a.mu; gl.rho = th + p->o_sigma; gl.rstride = P; gl.act = as; gl.count = cn; gl.lp_out = olp; gl.B = B; gl.A = p->A; gl.mode = P_LOGP;
      hipLaunchKernelGGL(gauss2_kernel, dim3(T), dim3(256), 0, st, gl);
      PCHK(p, hipGetLastError());
    }
    int rc = primal(s, th, xs, as, ad, cn, loss_kind, olp, 0, pl.g, pl.hv /* scratch for the step loss */,
                    s_done ? s_done + (size_t)bi * TB : nullptr);
    if (rc) return rc;
    mask(pl.g);
    hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, th, P, pl.g, inner_lr, (int)P, th + TP);
    PCHK(p, hipGetLastError());
  }
  float* thK = pl.theta + (size_t)K * TP;
  if (theta_out) PCHK(p, hipMemcpyAsync(theta_out, thK, TP * sizeof(float), hipMemcpyDeviceToDevice, st));
  // ---- query loss: VPG = a2c loss; PPO = ppo loss against the adapted policy itself (ratio == 1: value -mean(A), gradient of A2C form)
  int rc = primal(pl.q, thK, q_states, q_actions, q_adv, q_count, loss_kind == MI_PLOSS_DICE ? MI_PLOSS_DICE : MI_PLOSS_A2C, nullptr,
                  loss_kind == MI_PLOSS_PPO ? 1 : 0, pl.lam, loss_out, q_done);
  if (rc) return rc;
  if (!with_grad) return MI_OK;
  // ---- adjoint recursion through the updates
  if (so) {
    for (int k = K - 1; k >= 0; --k) {
      StepSet& s = pl.st[k];
      const float* th = pl.theta + (size_t)k * TP;
      const int bi = step_batch[k];
      const float* xs = s_states + (size_t)bi * TB * p->S;
      const float* as = s_actions + (size_t)bi * TB * p->A;
      const int32_t* cn = s_count ? s_count + (size_t)bi * T : nullptr;
      // v = [mask] lam ; hv = H_k v
      PCHK(p, hipMemcpyAsync(pl.vmask, pl.lam, TP * sizeof(float), hipMemcpyDeviceToDevice, st));
      mask(pl.vmask);
      PCHK(p, hipMemsetAsync(pl.hv, 0, TP * sizeof(float), st));
      rc = mlp_tangent_forward(p, st, T, B, xs, th, P, s.a, pl.vmask, pl.ta);
      if (rc) return rc;
      Gauss2Args gt{};
      gt.mu = s.a.mu; gt.mud = pl.ta.mu; gt.rho = th + p->o_sigma; gt.rstride = P; gt.rhod = pl.vmask + p->o_sigma; gt.vstride = P;
      gt.act = as; gt.count = cn; gt.coef = s.coef; gt.coef2 = s.coef2; gt.dmu = pl.rdmu; gt.drho = pl.hv + p->o_sigma; gt.gstride = P;
      gt.B = B; gt.A = p->A; gt.mode = P_TANGENT;
      if (loss_kind == MI_PLOSS_DICE) {                      // the episode recurrences couple the samples of a replay
        gt.kind = MI_PLOSS_DICE; gt.done = s_done + (size_t)bi * TB; gt.adv = s_adv + (size_t)bi * TB;
        gt.scratch = pl.oldlp + (size_t)bi * TB;             // free: old log-probs exist only for PPO
      }
      hipLaunchKernelGGL(gauss2_kernel, dim3(T), dim3(256), 0, st, gt);
      PCHK(p, hipGetLastError());
      rc = mlp_tangent_backward(p, st, T, B, xs, th, P, s.a, pl.ta, pl.vmask, s.dmu, s.d2, s.d1, s.pre2, s.pre1, pl.rdmu, pl.r2, pl.r1, pl.hv);
      if (rc) return rc;
      mask(pl.hv);      // the body sat under no_grad during the update: no path from theta_{k+1} back into it
      hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, pl.lam, P, pl.hv, inner_lr, (int)P, pl.lam);
      PCHK(p, hipGetLastError());
    }
  }
  hipLaunchKernelGGL(mean_tasks_kernel, dim3(ceil_div((int)P, 256)), dim3(256), 0, st, pl.lam, T, (int)P, 1.f, (const float*)nullptr, 0.f, grad_out);
  PCHK(p, hipGetLastError());
  return MI_OK;
}

extern "C" int mi_policy_meta_batch(mi_policy* p, void* stream, const float* theta, int steps, const int32_t* step_batch,
                                    const int32_t* step_new_old, int n_batches, const float* s_states, const float* s_actions,
                                    const float* s_adv, const int32_t* s_count, const float* q_states, const float* q_actions,
                                    const float* q_adv, const int32_t* q_count, int tasks, int batch, int loss_kind, float clip,
                                    float inner_lr, int head_only, int second_order, int with_grad, float* loss_out,
                                    float* theta_out, float* grad_out, void* workspace, size_t workspace_bytes) {
  if (loss_kind == MI_PLOSS_DICE) return pfail(p, MI_ERR_ARG, "MI_PLOSS_DICE needs the episode-end flags: call mi_policy_meta_batch_dones");
  return policy_meta_batch_impl(p, stream, theta, steps, step_batch, step_new_old, n_batches, s_states, s_actions, s_adv, s_count, nullptr,
                                q_states, q_actions, q_adv, q_count, nullptr, tasks, batch, loss_kind, clip, inner_lr, head_only,
                                second_order, with_grad, loss_out, theta_out, grad_out, workspace, workspace_bytes);
}
// The same with the replays' episode-end flags (cherry `dones`: s_done [n_batches, tasks, batch], q_done [tasks, batch]; 1 at
// the last step of every episode): required by MI_PLOSS_DICE, ignored by the other losses.
extern "C" int mi_policy_meta_batch_dones(mi_policy* p, void* stream, const float* theta, int steps, const int32_t* step_batch,
                                          const int32_t* step_new_old, int n_batches, const float* s_states, const float* s_actions,
                                          const float* s_adv, const int32_t* s_count, const float* s_done, const float* q_states,
                                          const float* q_actions, const float* q_adv, const int32_t* q_count, const float* q_done,
                                          int tasks, int batch, int loss_kind, float clip, float inner_lr, int head_only,
                                          int second_order, int with_grad, float* loss_out, float* theta_out, float* grad_out,
                                          void* workspace, size_t workspace_bytes) {
  return policy_meta_batch_impl(p, stream, theta, steps, step_batch, step_new_old, n_batches, s_states, s_actions, s_adv, s_count, s_done,
                                q_states, q_actions, q_adv, q_count, q_done, tasks, batch, loss_kind, clip, inner_lr, head_only,
                                second_order, with_grad, loss_out, theta_out, grad_out, workspace, workspace_bytes);
}

// =====================================================================================================================
// MAML-TRPO with any number of inner updates (params['adapt_steps'] > 1): meta_surrogate_loss replays `steps` second-order
// trpo_update calls, one per support replay (rl.py:447-453), before the surrogate / KL on the query replay.  With
// J = d theta_K / d theta = prod_k (I - lr H_k(theta_k)):
//   grad mean_t S_t = mean_t J_t^T grad S_t(theta_K)                 (adjoint recursion, as in mi_policy_meta_batch)
//   Fvp(v)          = mean_t J_t^T F_t J_t v + damping v              (tangent recursion forward, Fisher at the query, adjoint back;
//                                                                     exact where the adapted policy equals the stored old one)
static int hvp_step(mi_policy* p, hipStream_t st, MetaPlan& pl, StepSet& s, int T, int B, const float* th, const float* xs,
                    const float* as, const int32_t* cn, const float* v, float* hv) {
  const size_t P = p->P;
  PCHK(p, hipMemsetAsync(hv, 0, (size_t)T * P * sizeof(float), st));
  int rc = mlp_tangent_forward(p, st, T, B, xs, th, P, s.a, v, pl.ta);
  if (rc) return rc;
  Gauss2Args gt{};
  gt.mu = s.a.mu; gt.mud = pl.ta.mu; gt.rho = th + p->o_sigma; gt.rstride = P; gt.rhod = v + p->o_sigma; gt.vstride = P;
  gt.act = as; gt.count = cn; gt.coef = s.coef; gt.coef2 = s.coef2; gt.dmu = pl.rdmu; gt.drho = hv + p->o_sigma; gt.gstride = P;
  gt.B = B; gt.A = p->A; gt.mode = P_TANGENT;
  hipLaunchKernelGGL(gauss2_kernel, dim3(T), dim3(256), 0, st, gt);
  PCHK(p, hipGetLastError());
  return mlp_tangent_backward(p, st, T, B, xs, th, P, s.a, pl.ta, v, s.dmu, s.d2, s.d1, s.pre2, s.pre1, pl.rdmu, pl.r2, pl.r1, hv);
}

extern "C" int mi_trpo_steps_workspace_bytes(const mi_policy* p, int tasks, int batch, int steps, size_t* bytes) {
  if (!p || !bytes || tasks < 1 || batch < 1 || steps < 1) return MI_ERR_ARG;
  MetaPlan pl;
  meta_plan(p, nullptr, tasks, batch, steps, 1, true, pl);
  *bytes = pl.bytes + align_up((size_t)2 * tasks * sizeof(float), 256);
  return MI_OK;
}

extern "C" int mi_trpo_surrogate_steps(mi_policy* p, void* stream, const float* theta, int steps, const float* s_states,
                                       const float* s_actions, const float* s_adv, const int32_t* s_count, const float* q_states,
                                       const float* q_actions, const float* q_adv, const int32_t* q_count, const float* old_loc,
                                       const float* old_scale, int tasks, int batch, float inner_lr, float* loss_out,
                                       float* kl_out, float* grad_out, void* workspace, size_t workspace_bytes) {
  if (!p || !theta || !s_states || !s_actions || !s_adv || !q_states || !q_actions || !q_adv || !old_loc || !old_scale || !loss_out ||
      !kl_out || !workspace || steps < 1)
    return pfail(p, MI_ERR_ARG, "null argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int T = tasks, B = batch, K = steps;
  const size_t P = p->P, TB = (size_t)T * B, TP = (size_t)T * P;
  MetaPlan pl;
  meta_plan(p, workspace, T, B, K, 1, true, pl);
  size_t need = 0;
  mi_trpo_steps_workspace_bytes(p, T, B, K, &need);
  if (need > workspace_bytes) return pfail(p, MI_ERR_WORKSPACE, "workspace too small: need " + std::to_string(need));
  float* loss_t = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + pl.bytes);
  float* kl_t = loss_t + T;
  hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, theta, (size_t)0, pl.g, 0.f, (int)P, pl.theta);
  PCHK(p, hipGetLastError());
  for (int k = 0; k < K; ++k) {                                   // trpo_update on support replay k (a2c loss, normalised advantages)
    StepSet& s = pl.st[k];
    float* th = pl.theta + (size_t)k * TP;
    const float* xs = s_states + (size_t)k * TB * p->S;
    const float* as = s_actions + (size_t)k * TB * p->A;
    int rc = mlp_forward(p, st, T, B, xs, th, P, s.a);
    if (rc) return rc;
    PCHK(p, hipMemsetAsync(pl.g, 0, TP * sizeof(float), st));
    Gauss2Args ga{};
    ga.mu = s.a.mu; ga.rho = th + p->o_sigma; ga.rstride = P; ga.act = as; ga.adv = s_adv + (size_t)k * TB;
    ga.count = s_count ? s_count + (size_t)k * T : nullptr; ga.coef = s.coef; ga.coef2 = s.coef2; ga.dmu = s.dmu;
    ga.drho = pl.g + p->o_sigma; ga.gstride = P; ga.loss = pl.hv; ga.B = B; ga.A = p->A; ga.kind = MI_PLOSS_A2C; ga.mode = P_PRIMAL;
    hipLaunchKernelGGL(gauss2_kernel, dim3(T), dim3(256), 0, st, ga);
    PCHK(p, hipGetLastError());
    rc = mlp_backward(p, st, T, B, xs, th, P, s.a, s.dmu, s.d2, s.d1, pl.g, s.pre2, s.pre1, false);
    if (rc) return rc;
    hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, th, P, pl.g, inner_lr, (int)P, th + TP);
    PCHK(p, hipGetLastError());
  }
  float* thK = pl.theta + (size_t)K * TP;
  int rc = mlp_forward(p, st, T, B, q_states, thK, P, pl.q.a);
  if (rc) return rc;
  PCHK(p, hipMemsetAsync(pl.lam, 0, TP * sizeof(float), st));
  GaussArgs gq{};
  gq.mu = pl.q.a.mu; gq.rho = thK + p->o_sigma; gq.rstride = P; gq.act = q_actions; gq.adv = q_adv; gq.count = q_count;
  gq.old_loc = old_loc; gq.old_scale = old_scale; gq.coef = pl.q.coef; gq.dmu = pl.q.dmu; gq.drho = pl.lam + p->o_sigma;
  gq.gstride = P; gq.loss = loss_t; gq.kl = kl_t; gq.B = B; gq.A = p->A; gq.mode = G_SURROGATE;
  PCHK(p, gauss(st, T, gq));
  hipLaunchKernelGGL(mean_tasks_kernel, dim3(1), dim3(64), 0, st, loss_t, T, 1, 1.f / (float)T, (const float*)nullptr, 0.f, loss_out);
  hipLaunchKernelGGL(mean_tasks_kernel, dim3(1), dim3(64), 0, st, kl_t, T, 1, 1.f / (float)T, (const float*)nullptr, 0.f, kl_out);
  PCHK(p, hipGetLastError());
  if (!grad_out) return MI_OK;
  rc = mlp_backward(p, st, T, B, q_states, thK, P, pl.q.a, pl.q.dmu, pl.q.d2, pl.q.d1, pl.lam);
  if (rc) return rc;
  for (int k = K - 1; k >= 0; --k) {
    const float* th = pl.theta + (size_t)k * TP;
    rc = hvp_step(p, st, pl, pl.st[k], T, B, th, s_states + (size_t)k * TB * p->S, s_actions + (size_t)k * TB * p->A,
                  s_count ? s_count + (size_t)k * T : nullptr, pl.lam, pl.hv);
    if (rc) return rc;
    hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, pl.lam, P, pl.hv, inner_lr, (int)P, pl.lam);
    PCHK(p, hipGetLastError());
  }
  hipLaunchKernelGGL(mean_tasks_kernel, dim3(ceil_div((int)P, 256)), dim3(256), 0, st, pl.lam, T, (int)P, 1.f / (float)T,
                     (const float*)nullptr, 0.f, grad_out);
  PCHK(p, hipGetLastError());
  return MI_OK;
}

extern "C" int mi_trpo_fvp_steps(mi_policy* p, void* stream, int steps, const float* s_states, const float* s_actions,
                                 const int32_t* s_count, const float* q_states, const int32_t* q_count, int tasks, int batch,
                                 float inner_lr, float damping, const float* v, float* out, void* workspace, size_t workspace_bytes) {
  if (!p || !s_states || !s_actions || !q_states || !v || !out || !workspace || steps < 1) return pfail(p, MI_ERR_ARG, "null argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int T = tasks, B = batch, K = steps;
  const size_t P = p->P, TB = (size_t)T * B, TP = (size_t)T * P;
  MetaPlan pl;
  meta_plan(p, workspace, T, B, K, 1, true, pl);
  if (pl.bytes > workspace_bytes) return pfail(p, MI_ERR_WORKSPACE, "workspace too small");
  auto sup = [&](int k, const float*& xs, const float*& as, const int32_t*& cn) {
    xs = s_states + (size_t)k * TB * p->S; as = s_actions + (size_t)k * TB * p->A; cn = s_count ? s_count + (size_t)k * T : nullptr;
  };
  // u = J v : u <- u - lr H_k u, k = 0 .. K-1   (pl.vmask holds u)
  hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, v, (size_t)0, pl.g, 0.f, (int)P, pl.vmask);
  PCHK(p, hipGetLastError());
  for (int k = 0; k < K; ++k) {
    const float *xs, *as; const int32_t* cn;
    sup(k, xs, as, cn);
    int rc = hvp_step(p, st, pl, pl.st[k], T, B, pl.theta + (size_t)k * TP, xs, as, cn, pl.vmask, pl.hv);
    if (rc) return rc;
    hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, pl.vmask, P, pl.hv, inner_lr, (int)P, pl.vmask);
    PCHK(p, hipGetLastError());
  }
  // w = F u at the query (Gaussian Fisher at new == old)
  float* thK = pl.theta + (size_t)K * TP;
  int rc = mlp_tangent_forward(p, st, T, B, q_states, thK, P, pl.q.a, pl.vmask, pl.ta);
  if (rc) return rc;
  PCHK(p, hipMemsetAsync(pl.lam, 0, TP * sizeof(float), st));
  GaussArgs gf{};
  gf.mu = pl.q.a.mu; gf.mud = pl.ta.mu; gf.rho = thK + p->o_sigma; gf.rstride = P; gf.rhod = pl.vmask + p->o_sigma; gf.vstride = P;
  gf.count = q_count; gf.dmu = pl.rdmu; gf.drho = pl.lam + p->o_sigma; gf.gstride = P; gf.B = B; gf.A = p->A; gf.mode = G_FISHER;
  PCHK(p, gauss(st, T, gf));
  rc = mlp_backward(p, st, T, B, q_states, thK, P, pl.q.a, pl.rdmu, pl.r2, pl.r1, pl.lam);
  if (rc) return rc;
  // out = mean_t J^T w + damping v
  for (int k = K - 1; k >= 0; --k) {
    const float *xs, *as; const int32_t* cn;
    sup(k, xs, as, cn);
    rc = hvp_step(p, st, pl, pl.st[k], T, B, pl.theta + (size_t)k * TP, xs, as, cn, pl.lam, pl.hv);
    if (rc) return rc;
    hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, pl.lam, P, pl.hv, inner_lr, (int)P, pl.lam);
    PCHK(p, hipGetLastError());
  }
  hipLaunchKernelGGL(mean_tasks_kernel, dim3(ceil_div((int)P, 256)), dim3(256), 0, st, pl.lam, T, (int)P, 1.f / (float)T, v, damping, out);
  PCHK(p, hipGetLastError());
  return MI_OK;
}

// =====================================================================================================================
// Exact Hessian-vector product of the mean KL where the re-adapted policy differs from the stored old one: ANIL-TRPO.
// The reference's anil_trpo.py adapts the stored old policies with the body under no_grad (core_functions/rl.py:381-382), but
// meta_surrogate_loss replays the inner step on clone_module(policy) with every parameter (rl.py:447-453): at the current
// parameters new != old, the KL gradient c_t = grad KL_t(theta'_t) is not zero, and
//   Hess_theta KL_t(theta'_t(theta)) v = J_t^T [Hess KL_t(theta'_t)] J_t v  -  lr * T_t[v, c_t],      J_t = I - lr H_t(theta),
// T_t[v, c] = d/d eps ( H_t(theta + eps v) c ): the third derivative of the inner loss contracted with v and c.  Hess KL u is a
// forward-over-reverse sweep over the query pass with the EXACT output Hessian of KL(new || old) (diag: 1/sigma_old^2 on the
// mean, 2 (sigma/sigma_old)^2 on log sigma) and a non-zero primal cotangent; T_t[v, c] = R_v{R_c{grad L_t}} is a SECOND-order
// tangent sweep over the support pass: every quantity carries a c-tangent, a v-tangent and a mixed (cv) tangent,
//   z_cv = W_c h_v + W_v h_c + W h_cv,     h_cv = phi' z_cv + phi'' z_c z_v,
//   dz_cv = phi' dh_cv + phi'' (z_v dh_c + z_c dh_v) + (phi''' z_c z_v + phi'' z_cv) dh,
//   dW_cv = dz_cv^T h + dz_c^T h_v + dz_v^T h_c + dz^T h_cv,     dh_in_cv = dz_cv W + dz_c W_v + dz_v W_c
// (tanh: phi' = 1 - h^2 =: p, phi'' = -2 h p, phi''' = p (6 h^2 - 2); ReLU: phi'' = phi''' = 0).  The c-tangent sweep depends on
// the context only and is computed once (mi_trpo_kl_prepare); v-tangent and mixed sweeps run per product.  All dense products
// go through the same fp32 matrix-pipe kernels as above with up to four summed terms.
struct EwArgs {
  const float* h;                         // stored activation of the layer
  const float *a0, *a1, *a2, *a3, *a4, *a5, *a6;
  float* out;
  size_t n;
  int act, mode;
};
enum { EW_ACT_T = 0, EW_ACT_CV = 1, EW_GATE_T = 2, EW_GATE_CV = 3 };
//  EW_ACT_T   out = p a0                                                      h_t from z_t = a0
//  EW_ACT_CV  out = p a0 + phi'' a1 a2                                        h_cv from z_cv = a0 (null: 0), z_c = a1, z_v = a2
//  EW_GATE_T  out = p a0 + phi'' a1 a2                                        dz_t from dh_t = a0, z_t = a1, dh = a2
//  EW_GATE_CV out = p a0 + phi'' (a1 a2 + a3 a4) + (phi''' a3 a1 + phi'' a5) a6
//             dz_cv from dh_cv = a0, z_v = a1, dh_c = a2, z_c = a3, dh_v = a4, z_cv = a5 (null: 0), dh = a6
__global__ __launch_bounds__(256) void ew_kernel(EwArgs a) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const float h = a.h[i];
  float p, p2, p3;
  if (a.act == ACT_TANH) { p = 1.f - h * h; p2 = -2.f * h * p; p3 = p * (6.f * h * h - 2.f); }
  else { p = h > 0.f ? 1.f : 0.f; p2 = 0.f; p3 = 0.f; }
  float o;
  if (a.mode == EW_ACT_T) o = p * a.a0[i];
  else if (a.mode == EW_ACT_CV || a.mode == EW_GATE_T) o = (a.a0 ? p * a.a0[i] : 0.f) + p2 * a.a1[i] * a.a2[i];
  else {
    const float zv = a.a1[i], zc = a.a3[i];
    o = p * a.a0[i] + p2 * (zv * a.a2[i] + zc * a.a4[i]) + (p3 * zc * zv + (a.a5 ? p2 * a.a5[i] : 0.f)) * a.a6[i];
  }
  a.out[i] = o;
}
static hipError_t ew(hipStream_t st, int act, int mode, size_t n, const float* h, float* out, const float* a0, const float* a1 = nullptr,
                     const float* a2 = nullptr, const float* a3 = nullptr, const float* a4 = nullptr, const float* a5 = nullptr,
                     const float* a6 = nullptr) {
  EwArgs a{h, a0, a1, a2, a3, a4, a5, a6, out, n, act, mode};
  hipLaunchKernelGGL(ew_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
  return hipGetLastError();
}

struct FTerm { const float* x; const float* w; size_t ws; };
// y = sum_k x_k w_k^T (+ bias): raw pre-activation, no gate
static hipError_t dense_fwd_n(hipStream_t st, int T, int B, int I, int O, const FTerm* tm, int n, const float* bias, size_t bs, float* y) {
  DenseArgs a{};
  for (int k = 0; k < n; ++k) { a.x[k] = tm[k].x; a.w[k] = tm[k].w; a.wstride[k] = tm[k].ws; }
  a.bias = bias; a.bstride = bs; a.y = y; a.B = B; a.I = I; a.O = O; a.nterms = n; a.act = ACT_NONE;
  launch_dense<true>(st, T, a);
  return hipGetLastError();
}
// dx = sum_k dy_k w_k: raw cotangent w.r.t. the layer input (before any phi' factor)
static hipError_t dense_bwd_x_n(hipStream_t st, int T, int B, int I, int O, const FTerm* tm, int n, float* dx) {
  DenseArgs a{};
  for (int k = 0; k < n; ++k) { a.x[k] = tm[k].x; a.w[k] = tm[k].w; a.wstride[k] = tm[k].ws; }
  a.y = dx; a.B = B; a.I = I; a.O = O; a.nterms = n; a.act = ACT_NONE;
  launch_dense<false>(st, T, a);
  return hipGetLastError();
}
struct WTerm { const float* dy; const float* x; };
static hipError_t dense_bwd_w_n(hipStream_t st, int T, int B, int I, int O, const WTerm* tm, int n, float* dw, float* db, size_t gs) {
  DenseWArgs a{};
  for (int k = 0; k < n; ++k) { a.dy[k] = tm[k].dy; a.x[k] = tm[k].x; }
  a.dw = dw; a.db = db; a.gstride = gs; a.B = B; a.I = I; a.O = O; a.nterms = n;
  hipLaunchKernelGGL(dense_wgrad_mfma_kernel, dim3(ceil_div(O, 32) * ceil_div(I + 1, 32), T), dim3(512), 0, st, a);
  return hipGetLastError();
}

struct GaussKlArgs {
  const float* mu;        // [T][B][A] policy mean on this pass
  const float* rho; size_t rstride;        // log-sigma parameter
  const float* old_loc; const float* old_scale;      // KL modes
  const float* mud;       // KL_HESS: tangent of mu;  TAN2: mixed tangent mu_cv
  const float* rhod; size_t vstride;                 // KL_HESS: tangent of rho
  const float *muc, *muv;                            // TAN2: first tangents of mu
  const float* rhoc; size_t cstride;                 // TAN2: c-direction of rho
  const float* rhov; size_t vvstride;                // TAN2: v-direction of rho
  const float* act; const float* coef;               // TAN2: actions, dL/dlogp per sample
  const int32_t* count;
  float* dmu; float* drho; size_t gstride;
  int B, A, mode;
};
enum { KL_GRAD = 0, KL_HESS = 1, G_TAN2 = 2 };
// One workgroup per task.
//  KL_GRAD  cotangent of mean KL(new || old):       dmu = (mu - mu_old) / (B D s_old^2),   drho = (vr - 1) / D,  vr = (s / s_old)^2
//  KL_HESS  its exact output Hessian times (mud, rhod):  dmu = mud / (B D s_old^2),        drho = 2 vr rhod / D
//  G_TAN2   second tangent of the A2C log-prob cotangents (kappa = coef / D, d = a - mu, iv = exp(-2 rho)):
//           R_cv{g_mu}  = kappa [ -mu_cv iv + 2 iv (mu_c rho_v + mu_v rho_c) + 4 d iv rho_c rho_v ]
//           R_cv{g_rho} = kappa [ 2 mu_c mu_v iv - 2 d mu_cv iv + 4 d iv (mu_c rho_v + mu_v rho_c) + 4 d^2 iv rho_c rho_v ]
__global__ __launch_bounds__(256) void gauss_kl_kernel(GaussKlArgs a) {
  __shared__ float red[256];
  const int t = blockIdx.x, tid = threadIdx.x;
  const int B = a.B, A = a.A, cnt = a.count ? a.count[t] : B;
  const float invB = 1.f / (float)cnt, invD = 1.f / (float)A;
  float acc[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) acc[k] = 0.f;
  for (int b = tid; b < B; b += 256) {
    const bool valid = b < cnt;
    const size_t ob = (size_t)t * B + b;
    for (int d = 0; d < A; ++d) {
      const float rp = a.rho[(size_t)t * a.rstride + d];
      const bool live = rp > LOG_EPS;
      const float r = fmaxf(rp, LOG_EPS);
      if (a.mode == G_TAN2) {
        const float iv = expf(-2.f * r);
        const float kap = valid ? a.coef[ob] * invD : 0.f;
        const float df = a.act[ob * A + d] - a.mu[ob * A + d];
        const float mc = a.muc[ob * A + d], mv = a.muv[ob * A + d], mcv = a.mud[ob * A + d];
        const float rc = live ? a.rhoc[(size_t)t * a.cstride + d] : 0.f, rv = live ? a.rhov[(size_t)t * a.vvstride + d] : 0.f;
        a.dmu[ob * A + d] = kap * (-mcv * iv + 2.f * iv * (mc * rv + mv * rc) + 4.f * df * iv * rc * rv);
        if (live) acc[d] += kap * (2.f * mc * mv * iv - 2.f * df * mcv * iv + 4.f * df * iv * (mc * rv + mv * rc) + 4.f * df * df * iv * rc * rv);
      } else {
        const float so = a.old_scale[(size_t)t * A + d];
        const float w = valid ? invB * invD / (so * so) : 0.f;
        a.dmu[ob * A + d] = a.mode == KL_GRAD ? w * (a.mu[ob * A + d] - a.old_loc[ob * A + d]) : w * a.mud[ob * A + d];
      }
    }
  }
  for (int k = 0; k < A; ++k) {
    float v;
    if (a.mode == G_TAN2) {
      red[tid] = acc[k];
      __syncthreads();
      for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
      }
      v = red[0];
      __syncthreads();
    } else {
      const float rp = a.rho[(size_t)t * a.rstride + k];
      const float sg = expf(fmaxf(rp, LOG_EPS)), so = a.old_scale[(size_t)t * A + k], vr = (sg / so) * (sg / so);
      v = rp > LOG_EPS ? (a.mode == KL_GRAD ? (vr - 1.f) * invD : 2.f * vr * a.rhod[(size_t)t * a.vstride + k] * invD) : 0.f;
    }
    if (tid == 0 && a.drho) a.drho[(size_t)t * a.gstride + k] = v;
  }
}

struct TanSweep { float *z1, *h1, *z2, *h2, *mu, *rdmu, *dh2, *dz2, *dh1, *dz1; };
struct GenPlan {
  float *k_dmu, *k_d2, *k_d1, *k_pre2, *k_pre1;      // KL primal cotangents on the query pass at theta'
  float* c;                                           // [T][P] grad KL_t(theta'_t)
  TanSweep sc, sv;                                    // c-tangent (context) and v-tangent (per product) sweeps over the support pass
  float *hcv1, *zcv2, *hcv2, *mucv, *rdmu_cv, *dh2cv, *dz2cv, *dh1cv, *dz1cv;
  float *s3, *scr;                                    // [T][P]: third-order term, scratch
  size_t bytes;
};
static void gen_plan(const mi_policy* p, void* ws, int T, int B, TrpoPlan& pl, GenPlan& gp) {
  trpo_plan(p, ws, T, B, pl);
  PBump b{reinterpret_cast<char*>(ws), pl.bytes};
  const size_t TB = (size_t)T * B, TP = (size_t)T * p->P;
  gp.k_dmu = b.f(TB * p->A); gp.k_d2 = b.f(TB * p->H2); gp.k_d1 = b.f(TB * p->H1); gp.k_pre2 = b.f(TB * p->H2); gp.k_pre1 = b.f(TB * p->H1);
  gp.c = b.f(TP);
  auto sweep = [&](TanSweep& s) {
    s.z1 = b.f(TB * p->H1); s.h1 = b.f(TB * p->H1); s.z2 = b.f(TB * p->H2); s.h2 = b.f(TB * p->H2); s.mu = b.f(TB * p->A);
    s.rdmu = b.f(TB * p->A); s.dh2 = b.f(TB * p->H2); s.dz2 = b.f(TB * p->H2); s.dh1 = b.f(TB * p->H1); s.dz1 = b.f(TB * p->H1);
  };
  sweep(gp.sc); sweep(gp.sv);
  gp.hcv1 = b.f(TB * p->H1); gp.zcv2 = b.f(TB * p->H2); gp.hcv2 = b.f(TB * p->H2); gp.mucv = b.f(TB * p->A); gp.rdmu_cv = b.f(TB * p->A);
  gp.dh2cv = b.f(TB * p->H2); gp.dz2cv = b.f(TB * p->H2); gp.dh1cv = b.f(TB * p->H1); gp.dz1cv = b.f(TB * p->H1);
  gp.s3 = b.f(TP); gp.scr = b.f(TP);
  gp.bytes = align_up(b.off, 256);
}
extern "C" int mi_trpo_general_workspace_bytes(const mi_policy* p, int tasks, int batch, size_t* bytes) {
  if (!p || !bytes || tasks < 1 || batch < 1) return MI_ERR_ARG;
  TrpoPlan pl; GenPlan gp;
  gen_plan(p, nullptr, tasks, batch, pl, gp);
  *bytes = gp.bytes;
  return MI_OK;
}

// First-order tangent sweep over the cached support pass at theta (shared) along direction d (stride ds floats per task, 0 = one
// direction for all tasks), keeping the pre-activation tangents the mixed sweep needs.
static int tan_sweep(mi_policy* p, hipStream_t st, TrpoPlan& pl, int T, int B, const float* theta, const float* xs, const float* as,
                     const int32_t* cn, const float* d, size_t ds, TanSweep& s, float* drho_scratch) {
  const size_t TB = (size_t)T * B;
  { FTerm tm[1] = {{xs, d + p->o_w1, ds}};
    PCHK(p, dense_fwd_n(st, T, B, p->S, p->H1, tm, 1, d + p->o_b1, ds, s.z1)); }
  PCHK(p, ew(st, p->act, EW_ACT_T, TB * p->H1, pl.sa.h1, s.h1, s.z1));
  { FTerm tm[2] = {{pl.sa.h1, d + p->o_w2, ds}, {s.h1, theta + p->o_w2, 0}};
    PCHK(p, dense_fwd_n(st, T, B, p->H1, p->H2, tm, 2, d + p->o_b2, ds, s.z2)); }
  PCHK(p, ew(st, p->act, EW_ACT_T, TB * p->H2, pl.sa.h2, s.h2, s.z2));
  { FTerm tm[2] = {{pl.sa.h2, d + p->o_w3, ds}, {s.h2, theta + p->o_w3, 0}};
    PCHK(p, dense_fwd_n(st, T, B, p->H2, p->A, tm, 2, d + p->o_b3, ds, s.mu)); }
  GaussArgs ga{};
  ga.mu = pl.sa.mu; ga.mud = s.mu; ga.rho = theta + p->o_sigma; ga.rstride = 0; ga.rhod = d + p->o_sigma; ga.vstride = ds;
  ga.act = as; ga.count = cn; ga.coef = pl.s_coef; ga.dmu = s.rdmu; ga.drho = drho_scratch + p->o_sigma; ga.gstride = p->P;
  ga.B = B; ga.A = p->A; ga.mode = G_TANGENT;
  PCHK(p, gauss(st, T, ga));
  { FTerm tm[2] = {{s.rdmu, theta + p->o_w3, 0}, {pl.s_dmu, d + p->o_w3, ds}};
    PCHK(p, dense_bwd_x_n(st, T, B, p->H2, p->A, tm, 2, s.dh2)); }
  PCHK(p, ew(st, p->act, EW_GATE_T, TB * p->H2, pl.sa.h2, s.dz2, s.dh2, s.z2, pl.s_pre2));
  { FTerm tm[2] = {{s.dz2, theta + p->o_w2, 0}, {pl.s_d2, d + p->o_w2, ds}};
    PCHK(p, dense_bwd_x_n(st, T, B, p->H1, p->H2, tm, 2, s.dh1)); }
  PCHK(p, ew(st, p->act, EW_GATE_T, TB * p->H1, pl.sa.h1, s.dz1, s.dh1, s.z1, pl.s_pre1));
  return MI_OK;
}

// After mi_trpo_surrogate(theta, ...) on the same workspace: the KL cotangent pass on the query replay (c_t = grad KL_t at theta'_t)
// and the c-tangent sweep over the support pass.  kl_grad_out (optional, [P]) = d mean_t KL_t / d theta = mean_t (I - lr H_t) c_t.
extern "C" int mi_trpo_kl_prepare(mi_policy* p, void* stream, const float* theta, const float* s_states, const float* s_actions,
                                  const int32_t* s_count, const float* q_states, const int32_t* q_count, const float* old_loc,
                                  const float* old_scale, int tasks, int batch, float inner_lr, float* kl_grad_out, void* workspace,
                                  size_t workspace_bytes) {
  if (!p || !theta || !s_states || !s_actions || !q_states || !old_loc || !old_scale || !workspace) return pfail(p, MI_ERR_ARG, "null argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int T = tasks, B = batch;
  const size_t P = p->P;
  TrpoPlan pl; GenPlan gp;
  gen_plan(p, workspace, T, B, pl, gp);
  if (gp.bytes > workspace_bytes) return pfail(p, MI_ERR_WORKSPACE, "workspace too small: need " + std::to_string(gp.bytes));
  PCHK(p, hipMemsetAsync(gp.c, 0, (size_t)T * P * sizeof(float), st));
  GaussKlArgs gk{};
  gk.mu = pl.qa.mu; gk.rho = pl.thetap + p->o_sigma; gk.rstride = P; gk.old_loc = old_loc; gk.old_scale = old_scale; gk.count = q_count;
  gk.dmu = gp.k_dmu; gk.drho = gp.c + p->o_sigma; gk.gstride = P; gk.B = B; gk.A = p->A; gk.mode = KL_GRAD;
  hipLaunchKernelGGL(gauss_kl_kernel, dim3(T), dim3(256), 0, st, gk);
  PCHK(p, hipGetLastError());
  int rc = mlp_backward(p, st, T, B, q_states, pl.thetap, P, pl.qa, gp.k_dmu, gp.k_d2, gp.k_d1, gp.c, gp.k_pre2, gp.k_pre1);
  if (rc) return rc;
  rc = tan_sweep(p, st, pl, T, B, theta, s_states, s_actions, s_count, gp.c, P, gp.sc, gp.scr);
  if (rc || !kl_grad_out) return rc;
  rc = support_hvp(p, st, pl, T, B, theta, s_states, s_actions, s_count, gp.c, pl.hv);
  if (rc) return rc;
  hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, gp.c, P, pl.hv, inner_lr, (int)P, pl.tmpP);
  hipLaunchKernelGGL(mean_tasks_kernel, dim3(ceil_div((int)P, 256)), dim3(256), 0, st, pl.tmpP, T, (int)P, 1.f / (float)T,
                     (const float*)nullptr, 0.f, kl_grad_out);
  PCHK(p, hipGetLastError());
  return MI_OK;
}

// trpo.hessian_vector_product(mean KL, params, damping)(v) (rl.py:417) at the parameters of the preceding mi_trpo_surrogate +
// mi_trpo_kl_prepare on this workspace, exact for new != old (ANIL-TRPO, rl/anil_trpo.py:129).
extern "C" int mi_trpo_fvp_general(mi_policy* p, void* stream, const float* theta, const float* s_states, const float* s_actions,
                                   const int32_t* s_count, const float* q_states, const int32_t* q_count, const float* old_scale,
                                   int tasks, int batch, float inner_lr, float damping, const float* v, float* out, void* workspace,
                                   size_t workspace_bytes) {
  if (!p || !theta || !s_states || !s_actions || !q_states || !old_scale || !v || !out || !workspace) return pfail(p, MI_ERR_ARG, "null argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int T = tasks, B = batch;
  const size_t P = p->P, TB = (size_t)T * B;
  TrpoPlan pl; GenPlan gp;
  gen_plan(p, workspace, T, B, pl, gp);
  if (gp.bytes > workspace_bytes) return pfail(p, MI_ERR_WORKSPACE, "workspace too small");
  // ---- u_t = J_t v
  hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, v, (size_t)0, pl.tmpP, 0.f, (int)P, pl.u);
  PCHK(p, hipGetLastError());
  int rc = support_hvp(p, st, pl, T, B, theta, s_states, s_actions, s_count, pl.u, pl.hv);
  if (rc) return rc;
  hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, pl.u, P, pl.hv, inner_lr, (int)P, pl.u);
  PCHK(p, hipGetLastError());
  // ---- w_t = Hess KL_t(theta'_t) u_t: tangent forward on the query pass, exact output Hessian, tangent backward
  rc = mlp_tangent_forward(p, st, T, B, q_states, pl.thetap, P, pl.qa, pl.u, pl.ta);
  if (rc) return rc;
  PCHK(p, hipMemsetAsync(pl.w, 0, (size_t)T * P * sizeof(float), st));
  GaussKlArgs gh{};
  gh.mu = pl.qa.mu; gh.rho = pl.thetap + p->o_sigma; gh.rstride = P; gh.old_scale = old_scale; gh.mud = pl.ta.mu;
  gh.rhod = pl.u + p->o_sigma; gh.vstride = P; gh.count = q_count; gh.dmu = pl.rdmu; gh.drho = pl.w + p->o_sigma; gh.gstride = P;
  gh.B = B; gh.A = p->A; gh.mode = KL_HESS;
  hipLaunchKernelGGL(gauss_kl_kernel, dim3(T), dim3(256), 0, st, gh);
  PCHK(p, hipGetLastError());
  rc = mlp_tangent_backward(p, st, T, B, q_states, pl.thetap, P, pl.qa, pl.ta, pl.u, gp.k_dmu, gp.k_d2, gp.k_d1, gp.k_pre2, gp.k_pre1,
                            pl.rdmu, pl.r2, pl.r1, pl.w);
  if (rc) return rc;
  // ---- r_t = J_t^T w_t  (left in pl.w)
  rc = support_hvp(p, st, pl, T, B, theta, s_states, s_actions, s_count, pl.w, pl.hv);
  if (rc) return rc;
  hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, pl.w, P, pl.hv, inner_lr, (int)P, pl.w);
  PCHK(p, hipGetLastError());
  // ---- s3_t = T_t[v, c_t]: v-tangent sweep, then the mixed sweep
  rc = tan_sweep(p, st, pl, T, B, theta, s_states, s_actions, s_count, v, 0, gp.sv, gp.scr);
  if (rc) return rc;
  const TanSweep &C = gp.sc, &V = gp.sv;
  const float* c = gp.c;
  PCHK(p, hipMemsetAsync(gp.s3, 0, (size_t)T * P * sizeof(float), st));
  PCHK(p, ew(st, p->act, EW_ACT_CV, TB * p->H1, pl.sa.h1, gp.hcv1, nullptr, C.z1, V.z1));
  { FTerm tm[3] = {{V.h1, c + p->o_w2, P}, {C.h1, v + p->o_w2, 0}, {gp.hcv1, theta + p->o_w2, 0}};
    PCHK(p, dense_fwd_n(st, T, B, p->H1, p->H2, tm, 3, nullptr, 0, gp.zcv2)); }
  PCHK(p, ew(st, p->act, EW_ACT_CV, TB * p->H2, pl.sa.h2, gp.hcv2, gp.zcv2, C.z2, V.z2));
  { FTerm tm[3] = {{V.h2, c + p->o_w3, P}, {C.h2, v + p->o_w3, 0}, {gp.hcv2, theta + p->o_w3, 0}};
    PCHK(p, dense_fwd_n(st, T, B, p->H2, p->A, tm, 3, nullptr, 0, gp.mucv)); }
  GaussKlArgs g2{};
  g2.mu = pl.sa.mu; g2.rho = theta + p->o_sigma; g2.rstride = 0; g2.mud = gp.mucv; g2.muc = C.mu; g2.muv = V.mu;
  g2.rhoc = c + p->o_sigma; g2.cstride = P; g2.rhov = v + p->o_sigma; g2.vvstride = 0; g2.act = s_actions; g2.coef = pl.s_coef;
  g2.count = s_count; g2.dmu = gp.rdmu_cv; g2.drho = gp.s3 + p->o_sigma; g2.gstride = P; g2.B = B; g2.A = p->A; g2.mode = G_TAN2;
  hipLaunchKernelGGL(gauss_kl_kernel, dim3(T), dim3(256), 0, st, g2);
  PCHK(p, hipGetLastError());
  { WTerm tm[4] = {{gp.rdmu_cv, pl.sa.h2}, {C.rdmu, V.h2}, {V.rdmu, C.h2}, {pl.s_dmu, gp.hcv2}};
    PCHK(p, dense_bwd_w_n(st, T, B, p->H2, p->A, tm, 4, gp.s3 + p->o_w3, gp.s3 + p->o_b3, P)); }
  { FTerm tm[3] = {{gp.rdmu_cv, theta + p->o_w3, 0}, {C.rdmu, v + p->o_w3, 0}, {V.rdmu, c + p->o_w3, P}};
    PCHK(p, dense_bwd_x_n(st, T, B, p->H2, p->A, tm, 3, gp.dh2cv)); }
  PCHK(p, ew(st, p->act, EW_GATE_CV, TB * p->H2, pl.sa.h2, gp.dz2cv, gp.dh2cv, V.z2, C.dh2, C.z2, V.dh2, gp.zcv2, pl.s_pre2));
  { WTerm tm[4] = {{gp.dz2cv, pl.sa.h1}, {C.dz2, V.h1}, {V.dz2, C.h1}, {pl.s_d2, gp.hcv1}};
    PCHK(p, dense_bwd_w_n(st, T, B, p->H1, p->H2, tm, 4, gp.s3 + p->o_w2, gp.s3 + p->o_b2, P)); }
  { FTerm tm[3] = {{gp.dz2cv, theta + p->o_w2, 0}, {C.dz2, v + p->o_w2, 0}, {V.dz2, c + p->o_w2, P}};
    PCHK(p, dense_bwd_x_n(st, T, B, p->H1, p->H2, tm, 3, gp.dh1cv)); }
  PCHK(p, ew(st, p->act, EW_GATE_CV, TB * p->H1, pl.sa.h1, gp.dz1cv, gp.dh1cv, V.z1, C.dh1, C.z1, V.dh1, nullptr, pl.s_pre1));
  { WTerm tm[1] = {{gp.dz1cv, s_states}};
    PCHK(p, dense_bwd_w_n(st, T, B, p->S, p->H1, tm, 1, gp.s3 + p->o_w1, gp.s3 + p->o_b1, P)); }
  // ---- out = mean_t (r_t - lr s3_t) + damping v
  hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ceil_div((int)P, 256), T), dim3(256), 0, st, pl.w, P, gp.s3, inner_lr, (int)P, pl.tmpP);
  hipLaunchKernelGGL(mean_tasks_kernel, dim3(ceil_div((int)P, 256)), dim3(256), 0, st, pl.tmpP, T, (int)P, 1.f / (float)T, v, damping, out);
  PCHK(p, hipGetLastError());
  return MI_OK;
}

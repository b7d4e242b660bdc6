// Classifier head for a whole meta-batch: Linear(F, ways) + CrossEntropyLoss(reduction='mean') forward, backward and
// tangent, one workgroup per task (per-task fast weights).
// Replaces (reference): MiniImagenetCNN.forward's `self.linear(x.view(-1, 25*hidden))` (core_functions/vision_models.py:109),
// OmniglotCNN.forward's mean+linear (:53-54), `loss(learner(adapt_data), adapt_labels)` (core_functions/vision.py:11,16),
// `accuracy` (vision.py:21-23) and the autograd backward / double-backward of addmm + log_softmax + nll_loss.
// Features arrive in NHWC flatten order; the engine permutes linear.weight's columns once at the boundary so the result
// equals the reference's NCHW `view`.
#include "mi_common.h"
#include "kernels.h"

// logits[n][w] = bl[w] + sum_i f[n][i] * wl[w][i]  (+ tangent terms), one (n,w) pair per wave iteration.
__device__ __forceinline__ float wave_dot(const float* __restrict__ x, const float* __restrict__ y, int len, int lane) {
  float s = 0.f;
  for (int i = lane; i < len; i += 64) s = fmaf(x[i], y[i], s);
  return wave_sum(s);
}

template <bool WITH_GRAD>
__global__ __launch_bounds__(256) void head_fwd_bwd_kernel(HeadArgs a) {
  extern __shared__ float sm[];
  const int task = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.n, F = a.feat, WY = a.ways;
  float* s_logit = sm;              // [N][WY]
  float* s_dl = sm + N * WY;        // [N][WY]
  float* s_red = s_dl + N * WY;     // [2][N]
  const float* f_t = a.f + (size_t)task * N * F;
  const float* wl_t = a.wl + (size_t)task * a.pstride;
  const float* bl_t = a.bl + (size_t)task * a.pstride;
  const int32_t* y_t = a.y + (size_t)task * N;

  for (int pair = wave; pair < N * WY; pair += 4) {
    const int n = pair / WY, w = pair - n * WY;
    const float d = wave_dot(f_t + (size_t)n * F, wl_t + (size_t)w * F, F, lane);
    if (lane == 0) s_logit[pair] = d + bl_t[w];
  }
  __syncthreads();
  for (int n = tid; n < N; n += 256) {
    const float* l = s_logit + n * WY;
    float mx = l[0];
    int am = 0;
    for (int w = 1; w < WY; ++w)
      if (l[w] > mx) { mx = l[w]; am = w; }      // first maximal index (torch.argmax)
    float se = 0.f;
    for (int w = 0; w < WY; ++w) se += expf(l[w] - mx);
    const float lse = mx + logf(se);
    const int y = y_t[n];
    const float inv = 1.f / se, invn = 1.f / (float)N;
    for (int w = 0; w < WY; ++w) {
      const float p = expf(l[w] - mx) * inv;
      const float dl = (p - (w == y ? 1.f : 0.f)) * invn;
      s_dl[n * WY + w] = dl;
      const size_t o = ((size_t)task * N + n) * WY + w;
      if (a.prob) a.prob[o] = p;
      if (a.dl) a.dl[o] = dl;
      if (a.logits) a.logits[o] = l[w];
    }
    s_red[n] = lse - l[y];
    s_red[N + n] = (am == y) ? 1.f : 0.f;
  }
  __syncthreads();
  if (tid == 0) {
    float ls = 0.f, cs = 0.f;
    for (int n = 0; n < N; ++n) { ls += s_red[n]; cs += s_red[N + n]; }
    a.loss[task] = ls / (float)N;
    a.acc[task] = cs / (float)N;
  }
  if (!WITH_GRAD) return;
  float* dwl_t = a.dwl + (size_t)task * a.gstride;
  float* dbl_t = a.dbl + (size_t)task * a.gstride;
  // dwl[w][i] = sum_n dl[n][w] f[n][i]
  for (int e = tid; e < WY * F; e += 256) {
    const int w = e / F, i = e - w * F;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s = fmaf(s_dl[n * WY + w], f_t[(size_t)n * F + i], s);
    dwl_t[e] = s;
  }
  for (int w = tid; w < WY; w += 256) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += s_dl[n * WY + w];
    dbl_t[w] = s;
  }
  // df[n][i] = sum_w dl[n][w] wl[w][i]
  if (a.df) {
    float* df_t = a.df + (size_t)task * N * F;
    for (int e = tid; e < N * F; e += 256) {
      const int n = e / F, i = e - n * F;
      float s = 0.f;
      for (int w = 0; w < WY; ++w) s = fmaf(s_dl[n * WY + w], wl_t[(size_t)w * F + i], s);
      df_t[e] = s;
    }
  }
}

// Tangent: ld = fd wl^T + f wld^T + bld ; probd = prob (ld - <prob, ld>) ; R{dl} = probd / N
//          R{dwl} = R{dl}^T f + dl^T fd ; R{dbl} = sum_n R{dl} ; R{df} = R{dl} wl + dl wld
__global__ __launch_bounds__(256) void head_tangent_kernel(HeadArgs a) {
  extern __shared__ float sm[];
  const int task = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.n, F = a.feat, WY = a.ways;
  float* s_ld = sm;                 // [N][WY]  -> R{dl}
  float* s_dl = sm + N * WY;        // [N][WY]
  const float* f_t = a.f + (size_t)task * N * F;
  const bool has_fd = a.fd != nullptr;          // ANIL: features carry no tangent (only the head is adapted)
  const float* fd_t = has_fd ? a.fd + (size_t)task * N * F : f_t;
  const float* wl_t = a.wl + (size_t)task * a.pstride;
  const float* wld_t = a.wld + (size_t)task * a.vstride;
  const float* bld_t = a.bld + (size_t)task * a.vstride;
  for (int pair = wave; pair < N * WY; pair += 4) {
    const int n = pair / WY, w = pair - n * WY;
    float d = wave_dot(f_t + (size_t)n * F, wld_t + (size_t)w * F, F, lane);
    if (has_fd) d += wave_dot(fd_t + (size_t)n * F, wl_t + (size_t)w * F, F, lane);
    if (lane == 0) s_ld[pair] = d + bld_t[w];
  }
  for (int e = tid; e < N * WY; e += 256) s_dl[e] = a.dl[(size_t)task * N * WY + e];
  __syncthreads();
  for (int n = tid; n < N; n += 256) {
    const float* pr = a.prob + ((size_t)task * N + n) * WY;
    float dot = 0.f;
    for (int w = 0; w < WY; ++w) dot = fmaf(pr[w], s_ld[n * WY + w], dot);
    const float invn = 1.f / (float)N;
    for (int w = 0; w < WY; ++w) s_ld[n * WY + w] = pr[w] * (s_ld[n * WY + w] - dot) * invn;
  }
  __syncthreads();
  float* dwl_t = a.dwl + (size_t)task * a.gstride;
  float* dbl_t = a.dbl + (size_t)task * a.gstride;
  for (int e = tid; e < WY * F; e += 256) {
    const int w = e / F, i = e - w * F;
    float s = 0.f;
    for (int n = 0; n < N; ++n) {
      s = fmaf(s_ld[n * WY + w], f_t[(size_t)n * F + i], s);
      if (has_fd) s = fmaf(s_dl[n * WY + w], fd_t[(size_t)n * F + i], s);
    }
    dwl_t[e] = s;
  }
  for (int w = tid; w < WY; w += 256) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += s_ld[n * WY + w];
    dbl_t[w] = s;
  }
  if (a.df) {
    float* df_t = a.df + (size_t)task * N * F;
    for (int e = tid; e < N * F; e += 256) {
      const int n = e / F, i = e - n * F;
      float s = 0.f;
      for (int w = 0; w < WY; ++w) {
        s = fmaf(s_ld[n * WY + w], wl_t[(size_t)w * F + i], s);
        s = fmaf(s_dl[n * WY + w], wld_t[(size_t)w * F + i], s);
      }
      df_t[e] = s;
    }
  }
}

// OmniglotCNN: x.mean(dim=[2,3]) (vision_models.py:53).  rows = T*N, p [rows][hw][c] -> f [rows][c]; linear, so the
// tangent uses the same kernel and the backward is a broadcast / hw.
__global__ void spatial_mean_kernel(const float* __restrict__ p, float* __restrict__ f, int rows, int hw, int c) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)rows * c) return;
  const size_t row = e / c, ch = e - row * c;
  float s = 0.f;
  for (int k = 0; k < hw; ++k) s += p[(row * hw + k) * c + ch];
  f[e] = s / (float)hw;
}
__global__ void spatial_mean_bwd_kernel(const float* __restrict__ df, float* __restrict__ dp, int rows, int hw, int c) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)rows * hw * c) return;
  const size_t ch = e % c, row = e / ((size_t)hw * c);
  dp[e] = df[row * c + ch] / (float)hw;
}

hipError_t launch_head_fwd_bwd(hipStream_t st, const HeadArgs& a, int tasks, int with_grad) {
  const size_t sm = (size_t)(2 * a.n * a.ways + 2 * a.n) * sizeof(float);
  if (with_grad) hipLaunchKernelGGL(head_fwd_bwd_kernel<true>, dim3(tasks), dim3(256), sm, st, a);
  else hipLaunchKernelGGL(head_fwd_bwd_kernel<false>, dim3(tasks), dim3(256), sm, st, a);
  return hipGetLastError();
}
hipError_t launch_head_tangent(hipStream_t st, const HeadArgs& a, int tasks) {
  const size_t sm = (size_t)(2 * a.n * a.ways) * sizeof(float);
  hipLaunchKernelGGL(head_tangent_kernel, dim3(tasks), dim3(256), sm, st, a);
  return hipGetLastError();
}
hipError_t launch_spatial_mean(hipStream_t st, const float* p, float* f, int rows, int hw, int c) {
  const size_t n = (size_t)rows * c;
  hipLaunchKernelGGL(spatial_mean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, f, rows, hw, c);
  return hipGetLastError();
}
hipError_t launch_spatial_mean_bwd(hipStream_t st, const float* df, float* dp, int rows, int hw, int c) {
  const size_t n = (size_t)rows * hw * c;
  hipLaunchKernelGGL(spatial_mean_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, df, dp, rows, hw, c);
  return hipGetLastError();
}

// Classifier head for a whole meta-batch: Linear(F, ways) + CrossEntropyLoss(reduction='mean') forward, backward and
// tangent (per-task fast weights).
// Replaces (reference): MiniImagenetCNN.forward's `self.linear(x.view(-1, 25*hidden))` (core_functions/vision_models.py:109),
// OmniglotCNN.forward's mean+linear (:53-54), `loss(learner(adapt_data), adapt_labels)` (core_functions/vision.py:11,16),
// `accuracy` (vision.py:21-23) and the autograd backward / double-backward of addmm + log_softmax + nll_loss.
// Features arrive in NHWC flatten order; the engine permutes linear.weight's columns once at the boundary so the result
// equals the reference's NCHW `view`.
#include "mi_common.h"
#include "kernels.h"
#include "head_bodies.h"

template <bool TANGENT>
__global__ __launch_bounds__(256) void head_rows_kernel(HeadArgs a) {
  const int task = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.y * 4 + wave;
  if (n >= a.n) return;
  head_row<TANGENT>(a, task, n, lane);
}

template <bool TANGENT>
__global__ __launch_bounds__(256) void head_grads_kernel(HeadArgs a) {
  extern __shared__ float sm[];
  const int task = blockIdx.x, tid = threadIdx.x;
  const int N = a.n, WY = a.ways;
  float* s_a = sm;               // [N][WY]: dl (primal) or R{dl} (tangent)
  float* s_b = sm + N * WY;      // [N][WY]: dl (tangent only)
  head_stage_dl<TANGENT>(a, task, tid, 256, s_a, s_b);
  __syncthreads();
  float* s_red = sm + (TANGENT ? 2 : 1) * N * WY;   // [3][8][64] partial sums of groups 1..3
  head_grads_chunk<TANGENT>(a, task, blockIdx.y, tid, s_a, s_b, s_red);
  if (blockIdx.y == 0) head_task_sums<TANGENT>(a, task, tid, 256, s_a);
}

// loss[t] = mean_n rowloss, acc[t] = mean_n rowhit (fixed order)
__global__ void head_reduce_kernel(const float* __restrict__ rowloss, const float* __restrict__ rowhit, int tasks, int n,
                                   float* __restrict__ loss, float* __restrict__ acc) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= tasks) return;
  float ls = 0.f, cs = 0.f;
  for (int k = 0; k < n; ++k) { ls += rowloss[(size_t)t * n + k]; cs += rowhit[(size_t)t * n + k]; }
  loss[t] = ls / (float)n;
  acc[t] = cs / (float)n;
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
  if (a.ways > 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(head_rows_kernel<false>, dim3(tasks, ceil_div(a.n, 4)), dim3(256), 0, st, a);
  if (with_grad) {     // the gradient launch also folds the row losses / hits into loss[t], acc[t]
    const size_t sm = (size_t)(a.n * a.ways + 3 * 8 * 64) * sizeof(float);
    hipLaunchKernelGGL(head_grads_kernel<false>, dim3(tasks, ceil_div(a.feat, 64)), dim3(256), sm, st, a);
  } else {
    hipLaunchKernelGGL(head_reduce_kernel, dim3(ceil_div(tasks, 64)), dim3(64), 0, st, a.rowloss, a.rowhit, tasks, a.n, a.loss, a.acc);
  }
  return hipGetLastError();
}
hipError_t launch_head_tangent(hipStream_t st, const HeadArgs& a, int tasks) {
  if (a.ways > 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(head_rows_kernel<true>, dim3(tasks, ceil_div(a.n, 4)), dim3(256), 0, st, a);
  const size_t sm = (size_t)(2 * a.n * a.ways + 3 * 8 * 64) * sizeof(float);
  hipLaunchKernelGGL(head_grads_kernel<true>, dim3(tasks, ceil_div(a.feat, 64)), dim3(256), sm, st, a);
  return hipGetLastError();
}
// Backward of the linear head from caller-supplied dlogits (a.dl): dWl, dbl, df.  Used by the step-wise learner whose loss
// is computed outside the engine (rc_vision.py:68-70 scales it, cl_vision.py:58-59 does not).
hipError_t launch_head_grads(hipStream_t st, const HeadArgs& a, int tasks) {
  const size_t sm = (size_t)(a.n * a.ways + 3 * 8 * 64) * sizeof(float);
  hipLaunchKernelGGL(head_grads_kernel<false>, dim3(tasks, ceil_div(a.feat, 64)), dim3(256), sm, st, a);
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

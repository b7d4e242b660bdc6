// Boundary / bookkeeping kernels: the reference's prepare_batch split, parameter-layout gather/scatter, fast-weight SGD
// update, Adam.
#include "mi_common.h"
#include "kernels.h"

// utils/data_pre.py:115-129 (reference): support = rows {0,2,4,...} of the task batch, query = the complement (odd rows),
// order preserved; here also NCHW -> NHWC and int64 -> int32 labels.  One thread per output pixel (all channels).
__global__ void prepare_batch_kernel(const float* __restrict__ data, const int64_t* __restrict__ labels, int tasks, int n2,
                                     int c, int h, int w, float* __restrict__ xs, float* __restrict__ xq,
                                     int32_t* __restrict__ ys, int32_t* __restrict__ yq) {
  const size_t hw = (size_t)h * w;
  const size_t total = (size_t)tasks * n2 * hw;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const size_t pix = e % hw;
  const size_t img = e / hw;  // task*n2 + row
  const size_t task = img / n2, row = img - task * n2;
  const size_t half = row >> 1;
  float* dst = ((row & 1) ? xq : xs) + ((task * (n2 >> 1) + half) * hw + pix) * c;
  const float* src = data + img * c * hw + pix;
  for (int ch = 0; ch < c; ++ch) dst[ch] = src[(size_t)ch * hw];
  if (pix == 0) {
    int32_t* yd = (row & 1) ? yq : ys;
    yd[task * (n2 >> 1) + half] = (int32_t)labels[img];
  }
}

// NCHW -> NHWC for plain forward calls (learner(x)): one thread per pixel.
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, size_t images, int c, int h, int w, float* __restrict__ dst) {
  const size_t hw = (size_t)h * w;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= images * hw) return;
  const size_t pix = e % hw, img = e / hw;
  for (int ch = 0; ch < c; ++ch) dst[e * c + ch] = src[(img * c + ch) * hw + pix];
}

// Task sampling from a dataset resident in HBM (replaces learn2learn TaskDataset.sample() + LoadData for a whole meta-batch,
// utils/data_pre.py:16-112): out[row] = image index[row] of the dataset, optionally rotated by rot[row] quarter turns
// counter-clockwise (RandomClassRotation, data_pre.py:34,46,58; the host draws one angle per class).  U8 = dataset stored as
// bytes (Mini-ImageNet's raw 0..255 pixels: 4x less HBM footprint and gather traffic), converted to fp32 exactly.
template <bool U8>
__global__ void sample_tasks_kernel(const void* __restrict__ dataset, const int64_t* __restrict__ index,
                                    const uint8_t* __restrict__ rot, size_t rows, int c, int h, int w, float* __restrict__ out) {
  const size_t img = (size_t)c * h * w;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // one thread per 4 consecutive output floats
  const size_t per = img / 4;
  if (e >= rows * per) return;
  const size_t row = e / per, o4 = (e - row * per) * 4;
  const size_t src = (size_t)index[row] * img;
  const int r = rot ? (rot[row] & 3) : 0;
  float v[4];
  if (r == 0) {
    if (U8) {
      const uchar4 b = *reinterpret_cast<const uchar4*>(static_cast<const uint8_t*>(dataset) + src + o4);
      v[0] = (float)b.x; v[1] = (float)b.y; v[2] = (float)b.z; v[3] = (float)b.w;
    } else {
      const float4 f = *reinterpret_cast<const float4*>(static_cast<const float*>(dataset) + src + o4);
      v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    }
  } else {
    const size_t hw = (size_t)h * w;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t o = o4 + k, ch = o / hw, pix = o - ch * hw;
      const int i = (int)(pix / w), j = (int)(pix - (size_t)i * w);
      int si, sj;                                   // h == w is checked by the launcher
      if (r == 1) { si = j; sj = w - 1 - i; }
      else if (r == 2) { si = h - 1 - i; sj = w - 1 - j; }
      else { si = h - 1 - j; sj = i; }
      const size_t so = src + ch * hw + (size_t)si * w + sj;
      v[k] = U8 ? (float)static_cast<const uint8_t*>(dataset)[so] : static_cast<const float*>(dataset)[so];
    }
  }
  *reinterpret_cast<float4*>(out + row * img + o4) = make_float4(v[0], v[1], v[2], v[3]);
}

// ANIL: prepare_batch's even/odd split applied to feature rows (data_pre.py:122-127 after :118-119) and its transpose.
__global__ void split_rows_kernel(const float* __restrict__ src, size_t rows2, int f, float* __restrict__ even,
                                  float* __restrict__ odd) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows2 * f) return;
  const size_t row = e / f, col = e - row * f;
  ((row & 1) ? odd : even)[(row >> 1) * f + col] = src[e];
}
__global__ void interleave_rows_kernel(const float* __restrict__ even, const float* __restrict__ odd, size_t rows2, int f,
                                       float* __restrict__ dst) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows2 * f) return;
  const size_t row = e / f, col = e - row * f;
  dst[e] = ((row & 1) ? odd : even)[(row >> 1) * f + col];
}
__global__ void split_labels_kernel(const int64_t* __restrict__ labels, size_t rows2, int32_t* __restrict__ ys,
                                    int32_t* __restrict__ yq) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows2) return;
  ((e & 1) ? yq : ys)[e >> 1] = (int32_t)labels[e];
}

// theta_eng[t][i] = theta_ref[perm[i]] for every task (learn2learn clone_module: each task starts from the meta-parameters)
__global__ void gather_params_kernel(const float* __restrict__ theta_ref, size_t src_stride, const int32_t* __restrict__ perm,
                                     int p, int pstride, float* __restrict__ theta_eng) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pstride) return;
  theta_eng[(size_t)blockIdx.y * pstride + i] = i < p ? theta_ref[(size_t)blockIdx.y * src_stride + perm[i]] : 0.f;
}

// out_ref[t][perm[i]] = g[t][i]: per-task gradients back in the reference's parameter order (step-wise learner.adapt).
__global__ void scatter_tasks_kernel(const float* __restrict__ g, const int32_t* __restrict__ perm, int p, int pstride,
                                     float* __restrict__ out_ref) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p) return;
  out_ref[(size_t)blockIdx.y * p + perm[i]] = g[(size_t)blockIdx.y * pstride + i];
}

// NHWC -> NCHW for representations handed back to the caller (get_rep / get_rep_i): one thread per pixel.
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, size_t images, int c, int h, int w, float* __restrict__ dst) {
  const size_t hw = (size_t)h * w;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= images * hw) return;
  const size_t pix = e % hw, img = e / hw;
  for (int ch = 0; ch < c; ++ch) dst[(img * c + ch) * hw + pix] = src[e * c + ch];
}

// out_ref[perm[i]] = sum_t lam[t][i]   (eval_loss.backward() accumulates over tasks, maml_vision.py:112); fixed task order.
__global__ void scatter_sum_kernel(const float* __restrict__ lam, const int32_t* __restrict__ perm, int p, int pstride,
                                   int tasks, float* __restrict__ out_ref) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p) return;
  float s = 0.f;
  for (int t = 0; t < tasks; ++t) s += lam[(size_t)t * pstride + i];
  out_ref[perm[i]] = s;
}

// out = a - alpha*b : learn2learn maml_update (p <- p - lr*g) and the adjoint recursion lam <- lam - alpha*H lam.
__global__ void axpy_kernel(const float* __restrict__ a, const float* __restrict__ b, float alpha, size_t n,
                            float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] - alpha * b[i];
}

// Streaming copy: the build's own measurement of the HBM roofline (bench.py times it in the same run as the engine kernels;
// SURVEY.md 8d "measured HBM roofline").  One workgroup per CU, 4 x 16 bytes per lane in flight, non-temporal both ways:
// the fastest of the grid x unroll x temporal variants measured on MI355X (6.08 TB/s; hipMemcpy D2D 5.18 TB/s).
__global__ __launch_bounds__(256) void stream_copy_kernel(const floatx4* __restrict__ src, floatx4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    floatx4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}

// torch.optim.Adam (defaults, no weight decay / amsgrad) on the flat meta-parameters; grad scaled by 1/meta_batch first
// (maml_vision.py:139-141).
__global__ void adam_kernel(float* __restrict__ theta, const float* __restrict__ grad, float* __restrict__ m,
                            float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float gscale,
                            float bc1, float bc2_sqrt) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = grad[i] * gscale;
  const float mi = b1 * m[i] + (1.f - b1) * g;
  const float vi = b2 * v[i] + (1.f - b2) * g * g;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  theta[i] -= (lr / bc1) * (mi / denom);
}

hipError_t launch_prepare_batch(hipStream_t st, const float* data, const int64_t* labels, int tasks, int n2, int c, int h,
                                int w, float* xs, float* xq, int32_t* ys, int32_t* yq) {
  const size_t total = (size_t)tasks * n2 * h * w;
  hipLaunchKernelGGL(prepare_batch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, data, labels, tasks, n2, c,
                     h, w, xs, xq, ys, yq);
  return hipGetLastError();
}
hipError_t launch_nchw_to_nhwc(hipStream_t st, const float* src, size_t images, int c, int h, int w, float* dst) {
  const size_t total = images * h * w;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, images, c, h, w, dst);
  return hipGetLastError();
}
hipError_t launch_sample_tasks(hipStream_t st, const void* dataset, int u8, const int64_t* index, const uint8_t* rot, size_t rows,
                               int c, int h, int w, float* out) {
  const size_t img = (size_t)c * h * w;
  if (img % 4 != 0 || (rot && h != w)) return hipErrorInvalidValue;
  const size_t total = rows * (img / 4);
  if (u8)
    hipLaunchKernelGGL(sample_tasks_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dataset, index, rot, rows, c, h, w, out);
  else
    hipLaunchKernelGGL(sample_tasks_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dataset, index, rot, rows, c, h, w, out);
  return hipGetLastError();
}
hipError_t launch_split_rows(hipStream_t st, const float* src, int tasks, int n2, int f, float* even, float* odd) {
  const size_t total = (size_t)tasks * n2 * f;   // n2 is even, so global row parity == row parity within the task
  hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, (size_t)tasks * n2, f, even, odd);
  return hipGetLastError();
}
hipError_t launch_interleave_rows(hipStream_t st, const float* even, const float* odd, int tasks, int n, int f, float* dst) {
  const size_t total = (size_t)tasks * 2 * n * f;
  hipLaunchKernelGGL(interleave_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, even, odd,
                     (size_t)tasks * 2 * n, f, dst);
  return hipGetLastError();
}
hipError_t launch_split_labels(hipStream_t st, const int64_t* labels, int tasks, int n2, int32_t* ys, int32_t* yq) {
  const size_t total = (size_t)tasks * n2;
  hipLaunchKernelGGL(split_labels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, labels, total, ys, yq);
  return hipGetLastError();
}
hipError_t launch_gather_params(hipStream_t st, const float* theta_ref, size_t src_stride, const int32_t* perm, int p, int pstride,
                                int tasks, float* theta_eng) {
  hipLaunchKernelGGL(gather_params_kernel, dim3(ceil_div(pstride, 256), tasks), dim3(256), 0, st, theta_ref, src_stride, perm, p,
                     pstride, theta_eng);
  return hipGetLastError();
}
hipError_t launch_scatter_tasks(hipStream_t st, const float* g, const int32_t* perm, int p, int pstride, int tasks, float* out_ref) {
  hipLaunchKernelGGL(scatter_tasks_kernel, dim3(ceil_div(p, 256), tasks), dim3(256), 0, st, g, perm, p, pstride, out_ref);
  return hipGetLastError();
}
hipError_t launch_nhwc_to_nchw(hipStream_t st, const float* src, size_t images, int c, int h, int w, float* dst) {
  const size_t total = images * h * w;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, images, c, h, w, dst);
  return hipGetLastError();
}
hipError_t launch_scatter_sum(hipStream_t st, const float* lam, const int32_t* perm, int p, int pstride, int tasks,
                              float* out_ref) {
  hipLaunchKernelGGL(scatter_sum_kernel, dim3(ceil_div(p, 256)), dim3(256), 0, st, lam, perm, p, pstride, tasks, out_ref);
  return hipGetLastError();
}
hipError_t launch_axpy(hipStream_t st, const float* a, const float* b, float alpha, size_t n, float* out) {
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, alpha, n, out);
  return hipGetLastError();
}
hipError_t launch_stream_copy(hipStream_t st, const void* src, void* dst, size_t bytes) {
  if (bytes % 16) return hipErrorInvalidValue;
  hipLaunchKernelGGL(stream_copy_kernel, dim3(256), dim3(256), 0, st, static_cast<const floatx4*>(src), static_cast<floatx4*>(dst), bytes / 16);
  return hipGetLastError();
}
hipError_t launch_adam(hipStream_t st, float* theta, const float* grad, float* m, float* v, size_t n, int step, float lr,
                       float b1, float b2, float eps, float gscale) {
  const float bc1 = (float)(1.0 - pow((double)b1, (double)step));
  const float bc2 = (float)sqrt(1.0 - pow((double)b2, (double)step));
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, theta, grad, m, v, n, lr, b1, b2, eps,
                     gscale, bc1, bc2);
  return hipGetLastError();
}

// Batch mean and biased batch variance of every block of one forward pass, for the host-side running-statistics update
// (mi_engine_set_bn_export): var = 1/rstd^2 - eps.
__global__ void bn_export_kernel(BnExportArgs a) {
  const int t = blockIdx.x;
  for (int i = threadIdx.x; i < a.ctot; i += blockDim.x) {
    int l = 0;
    while (l + 1 < a.nl && i >= a.off[l + 1]) ++l;
    const int c = i - a.off[l];
    const float r = a.rstd[l][(size_t)t * a.c[l] + c];
    a.out[((size_t)t * 2 + 0) * a.ctot + i] = a.mu[l][(size_t)t * a.c[l] + c];
    a.out[((size_t)t * 2 + 1) * a.ctot + i] = 1.f / (r * r) - (float)MI_BN_EPS;
  }
}
hipError_t launch_bn_export(hipStream_t st, const BnExportArgs& a, int tasks) {
  hipLaunchKernelGGL(bn_export_kernel, dim3(tasks), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Largest magnitude of a tensor per task -- its exponent field, as fp32 bits -- in the task's cell (mi_common.h: MI_CELL_WORDS words per task; zeroed by the caller; max over
// unsigned bit patterns of |x|: order-independent, NaN and infinity sort on top).  The two-plane fp16 operand form of the convolutions (bf16_split.h) takes its
// per-(task, tensor) scale from these cells.  Inside the engine the tensors' PRODUCERS write them (one v_max per value, one atomic per
// wave); this kernel serves tensors that come from elsewhere (the standalone operator entries, kernels without the hook).
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, size_t per_task, unsigned* __restrict__ cell) {
  const int task = blockIdx.y;
  const float* xt = x + (size_t)task * per_task;
  unsigned m = 0u;
  const size_t n4 = (((uintptr_t)xt & 15u) == 0u) ? per_task / 4 : 0;
  const floatx4* x4 = reinterpret_cast<const floatx4*>(xt);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const floatx4 v = x4[i];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const unsigned b = __float_as_uint(v[q]) & 0x7fffffffu; m = b > m ? b : m; }
  }
  if (blockIdx.x == 0)
    for (size_t i = n4 * 4 + threadIdx.x; i < per_task; i += 256) { const unsigned b = __float_as_uint(xt[i]) & 0x7fffffffu; m = b > m ? b : m; }
  mi_amax_commit(m, cell, task);
}
hipError_t launch_amax(hipStream_t st, const float* x, size_t per_task, int tasks, unsigned* cell) {
  long blocks = (long)((per_task / 4 + 2047) / 2048);         // ~8 float4 per thread
  if (blocks < 1) blocks = 1;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks, tasks), dim3(256), 0, st, x, per_task, cell);
  return hipGetLastError();
}

// Standalone operator entry points (no engine plan behind them): a per-device scratch of cells, [8 slots][256 tasks] cells (mi_common.h: 16 sub-cells of one 256-byte line each), filled by
// launch_amax on the caller's stream.  Returns nullptr -- the launch then takes the bf16 form -- unless the fp16 form is selected.
// One stream at a time per device (the unit tests' use).
const unsigned* standalone_amax(hipStream_t st, int slot, const float* x, size_t per_task, int tasks, hipError_t* err) {
  *err = hipSuccess;
  constexpr int kSlots = 8, kTasks = 256;
  if (conv_operand_form() != 2 || !x || tasks > kTasks || slot < 0 || slot >= kSlots) return nullptr;
  static unsigned* g_cells[64] = {};
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return nullptr;
  if (!g_cells[d] && hipMalloc(&g_cells[d], (size_t)kSlots * kTasks * MI_CELL_WORDS * sizeof(unsigned)) != hipSuccess) { g_cells[d] = nullptr; return nullptr; }
  unsigned* c = g_cells[d] + (size_t)slot * kTasks * MI_CELL_WORDS;
  *err = hipMemsetAsync(c, 0, (size_t)tasks * MI_CELL_WORDS * sizeof(unsigned), st);
  if (*err == hipSuccess) *err = launch_amax(st, x, per_task, tasks, c);
  return *err == hipSuccess ? c : nullptr;
}

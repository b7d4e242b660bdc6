// Kernel-level entry points of the tangent (R-operator) and fused block-1 kernels for the unit parity tests
// (tests/test_gpu_tangent_kernels.py against oracle/kernels_ref.py).  Thin argument plumbing only: every function fills the
// launcher structs of kernels.h exactly as engine.hip does and runs the same launch sequence (kernel + bn_finalize).
#include <string>
#include "mi_common.h"
#include "kernels.h"
#include "../../include/mi_maml.h"

#define EPI_NONE 0
#define EPI_STATS 1
#define EPI_TSTATS 2

int mi_internal_fail(int code, const char* msg);   // engine.hip: sets the global error string

#define TCHK(call)                                                                                           \
  do {                                                                                                       \
    hipError_t _s = (call);                                                                                  \
    if (_s != hipSuccess) return mi_internal_fail(MI_ERR_HIP, (std::string(#call) + ": " + hipGetErrorString(_s)).c_str()); \
  } while (0)

static hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" int mi_conv3x3_tangent(void* stream, const float* x0, const float* w0, const float* x1, const float* w1, size_t pstride,
                                  const float* z, const float* mu, const float* rstd, int tasks, int n, int h, int wd, int ci, int co,
                                  int stride, float* zd, float* m1, float* m2, void* scratch, size_t scratch_bytes) {
  if (!x0 || !w0 || !z || !mu || !rstd || !zd || !m1 || !m2 || !scratch || (x1 && !w1))
    return mi_internal_fail(MI_ERR_ARG, "mi_conv3x3_tangent: null argument");
  ConvArgs ca{};
  ca.in[0] = x0; ca.wt[0] = w0; ca.in[1] = x1; ca.wt[1] = w1; ca.wstride = pstride;
  ca.out = zd; ca.z = z; ca.mu = mu; ca.rstd = rstd;
  ca.partial = reinterpret_cast<double*>(scratch);
  ca.g = ConvGeom{n, h, wd, (h - 1) / stride + 1, (wd - 1) / stride + 1, ci, co, stride};
  ca.mpix = n * ca.g.ho * ca.g.wo;
  if ((size_t)tasks * conv_max_blocks_per_task(ca.g) * 2 * co * sizeof(double) > scratch_bytes)
    return mi_internal_fail(MI_ERR_WORKSPACE, "mi_conv3x3_tangent: scratch too small");
  // (fp16 operand form: the operands' largest magnitudes, which the engine's producers leave behind, from a reduction launch here)
  hipError_t aerr = hipSuccess;
  ca.amax[0] = standalone_amax(S(stream), 0, x0, (size_t)n * h * wd * ci, tasks, &aerr); TCHK(aerr);
  if (x1) { ca.amax[1] = standalone_amax(S(stream), 1, x1, (size_t)n * h * wd * ci, tasks, &aerr); TCHK(aerr); }
  int blk = 0;
  TCHK(launch_conv3x3(S(stream), ca, tasks, x1 ? 2 : 1, EPI_TSTATS, 0, &blk));
  TCHK(launch_bn_finalize(S(stream), ca.partial, blk, tasks, co, 1.0 / (double)ca.mpix, FIN_TSTATS, m1, co, m2, co));
  return MI_OK;
}

extern "C" int mi_conv3x3_bwd2(void* stream, const float* x0, const float* dz0, const float* x1, const float* dz1, const float* w0,
                               const float* w1, size_t pstride, int tasks, int n, int h, int wd, int ci, int co, int stride,
                               float* dx, float* dw9, size_t gstride, void* scratch, size_t scratch_bytes) {
  if (!x0 || !dz0 || !x1 || !dz1 || !dw9 || !scratch || (dx && (!w0 || !w1)))
    return mi_internal_fail(MI_ERR_ARG, "mi_conv3x3_bwd2: null argument");
  const int ho = (h - 1) / stride + 1, wo = (wd - 1) / stride + 1;
  WgradArgs wa{};
  // R{dW} = wgrad(x, R{dz}) + wgrad(xd, dz): term 0 pairs x0 with dz0, term 1 pairs x1 with dz1
  wa.x[0] = x0; wa.dz[0] = dz0; wa.x[1] = x1; wa.dz[1] = dz1;
  wa.partial = reinterpret_cast<float*>(scratch);
  wa.g = ConvGeom{n, h, wd, ho, wo, ci, co, stride};
  wa.mpix = n * ho * wo;
  if (wgrad_partial_floats(wa.g, tasks) * sizeof(float) > scratch_bytes)
    return mi_internal_fail(MI_ERR_WORKSPACE, "mi_conv3x3_bwd2: scratch too small");
  hipError_t aerr = hipSuccess;
  wa.amax_x[0] = standalone_amax(S(stream), 0, x0, (size_t)n * h * wd * ci, tasks, &aerr); TCHK(aerr);
  wa.amax_dz[0] = standalone_amax(S(stream), 1, dz0, (size_t)n * ho * wo * co, tasks, &aerr); TCHK(aerr);
  wa.amax_x[1] = standalone_amax(S(stream), 2, x1, (size_t)n * h * wd * ci, tasks, &aerr); TCHK(aerr);
  wa.amax_dz[1] = standalone_amax(S(stream), 3, dz1, (size_t)n * ho * wo * co, tasks, &aerr); TCHK(aerr);
  int nch = 0;
  TCHK(launch_wgrad3x3(S(stream), wa, tasks, 2, &nch));
  TCHK(launch_wgrad_reduce(S(stream), wa.partial, nch, 9 * ci * co, tasks, dw9, gstride));
  if (dx) {
    ConvArgs ca{};
    // R{dx} = dgrad(dz0, w0) + dgrad(dz1, w1)
    ca.in[0] = dz0; ca.wt[0] = w0; ca.in[1] = dz1; ca.wt[1] = w1; ca.wstride = pstride; ca.out = dx;
    ca.amax[0] = wa.amax_dz[0]; ca.amax[1] = wa.amax_dz[1];
    ca.g = ConvGeom{n, ho, wo, h, wd, co, ci, stride};
    ca.mpix = n * h * wd;
    TCHK(launch_conv3x3(S(stream), ca, tasks, 2, EPI_NONE, 1, nullptr));
  }
  return MI_OK;
}

static BnArgs bn_args(const mi_bn_tangent_args* a) {
  BnArgs b{};
  b.z = a->z; b.zd = a->zd; b.mu = a->mu; b.rstd = a->rstd; b.m1 = a->m1; b.m2 = a->m2;
  b.gamma = a->gamma; b.beta = a->beta; b.pstride = a->pstride;
  b.gammad = a->gammad; b.betad = a->betad; b.vstride = a->vstride;
  b.dgamma = a->dgamma; b.dbeta = a->dbeta; b.gstride = a->gstride;
  b.dp = a->dp; b.dpd = a->dpd;
  b.n = a->n; b.ho = a->ho; b.wo = a->wo; b.c = a->c;
  b.inv_m = 1.f / (float)(a->n * a->ho * a->wo);
  return b;
}

extern "C" int mi_bn_tangent_fwd(void* stream, const mi_bn_tangent_args* a, float* pd) {
  if (!a || !pd || !a->z || !a->zd || !a->mu || !a->rstd || !a->m1 || !a->m2 || !a->gamma || !a->beta || !a->gammad || !a->betad)
    return mi_internal_fail(MI_ERR_ARG, "mi_bn_tangent_fwd: null argument");
  BnArgs b = bn_args(a);
  b.out = pd;
  TCHK(launch_bn_tan_fwd(S(stream), b, a->tasks, a->pool));
  return MI_OK;
}

extern "C" int mi_bn_tangent_bwd(void* stream, const mi_bn_tangent_args* a, float* rdgamma, float* rdbeta, size_t hstride,
                                 float* rdz, void* scratch, size_t scratch_bytes) {
  if (!a || !rdgamma || !rdbeta || !rdz || !scratch || !a->dp || !a->dpd || !a->dgamma || !a->dbeta)
    return mi_internal_fail(MI_ERR_ARG, "mi_bn_tangent_bwd: null argument");
  BnArgs b = bn_args(a);
  b.partial = reinterpret_cast<double*>(scratch);
  if ((size_t)a->tasks * bn_blocks_per_task(a->n, a->ho, a->wo, a->c, a->pool, a->tasks) * 2 * a->c * sizeof(double) > scratch_bytes)
    return mi_internal_fail(MI_ERR_WORKSPACE, "mi_bn_tangent_bwd: scratch too small");
  int blk = 0;
  TCHK(launch_bn_tan_bwd_reduce(S(stream), b, a->tasks, a->pool, &blk));
  TCHK(launch_bn_finalize(S(stream), b.partial, blk, a->tasks, a->c, 1.0, FIN_SUMS, rdgamma, hstride, rdbeta, hstride));
  b.rdgamma = rdgamma; b.rdbeta = rdbeta; b.hstride = hstride; b.out = rdz;
  TCHK(launch_bn_tan_bwd_apply(S(stream), b, a->tasks, a->pool));
  return MI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused block 1 (block1.hip, gram.hip, pooled_reduce of bn_pool.hip)
static B1Args b1_from(const mi_block1_args* a) {
  B1Args b{};
  b.x = a->x; b.w = a->w; b.wd = a->wd; b.wstride = a->pstride; b.vstride = a->vstride;
  b.mu = a->mu; b.rstd = a->rstd; b.m1 = a->m1; b.m2 = a->m2;
  b.gamma = a->gamma; b.beta = a->beta; b.pstride = a->pstride;
  b.gammad = a->gammad; b.betad = a->betad;
  b.dgamma = a->dgamma; b.dbeta = a->dbeta; b.gstride = a->gstride;
  b.rdgamma = a->rdgamma; b.rdbeta = a->rdbeta; b.hstride = a->hstride;
  b.dp = a->dp; b.dpd = a->dpd;
  b.n = a->n; b.hh = a->h; b.ww = a->w_; b.co = a->co;
  b.inv_m = 1.f / (float)(a->n * a->h * a->w_);
  return b;
}

extern "C" size_t mi_block1_scratch_bytes(int tasks, int n, int h, int w, int ci, int co) {
  int bpt = block1_blocks_per_task(n, h, w, co, tasks);
  const int sb = sparse_wgrad_blocks_per_task(n, h, w, co, tasks);
  if (sb > bpt) bpt = sb;
  const int pr = pooled_reduce_blocks(n * (h / 2) * (w / 2), co, tasks);
  const size_t wg = (size_t)tasks * bpt * 9 * ci * co * sizeof(float);
  const size_t red = (size_t)tasks * (bpt > pr ? bpt : pr) * 2 * co * sizeof(double);
  return align_up(wg, 256) + align_up(red, 256);
}

// mode = one of MI_B1_* (= the B1_* launch modes).  Reduction modes write (out0, out1) [tasks][ostride] through bn_finalize;
// forward modes write p_out (+ zh_out / arg_out); weight-gradient modes write dW [tasks][ostride] (9*ci*co per task).
extern "C" int mi_block1_run(void* stream, int mode, const mi_block1_args* a, float* p_out, float* zh_out, uint8_t* arg_out,
                             float* out0, float* out1, size_t ostride, void* scratch, size_t scratch_bytes) {
  if (!a || !a->x || !a->w) return mi_internal_fail(MI_ERR_ARG, "mi_block1_run: null argument");
  if (!block1_supported(a->ci, 1, 1, a->h, a->w_, a->co)) return mi_internal_fail(MI_ERR_ARG, "mi_block1_run: geometry not supported by the fused block-1 kernels");
  if (scratch_bytes < mi_block1_scratch_bytes(a->tasks, a->n, a->h, a->w_, a->ci, a->co)) return mi_internal_fail(MI_ERR_WORKSPACE, "mi_block1_run: scratch too small");
  B1Args b = b1_from(a);
  {   // carve: [weight-gradient partials | fp64 reduction partials]
    int bpt = block1_blocks_per_task(a->n, a->h, a->w_, a->co, a->tasks);
    const int sb = sparse_wgrad_blocks_per_task(a->n, a->h, a->w_, a->co, a->tasks);
    if (sb > bpt) bpt = sb;
    b.wpartial = reinterpret_cast<float*>(scratch);
    b.partial = reinterpret_cast<double*>(reinterpret_cast<char*>(scratch) + align_up((size_t)a->tasks * bpt * 9 * a->ci * a->co * sizeof(float), 256));
  }
  b.out = p_out; b.zh_out = zh_out; b.arg_out = arg_out;
  b.arg_in = a->arg_in; b.zh_in = a->zh_in;
  const double inv_m = 1.0 / ((double)a->n * a->h * a->w_);
  int blk = 0;
  const int force = mode & B1_FORCE_GENERAL;      // 0x100: run the general block1_kernel where a lean forward kernel exists
  mode &= ~B1_FORCE_GENERAL;
  switch (mode) {
    case B1_STATS:
      TCHK(launch_block1(S(stream), b, a->tasks, a->ci, B1_STATS, &blk));
      TCHK(launch_bn_finalize(S(stream), b.partial, blk, a->tasks, a->co, inv_m, FIN_STATS, out0, ostride, out1, ostride));
      break;
    case B1_TSTATS:
      TCHK(launch_block1(S(stream), b, a->tasks, a->ci, B1_TSTATS, &blk));
      TCHK(launch_bn_finalize(S(stream), b.partial, blk, a->tasks, a->co, inv_m, FIN_TSTATS, out0, ostride, out1, ostride));
      break;
    case B1_BWD_REDUCE:
    case B1_TBWD_REDUCE:
      TCHK(launch_block1(S(stream), b, a->tasks, a->ci, mode, &blk));
      TCHK(launch_bn_finalize(S(stream), b.partial, blk, a->tasks, a->co, 1.0, FIN_SUMS, out0, ostride, out1, ostride));
      break;
    case B1_FWD:
    case B1_TFWD:
    case B1_TFWD_ARG:
      if (!p_out) return mi_internal_fail(MI_ERR_ARG, "mi_block1_run: forward modes need p_out");
      TCHK(launch_block1(S(stream), b, a->tasks, a->ci, mode | force, nullptr));
      break;
    case B1_BWD_WGRAD:
    case B1_TBWD_WGRAD:
      TCHK(launch_block1(S(stream), b, a->tasks, a->ci, mode, &blk));
      TCHK(launch_wgrad_reduce(S(stream), b.wpartial, blk, 9 * a->ci * a->co, a->tasks, out0, ostride));
      break;
    default:
      return mi_internal_fail(MI_ERR_ARG, "mi_block1_run: unknown mode");
  }
  return MI_OK;
}

// dgamma / dbeta (tangent: R{dgamma} / R{dbeta}) of a fused block 1 from pooled-resolution tensors (pooled_reduce_kernel + finalize)
extern "C" int mi_pooled_reduce(void* stream, const float* p, const float* zh, const float* zhd, const float* dp, const float* dpd,
                                int tasks, int rows, int c, float* out0, float* out1, size_t ostride, void* scratch,
                                size_t scratch_bytes) {
  if (!p || !zh || !dp || !out0 || !out1 || !scratch || ((zhd == nullptr) != (dpd == nullptr)))
    return mi_internal_fail(MI_ERR_ARG, "mi_pooled_reduce: null argument");
  if ((size_t)tasks * pooled_reduce_blocks(rows, c, tasks) * 2 * c * sizeof(double) > scratch_bytes)
    return mi_internal_fail(MI_ERR_WORKSPACE, "mi_pooled_reduce: scratch too small");
  PoolRedArgs pr{p, zh, zhd, dp, dpd, reinterpret_cast<double*>(scratch), rows, c};
  int blk = 0;
  TCHK(launch_pooled_reduce(S(stream), pr, tasks, zhd ? 1 : 0, &blk));
  TCHK(launch_bn_finalize(S(stream), pr.partial, blk, tasks, c, 1.0, FIN_SUMS, out0, ostride, out1, ostride));
  return MI_OK;
}

// Block-1 weight gradient without conv1: sparse_wgrad_kernel (MFMA over the pooling argmax) + gram_wgrad_kernel (dense parts
// from the input Gram matrix g of mi_input_gram).  a->arg_in = the argmax bytes of MI_B1_FWD; tangent != 0: R{dW}.
extern "C" int mi_block1_wgrad_gram(void* stream, const mi_block1_args* a, const double* g, int tangent, float* dw, size_t ostride,
                                    void* scratch, size_t scratch_bytes) {
  if (!a || !a->x || !a->w || !a->arg_in || !a->dp || !g || !dw || !scratch || (tangent && (!a->dpd || !a->wd)))
    return mi_internal_fail(MI_ERR_ARG, "mi_block1_wgrad_gram: null argument");
  if (!sparse_wgrad_supported(a->w_, a->ci, a->co)) return mi_internal_fail(MI_ERR_ARG, "mi_block1_wgrad_gram: input row too wide");
  if (scratch_bytes < mi_block1_scratch_bytes(a->tasks, a->n, a->h, a->w_, a->ci, a->co)) return mi_internal_fail(MI_ERR_WORKSPACE, "mi_block1_wgrad_gram: scratch too small");
  SparseWgArgs sw{};
  sw.x = a->x; sw.arg = a->arg_in; sw.dp = a->dp; sw.dpd = a->dpd; sw.rstd = a->rstd; sw.m2 = a->m2;
  sw.gamma = a->gamma; sw.pstride = a->pstride; sw.gammad = a->gammad; sw.vstride = a->vstride;
  sw.wpartial = reinterpret_cast<float*>(scratch); sw.n = a->n; sw.hh = a->h; sw.ww = a->w_; sw.co = a->co;
  int blk = 0;
  TCHK(launch_sparse_wgrad(S(stream), sw, a->tasks, a->ci, tangent ? 1 : 0, &blk));
  GramWgArgs gw{};
  gw.g = g; gw.spartial = sw.wpartial; gw.nblk = blk; gw.w = a->w; gw.wstride = a->pstride; gw.wd = a->wd; gw.vstride = a->vstride;
  gw.mu = a->mu; gw.rstd = a->rstd; gw.m1 = a->m1; gw.m2 = a->m2;
  gw.gamma = a->gamma; gw.pstride = a->pstride; gw.gammad = a->gammad;
  gw.dgamma = a->dgamma; gw.dbeta = a->dbeta; gw.gstride = a->gstride;
  gw.rdgamma = a->rdgamma; gw.rdbeta = a->rdbeta; gw.hstride = a->hstride;
  gw.out = dw; gw.ostride = ostride; gw.ci = a->ci; gw.co = a->co; gw.inv_m = 1.0 / ((double)a->n * a->h * a->w_);
  TCHK(launch_gram_wgrad(S(stream), gw, a->tasks, tangent ? 1 : 0));
  return MI_OK;
}

// Launch geometry of the conv kernels for a given problem (test aid: lets a test assert it reached tiles_per_wave > 1)
extern "C" int mi_debug_conv_tiles_per_wave(int tasks, int n, int ho, int wo, int co) {
  return conv_tiles_per_wave(n * ho * wo, tasks, co / 32);
}

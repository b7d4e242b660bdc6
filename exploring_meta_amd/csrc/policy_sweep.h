// One-kernel sweeps of the MAML-TRPO Fisher-vector product (reference core_functions/rl.py:417-418: trpo.hessian_vector_product of
// the mean KL inside conjugate gradient, 11 products per meta-iteration).  Round 2 ran one product as ~34 launches of per-layer
// dense kernels (every [rows, 100] intermediate a round trip through HBM, 0.86 ms, 10 % of the fp32 matrix peak); here a product
// is three sweeps over the stored passes plus three small folds:
//     A   hv_t = H_t v          over the support pass at theta          -> u_t = v - lr hv_t
//     B   w_t  = F_t u_t        over the query pass at theta'_t         (Gaussian Fisher: tangent forward, cotangent, backward)
//     C   hv_t = H_t w_t        over the support pass                   -> out = mean_t (w_t - lr hv_t) + damping v
// A sweep takes a slab of 32 rows through the WHOLE chain inside one workgroup: tangent forward of the hidden layer, the head and
// the Gaussian tangent on the vector unit, tangent backward and the weight-gradient products -- both 100 x 100 weight matrices
// (the pass's and the direction's) stay in LDS for the workgroup's lifetime (80 KB), the slab's activations live in LDS / registers
// and nothing but the per-workgroup gradient partial is written.  Matrix work on v_mfma_f32_32x32x2_f32 (exact fp32):
//   * wave w of the 4 owns output columns [32w, 32w+32) of every product; the reduction index is dealt to the lane halves as
//     [0, KH0) / [KH0, H) so a lane's A operands (and the forward products' B operands) are 16-byte LDS reads;
//   * the weight gradient dW2 (100 x 100, the sum over rows) accumulates in registers across all slabs of a task (4 x 16
//     accumulators per lane), one deterministic partial per (workgroup, task), folded in a fixed order.
// The 2-wide layers (states -> hidden, hidden -> actions) and the per-row Gaussian formulas are vector work.
// ReLU policies with equal hidden widths H % 8 == 4 or 0, H <= 128, S <= 4, A <= 6 (DiagNormalPolicy defaults: 2-100-100-2); other
// shapes keep the per-layer path.
#pragma once
#include "mi_common.h"

#define LOG_EPS (-13.815510557964274f)   // log(1e-6), policies.py:14,51
#define HALF_LOG_2PI 0.9189385332046727f
enum { SW_HVP = 0, SW_FISHER = 1, SW_PRIMAL = 2 };

#define SW_MAX_S 4
#define SW_MAX_A 6

struct SweepArgs {
  const float* x;        // states of the pass        [T][B][S]
  const float* act;      // actions                    [T][B][A]   (HVP)
  const float* h1;       // stored activations         [T][B][H]
  const float* h2;
  const float* mu;       // [T][B][A]                               (HVP)
  const float* coef;     // dL/dlogp per row  [T][B]                (HVP)
  const float* dmu;      // primal cotangents [T][B][A], [T][B][H]  (HVP)
  const float* d2;
  const int32_t* count;  // [T] valid rows (null: B)
  const float* theta; size_t tstride;   // parameters of the pass (stride 0 = shared by all tasks)
  const float* dir; size_t dstride;     // direction of the product
  float* partial;        // [T][slots][pitch], pitch = P + 2 (the primal sweep's loss / KL partial sums ride behind the gradient)
  int pitch;
  // primal sweep (SW_PRIMAL): forward + loss + backward of the pass in one go, leaving what the later sweeps read
  const float* adv;      // [T][B]
  const float* old_loc;  // [T][B][A]   surrogate only
  const float* old_scale;   // [T][A]   surrogate only
  int surrogate;         // 0: L = -mean(logp adv) (rl.py:358); 1: L = -mean(exp(logp - logp_old) adv), KL(new || old) (rl.py:459-469)
  int fwd_only;          // loss / KL only (line-search evaluations of the query pass): no backward, gradient partials are zero
  float *h1_out, *h2_out, *mu_out, *dmu_out, *d2_out, *coef_out;     // [T][B][.] stores of the pass (null: not kept)
  int T, B, S, A, spt, spw, slots;      // spt = slabs per task, spw = VIRTUAL slabs per workgroup (a marker in front of every task's slabs: policy_sweep.hip)
  int o_sigma, o_w1, o_b1, o_w2, o_b2, o_w3, o_b3, P;
  unsigned long long* stamps;   // debug: shader-clock stamps of workgroup 0's stages (null in production)
};

// Fold the per-workgroup partials of a sweep in slot order (deterministic) and finish the phase:
//   mode 0 (after A): out[t][p] = v[p] - lr sum                          (u_t)
//   mode 1 (after B): out[t][p] = sum; sigma slots: the Gaussian Fisher's 2 u / D where sigma is not clamped   (w_t)
//   mode 2 (after C): out[t][p] = w[t][p] - lr sum    (then the mean over tasks + damping v: by the last workgroup of a parameter block
//                     when FoldArgs::counter is set, by the caller's mean_tasks launch otherwise)
//   mode 3 (primal)  : out[t][p] = sum (the pass's gradient); loss_t[t], kl_t[t] = the two extra slots
struct FoldArgs {
  const float* partial; int slots, spt, spw, T, P, pitch;
  float *loss_t, *kl_t;    // mode 3: per-task loss / KL sums of a primal sweep
  const float* v;          // [P]
  const float* w;          // [T][P]   (mode 2)
  const float* thetap;     // [T][P]   (mode 1: rho of theta')
  const float* u;          // [T][P]   (mode 1: direction)
  float lr, damping;
  int o_sigma, A;
  float* out;
  int mode;
  // mode 2 with a counter: the workgroup of a parameter block that finishes LAST among the T tasks also forms
  //   mean_out[p] = inv_T sum_t out[t][p] + damping v[p]   (tasks in order: the value does not depend on who is last)
  // -- the mean over tasks without its own launch.  counter: one unsigned per parameter block, zero on entry, zero again on exit.
  unsigned* counter; float* mean_out; float inv_T;
};
// launchers (policy_sweep.hip, built WITHOUT -amdgpu-mfma-vgpr-form: the sweep keeps its 80 accumulator registers in the AGPR half
// of the register file, where only MFMAs reach them, and all 256 architectural VGPRs for operands, prefetch and vector work)
bool policy_sweep_supported(int act_relu, int h1, int h2, int s, int a);
hipError_t launch_policy_sweep(hipStream_t st, const SweepArgs& a, int grid, int mode);      // mode = SW_*
hipError_t launch_policy_sweep_fold(hipStream_t st, const FoldArgs& f, int tasks);

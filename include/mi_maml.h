/* mi_maml.h -- C ABI of the MI355X-native MAML/ANIL inner/outer-loop engine (libmi_maml.so).
 *
 * Drop-in boundary for the reference's hot path (Kostis-S-Z/exploring_meta).  The reference is pure Python with no FFI of
 * its own; each entry point below names the reference interface it replaces (file:line under /root/reference).  A
 * maintainer binds these with ctypes; exploring_meta_amd/_lib.py is that binding.
 *
 * Conventions: extern "C", plain pointers and sizes, no torch types.  Every function returns 0 on success and a negative
 * code on error; mi_last_error() returns a message for the last failing call on that engine (or the global message when
 * engine is NULL).  All tensor pointers are caller-owned DEVICE memory (fp32 unless noted), the engine keeps nothing
 * across calls except its model description; work is enqueued on `stream` (a hipStream_t passed as void*), stream
 * ordered, with no internal synchronisation.  One engine per device; a handle is not re-entrant.
 *
 * Parameter vectors ("theta", gradients) are FLAT fp32 vectors in the reference's `module.parameters()` order and
 * layouts: per ConvBlock  normalize.weight[C], normalize.bias[C], conv.weight[Co,Ci,3,3], conv.bias[Co]
 * (vision_models.py:168-186), then linear.weight[ways,F], linear.bias[ways] (vision_models.py:47,103).
 */
#ifndef MI_MAML_H
#define MI_MAML_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK 0
#define MI_ERR_ARG (-1)      /* invalid argument / unsupported shape */
#define MI_ERR_HIP (-2)      /* HIP runtime error */
#define MI_ERR_WORKSPACE (-3) /* workspace too small */

typedef struct mi_engine mi_engine;

/* Model family of core_functions/vision_models.py: ConvBase (:121-146) + optional classifier head.
 *   MiniImagenetCNN(ways)   (:66-118): {4, 3,84,84, 32, max_pool=1, ways, head_mean_pool=0}
 *   OmniglotCNN(ways)       (:10-63) : {4, 1,28,28, 64, max_pool=0, ways, head_mean_pool=1}
 *   ANIL trunk ConvBase(...)(anil_vision.py:86-91): ways=0 is not a classifier; see mi_anil_* below. */
typedef struct {
  int32_t n_layers;       /* ConvBlocks */
  int32_t in_channels, in_h, in_w;
  int32_t hidden;         /* filters per block (32 or 64) */
  int32_t max_pool;       /* 1: stride-1 conv + MaxPool2d(2,2,floor); 0: stride-2 conv, no pooling (vision_models.py:157-165) */
  int32_t ways;           /* classifier outputs */
  int32_t head_mean_pool; /* 1: x.mean(dim=[2,3]) then Linear(hidden,ways) (:53-54); 0: view(-1, hw*hidden) then Linear (:109) */
} mi_model_desc;

int mi_engine_create(const mi_model_desc* desc, int device, mi_engine** out);
void mi_engine_destroy(mi_engine* e);
const char* mi_last_error(const mi_engine* e);
const char* mi_version(void);

/* Ablation/test switch: 0 = block 1 through the generic (z-storing) kernels, 1 = fused conv-recompute kernels (default when
 * the geometry allows: Ci in {1,3}, stride-1 conv + pooling, even H/W) with the BatchNorm statistics of repeated support
 * passes taken from the input Gram matrix (mi_input_gram) and the BatchNorm-backward reductions from zhat kept at the pooling
 * argmax, 2 = fused kernels only: every statistic / reduction by a conv-recompute pass.  Since round 6 mode 1 also takes the QUERY pass of a
 * call with a backward half through a Gram matrix of the query images (formed on the engine's side stream beside the inner loop); 3 = as 1 with
 * the Gram matrix for the support passes only (the behaviour of rounds 2 - 5: the query pass by conv-recompute kernels). */
int mi_engine_set_fused_block1(mi_engine* e, int on);

/* 1 (default): the weight gradients of blocks >= 2 run on an engine-owned side stream, forked from the caller's stream once
 * dz of the block is written and joined before the pass's gradients are used (they are matrix-bound, the BatchNorm kernels
 * that follow on the main stream are HBM-bound).  0: every kernel on the caller's stream.  1 (default): one fork of the side stream per
 * hidden block.  3: ONE fork per backward pass (the weight gradients issued together once the last hidden block's dz exists): saves two
 * event records per pass -- each idles the caller's stream for ~7 us -- but loses more overlap than that (measured: +2.9 % on cfg2).
 * Results are identical in every mode (per-launch times from mi_profile_* are only additive with 0). */
int mi_engine_set_overlap(mi_engine* e, int on);

/* 1: a call of mi_meta_batch_maml / mi_meta_batch_anil whose arguments (every pointer, size and scalar, the stream included) equal
 * those of an earlier call on this engine is replayed as ONE hipGraphLaunch of the captured launch sequence (first sight of a
 * signature runs eagerly, the second is captured and instantiated; a small cache holds up to 8 signatures).  For callers that keep
 * their parameter / data / output / workspace buffers in place between meta-iterations (bench.py, the drivers): it removes the host
 * launch cost of the ~75-280 kernel launches of a call, which dominates the few-image configurations.  Default 0.  Results are
 * those of the eager call.  mi_profile_enable and mi_debug_set_trace suspend it for their calls. */
int mi_engine_set_graph(mi_engine* e, int on);

/* While buf != NULL every forward pass of mi_meta_batch_maml (inner steps 0..K-1, then the query pass) and the trunk pass of
 * mi_meta_batch_anil also writes the BatchNorm batch statistics of every block: buf [passes][tasks][2][C_total] floats (C_total = sum
 * of the blocks' filters, block-major; [0] batch mean, [1] biased batch variance).  torch.nn.BatchNorm2d updates running_mean /
 * running_var from these on every learner(x) of the reference (vision_models.py:168-174; the buffers are shared by learn2learn's clones
 * and saved by utils/experiment.py:85-90); core_functions/vision_models.py (running_stats_contribution / apply_running_stats) folds them in the reference's call order. */
/* Only the fused calls export (their passes are numbered from 0 per call); mi_forward_logits / mi_learner_* run with the export paused. */
int mi_engine_set_bn_export(mi_engine* e, float* buf, size_t floats);

/* 1 (default): the per-workgroup fp64 partials of every BatchNorm statistic / reduction are folded, in a fixed order, by the
 * last workgroup of the producing kernel (arrival counter per task); 0: by separate bn_finalize launches.  Bit-identical
 * results either way; the switch exists for ablation and tests. */
int mi_engine_set_fused_finalize(mi_engine* e, int on);

/* 1 (default): the BatchNorm-backward sums of a fused block 1 (dgamma, dbeta and, in the Hessian-vector product, their tangents) are
 * formed in the epilogue of block 2's dgrad kernel, whose output tile IS the cotangent of block 1's pooled output; 0: by a
 * separate streaming pass over the pooled-resolution tensors.  Same fp64 sums in another order (results agree to fp32 rounding). */
int mi_engine_set_fused_block1_reduce(mi_engine* e, int on);

/* Debug/test aid: byte offsets of {theta, g, xs, sup[0].p[0], sup[0].dp[0], sup[0].mu[0], sup[0].rstd[0], sup[0].p[1],
 * qry.p[0], total} inside the workspace of a mi_meta_batch_maml call with these sizes (out: 10 entries). */
int mi_debug_plan_offsets(const mi_engine* e, int tasks, int ways, int shots, int adapt_steps, int second_order, size_t* out);

/* Number of fp32 parameters (= sum(p.numel() for p in model.parameters())). */
int mi_param_count(const mi_engine* e, size_t* n);

/* Bytes of caller-provided scratch for one mi_meta_batch_* call with these sizes. */
int mi_workspace_bytes(const mi_engine* e, int tasks, int ways, int shots, int adapt_steps, int second_order, size_t* bytes);

/* One meta-batch of MAML tasks: replaces, for `tasks` tasks at once, the body of the reference's per-task loop
 *   learner = maml.clone(); fast_adapt(batch, learner, loss, adapt_steps, shots, ways, device); eval_loss.backward()
 * (vision/maml_vision.py:102-114; core_functions/vision.py:6-18; utils/data_pre.py:115-129; learn2learn MAML.clone/adapt)
 * with loss = CrossEntropyLoss(reduction='mean') (maml_vision.py:86).
 *   theta       [P]                      meta-parameters (read only)
 *   data        [tasks, 2*shots*ways, C, H, W]  NCHW fp32, exactly the reference's task batches stacked
 *   labels      [tasks, 2*shots*ways]    int64, as sampled by the reference (sorted by class)
 *   inner_lr    MAML(model, lr=...)      (maml_vision.py:84)
 *   second_order 1 = create_graph inner updates (the reference's hard-coded first_order=False), 0 = first-order MAML
 *   with_grad   0 = evaluation only (core_functions/vision.py:26-42 evaluate, and the validation half :117-124)
 * Outputs:
 *   loss_out    [tasks]  query loss per task          (valid_loss, vision.py:16)
 *   acc_out     [tasks]  query accuracy per task      (valid_accuracy, vision.py:17,21-23)
 *   meta_grad_out [P]    SUM over tasks of d valid_loss / d theta  (what .backward() accumulates, maml_vision.py:112);
 *                        the caller applies 1/meta_batch_size (:139-140).  May be NULL when with_grad == 0.
 *   logits_out  [tasks, shots*ways, ways] query predictions, or NULL. */
int mi_meta_batch_maml(mi_engine* e, void* stream, const float* theta, const float* data, const int64_t* labels,
                       int tasks, int ways, int shots, int adapt_steps, float inner_lr, int second_order, int with_grad,
                       float* loss_out, float* acc_out, float* meta_grad_out, float* logits_out,
                       void* workspace, size_t workspace_bytes);

/* The train AND the validation half of one meta-iteration in the same launches.  The reference runs, per train task, a second
 *   learner = maml.clone(); fast_adapt(valid_batch, ...)        without backward
 * (vision/maml_vision.py:117-124): the K support steps and the query forward of those validation tasks are the same kernels as the
 * train tasks'.  `tasks` task batches are stacked as for mi_meta_batch_maml; the FIRST grad_tasks of them are train tasks -- query
 * backward, second-order adjoint recursion, summed into meta_grad_out -- the remaining tasks - grad_tasks are adapted and scored only.
 * loss_out / acc_out [tasks] cover both halves.  grad_tasks == tasks is mi_meta_batch_maml(with_grad = 1), grad_tasks == 0 is
 * with_grad = 0; the workspace is mi_workspace_bytes(tasks, ..., second_order). */
int mi_meta_batch_maml_tv(mi_engine* e, void* stream, const float* theta, const float* data, const int64_t* labels,
                          int tasks, int grad_tasks, int ways, int shots, int adapt_steps, float inner_lr, int second_order,
                          float* loss_out, float* acc_out, float* meta_grad_out, float* logits_out,
                          void* workspace, size_t workspace_bytes);

/* Ablation / test switch: 1 (default) = every pass of mi_meta_batch_maml ends in ONE "advance" launch (weight-gradient partial folds,
 * block 1's Gram-matrix assembly, p <- p - lr g of learn2learn's maml_update / the adjoint recursion, the next pass's Gram statistics);
 * 0 = the separate launches (reduce_partials x3, gram_wgrad, axpy, gram_stats, a memset).  Bit-identical results. */
int mi_engine_set_fused_tail(mi_engine* e, int on);

/* Ablation / test switch: 1 (default) = the LAST ConvBlock's BatchNorm + ReLU + MaxPool (core_functions/vision_models.py:188-193), the linear
 * head with its cross-entropy and accuracy (vision_models.py:109, core_functions/vision.py:11,16-18,21-23), the head's backward and that block's
 * BatchNorm-backward sums -- in the Hessian-vector passes their tangents -- run as ONE launch per pass, four workgroups (row groups) per task whose
 * sums meet in the last-arriving workgroup (csrc/tail.hip); 0 = the four separate launches per pass (bn_fwd, head rows, head grads,
 * bn_bwd_reduce).  Same arithmetic in the same order: the pooled output, logits, loss, accuracy, head gradients and feature cotangents are
 * bit-identical; the BatchNorm-backward sums are the same fp64 terms folded in a fixed order that no longer depends on the tasks per call.  Applies to nets whose last block is a generic (hidden -> hidden) block
 * feeding a flattened head (MiniImagenetCNN; not the mean-pooled OmniglotCNN head) outside the opt-in fp16 operand form. */
int mi_engine_set_fused_last_block(mi_engine* e, int on);
/* Debug aid (tools/tail_stamps.py): the 100 MHz wall clock at the stage boundaries of the one-launch tail, thread 0 of each of its four workgroups
 * per task: buf [tasks][4][16] 64-bit words of device memory, overwritten by every tail launch; NULL = off (the default). */
int mi_debug_tail_stamps(mi_engine* e, unsigned long long* buf);

/* One meta-batch of ANIL tasks (vision/anil_vision.py:116-122 with features = Sequential(ConvBase, view(-1, fc_neurons)),
 * head = MAML(Linear(fc_neurons, ways)), :86-94): the trunk runs once per task on all 2*shots*ways images (BatchNorm over
 * support and query together, utils/data_pre.py:118-119), only the head is adapted, and the outer gradient reaches both.
 * The engine is created with the trunk's ConvBase geometry and `ways`; theta / meta_grad_out are
 * [features.parameters() ..., head.weight[ways, fc_neurons], head.bias[ways]] -- the order of the reference's optimizer list
 * (anil_vision.py:97).  Other arguments as mi_meta_batch_maml. */
int mi_anil_workspace_bytes(const mi_engine* e, int tasks, int ways, int shots, int adapt_steps, size_t* bytes);
int mi_meta_batch_anil(mi_engine* e, void* stream, const float* theta, const float* data, const int64_t* labels,
                       int tasks, int ways, int shots, int adapt_steps, float inner_lr, int second_order, int with_grad,
                       float* loss_out, float* acc_out, float* meta_grad_out, float* logits_out,
                       void* workspace, size_t workspace_bytes);

/* Plain classifier forward, no adaptation: `learner(x)` / `model(x)` (vision_models.py:51-55,107-110), BatchNorm in train
 * mode over the n images of each of the `tasks` batches.  x [tasks, n, C, H, W] NCHW; logits_out [tasks, n, ways]. */
int mi_forward_workspace_bytes(const mi_engine* e, int tasks, int n, size_t* bytes);
int mi_forward_logits(mi_engine* e, void* stream, const float* theta, const float* x, int tasks, int n, float* logits_out,
                      void* workspace, size_t workspace_bytes);

/* Step-wise learner: the reference's `learner = maml.clone(); learner(x); learner.adapt(loss); learner.get_rep_i(x, i)`
 * (misc_scripts/cl_vision.py:56-66, misc_scripts/rc_vision.py:66-86, core_functions/maml.py:15-19,
 * core_functions/vision_models.py:57-63,112-118) with the fast weights held by the caller.
 *   theta [theta_tasks, P]  reference parameter order; theta_tasks = 1 (all task batches share theta) or = tasks
 *   x     [tasks, n, C, H, W] NCHW; BatchNorm in train mode over the n images of each task batch
 * mi_learner_forward:  logits_out [tasks, n, ways] (or NULL); rep_out (or NULL) = output of the first `rep_layer` ConvBlocks,
 *   rep_layer in 1..layers, NCHW [tasks, n, hidden, h', w'] (`Sequential(*base.children()[:layer])(x)`; layers = base(x)).
 * mi_learner_backward: grad_out [theta_tasks, P] = d sum(logits * dlogits) / d theta (summed over task batches when theta
 *   is shared): the vector-Jacobian product autograd asks of `learner(x)`; conv biases get exact zeros (batch-stat BN).
 *   Its own derivative is mi_learner_hvp; the fused second-order path is mi_meta_batch_maml.
 * Workspace: mi_forward_workspace_bytes(e, tasks, n). */
int mi_learner_forward(mi_engine* e, void* stream, const float* theta, int theta_tasks, const float* x, int tasks, int n,
                       float* logits_out, int rep_layer, float* rep_out, void* workspace, size_t workspace_bytes);
int mi_learner_backward(mi_engine* e, void* stream, const float* theta, int theta_tasks, const float* x, const float* dlogits,
                        int tasks, int n, float* grad_out, void* workspace, size_t workspace_bytes);

/* Double backward of the step-wise learner (learn2learn `MAML.adapt` of a second-order learner differentiates through
 * `grad(loss, params, create_graph=True)`, learn2learn maml.py adapt / maml_update; driven step-wise at misc_scripts/rc_vision.py:67-70).
 * With s(theta) = sum(logits(theta) * dlogits) and g = ds/dtheta (mi_learner_backward), for a cotangent v [theta_tasks, P] on g:
 *   grad_theta_out [theta_tasks, P] = (d^2 s / dtheta^2) v with dlogits held fixed (forward-over-reverse sweep),
 *   logits_dot_out [tasks, n, ways] = J(theta) v  -- the vector-Jacobian product with respect to dlogits.
 * Workspace: mi_learner_hvp_workspace_bytes(e, tasks, n). */
int mi_learner_hvp_workspace_bytes(const mi_engine* e, int tasks, int n, size_t* bytes);
int mi_learner_hvp(mi_engine* e, void* stream, const float* theta, int theta_tasks, const float* x, const float* dlogits,
                   const float* v, int tasks, int n, float* grad_theta_out, float* logits_dot_out, void* workspace,
                   size_t workspace_bytes);

/* One recurrence of cherry.algorithms.trpo.conjugate_gradient (reference rl.py:418, the loop body) on device vectors, fp64, one launch:
 *   alpha = rr[0] / (p . ap + eps);  x += alpha p;  r -= alpha ap;  rr_new = r . r;  p = r + (rr_new / rr[0]) p;  rr[0] = rr_new, rr[1] = alpha.
 * x, r, p: fp64 [n];  ap: fp32 [n] (the Fisher-vector product of p32);  p32: fp32 [n], the new p for the next product. */
int mi_cg_update(void* stream, double* x, double* r, double* p, const float* ap, double* rr, float* p32, size_t n, double eps);
/* The same with the reference's convergence test `if r_dot_new < tol: break` (cherry conjugate_gradient, rl.py:418) taken on the device:
 * rr is [3] (rr[2] = 0 on entry of a solve); once rr_new < tol, rr[2] latches to 1 and every later call of the solve leaves x, r, p, p32
 * untouched -- the host loop runs its fixed number of iterations without synchronising, x is exactly the x at the break. */
int mi_cg_update_checked(void* stream, double* x, double* r, double* p, const float* ap, double* rr, float* p32, size_t n, double eps,
                         double tol);

/* The start of that solve as one launch: r = p = b (fp64), x = 0, p32 = b, rr = (b . b, 0, 0)  (rl.py:418; cherry's conjugate_gradient
 * begins with x = 0, r = p = b). */
int mi_cg_init(void* stream, const float* b, double* x, double* r, double* p, float* p32, double* rr, size_t n);

/* The trust-region step from the solve's direction and its Fisher-vector product (reference rl.py:419-421), one launch:
 *   shs = 0.5 step . fstep (summed in fp64);  lagrange = sqrt(shs / max_kl);  out = step / lagrange;  *lagrange_out = lagrange (or NULL).
 * A negative shs yields NaN, as in the reference. */
int mi_trpo_scale_step(void* stream, const float* step, const float* fstep, size_t n, float max_kl, float* out, float* lagrange_out);

/* Generalised advantage estimation with cherry's LinearValue baseline for a list of replays, one launch (reference
 * core_functions/rl.py:95-110 compute_advantages: ch.td.discount, LinearValue.fit / __call__ (features [s, s^2, t, t^2, t^3, 1],
 * t = row/100, ridge normal equations with `reg`; rl/maml_trpo.py:85 passes env.action_size as reg), bootstraps,
 * cherry.pg.generalized_advantage; normalize != 0 applies ch.normalize (rl.py:355) -- (adv - mean) / (unbiased std + 1e-8)).
 *   states, next_states [replays, rows, state_dim]; rewards, dones [replays, rows]; count [replays] rows in use (NULL = rows);
 *   adv_out [replays, rows] (0 past a replay's count); weight_out [replays, 2*state_dim+4] fitted baseline weights (fp64) or NULL;
 *   weight_in [replays, 2*state_dim+4] != NULL: use these baseline weights instead of fitting (compute_advantages' update_vf=False,
 *   rl.py:401 -- the query replay's loss is computed with the baseline fitted to the last support replay).
 * fp64 arithmetic throughout; state_dim <= 8 and rows <= mi_gae_max_rows(state_dim) (the replay is staged in LDS). */
int mi_gae_max_rows(int state_dim);
int mi_gae_advantages(void* stream, const float* states, const float* next_states, const float* rewards, const float* dones,
                      const int32_t* count, const double* weight_in, int replays, int rows, int state_dim, double gamma, double tau,
                      double reg, int normalize, float* adv_out, double* weight_out);

/* Assembly of a padded device batch from a list of contiguous fp32 device arrays -- the fields of the replays a meta-iteration collected
 * (reference core_functions/rl.py:444-465 walks them task by task, replay by replay) and the parameters of the stored adapted policies
 * (rl.py:447-449): segment k copies nfloat[k] floats from src[k] to dst[k] and writes zeros up to npad[k] (>= nfloat[k]) floats.
 * src, dst, nfloat, npad are HOST arrays of nseg entries holding DEVICE pointers; the pointers travel in kernel arguments, one launch per
 * 128 segments. */
int mi_copy_segments(void* stream, const void* const* src, void* const* dst, const uint32_t* nfloat, const uint32_t* npad, int nseg);
/* n int32 values from HOST memory to the device array dst, carried in kernel arguments (512 per launch): the per-replay row counts of
 * the batch above without a staging copy. */
int mi_upload_i32(void* stream, int32_t* dst, const int32_t* host_values, int n);

/* Adam step on the flat meta-parameters with torch.optim.Adam defaults (maml_vision.py:85,139-141):
 * grad is first scaled by grad_scale (= 1/meta_batch_size). step is the 1-based step count after increment. */
int mi_adam_step(void* stream, float* theta, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, int step,
                 float lr, float beta1, float beta2, float eps, float grad_scale);

/* ------------------------------------------------------------------------------------------------------------------
 * Per-kernel entry points (unit parity tests against oracle/kernels_ref.py).  Engine-internal layouts:
 * activations NHWC [tasks, N, H, W, C]; conv weights tap-major [9, Ci, Co]; all per-task parameter pointers advance by
 * `pstride` floats per task. */

/* utils/data_pre.py:115-129 prepare_batch: even rows -> support, odd rows -> query, NCHW -> NHWC, labels -> int32. */
int mi_prepare_batch(void* stream, const float* data, const int64_t* labels, int tasks, int n2, int c, int h, int w,
                     float* xs, float* xq, int32_t* ys, int32_t* yq);

/* Block-1 statistics without running conv1 (gram.hip): conv1's output is linear in its weights, so BatchNorm's per-channel
 * sums are quadratic forms of G = P^T P, P = the zero-padded 3x3xCi patches of the images (+ a constant-one column).
 *   mi_input_gram: x [tasks,n,H,W,ci] NHWC (ci = 1 or 3) -> g_out [tasks, NG, NG] fp64, NG = 32 (ci=3) / 16 (ci=1);
 *     entry (a, b) for a,b < 9ci = sum over pixels of patch_a * patch_b (a = tap*ci + c), row/column 9ci = the patch sums.
 *   mi_gram_bn_stats: w9d == NULL -> out0 = mean, out1 = 1/sqrt(biased var + eps) of conv3x3(x, w9) over `pixels` = n*H*W
 *     positions (what mi_conv3x3_bn_stats finalises); w9d != NULL -> out0 = mean(zd), out1 = mean(zhat * zd) for the tangent
 *     zd = conv3x3(x, w9d), zhat = (z - mu) * rstd (the statistics of the tangent BatchNorm). */
size_t mi_input_gram_scratch_bytes(int tasks, int n, int h, int ci);
int mi_input_gram(void* stream, const float* x, int tasks, int n, int h, int w, int ci, void* scratch, size_t scratch_bytes,
                  double* g_out);
int mi_gram_bn_stats(void* stream, const double* g, int tasks, int ci, int co, const float* w9, size_t pstride, const float* w9d,
                     size_t vstride, int pixels, float* out0, float* out1, const float* mu, const float* rstd);

/* Device-to-device streaming copy (bytes % 16 == 0): the kernel bench.py uses to measure the achievable HBM bandwidth in the
 * same run as the engine kernels ("measured HBM roofline"). */
int mi_stream_copy(void* stream, const void* src, void* dst, size_t bytes);

/* Task sampling from a dataset resident in HBM: `tasks.sample()` of learn2learn's TaskDataset with LoadData
 * (utils/data_pre.py:16-112; call sites vision/maml_vision.py:103,116) for a whole meta-batch, no host pixel traffic.
 *   dataset [num_images, C, H, W]  fp32, or uint8 when dataset_is_u8 (raw 0..255 pixels, converted exactly)
 *   index   [tasks, n2] int64      image ids drawn on the host (NWays / KShots(2*shots) / ConsecutiveLabels order), each in
 *                                  [0, num_images) -- the caller guarantees the range, the kernel does not check it
 *   rot     [tasks, n2] uint8      quarter turns counter-clockwise per row (RandomClassRotation, data_pre.py:34), or NULL
 *   data_out [tasks, n2, C, H, W]  exactly the stacked task batches mi_meta_batch_maml / _anil take */
int mi_sample_tasks(void* stream, const void* dataset, int dataset_is_u8, size_t num_images, int c, int h, int w,
                    const int64_t* index, const uint8_t* rot, int tasks, int n2, float* data_out);

/* conv3x3 pad 1 (ConvBlock.conv, vision_models.py:177-185, bias dropped: batch-stat BN cancels it) + per-channel
 * sum / sum-of-squares partials; then finalize -> mu, rstd (BatchNorm2d train mode, eps 1e-5, biased variance). */
int mi_conv3x3_bn_stats(void* stream, const float* x, const float* w9, size_t pstride, int tasks, int n, int h, int wd,
                        int ci, int co, int stride, float* z, float* mu, float* rstd, void* scratch, size_t scratch_bytes);
/* BN-apply + ReLU + MaxPool2d(2,2) (or identity). */
int mi_bn_relu_pool(void* stream, const float* z, const float* mu, const float* rstd, const float* gamma, const float* beta,
                    size_t pstride, int tasks, int n, int ho, int wo, int c, int pool, float* p);
/* backward of BN+ReLU+pool: dgamma, dbeta, dz. */
int mi_bn_relu_pool_bwd(void* stream, const float* z, const float* mu, const float* rstd, const float* gamma, const float* beta,
                        size_t pstride, const float* dp, int tasks, int n, int ho, int wo, int c, int pool,
                        float* dgamma, float* dbeta, size_t gstride, float* dz, void* scratch, size_t scratch_bytes);
/* conv backward: dx = dgrad(dz, w9) (may be NULL), dw9 = wgrad(x, dz). */
int mi_conv3x3_bwd(void* stream, const float* x, const float* dz, const float* w9, size_t pstride, int tasks, int n, int h,
                   int wd, int ci, int co, int stride, float* dx, float* dw9, size_t gstride, void* scratch, size_t scratch_bytes);
/* Linear + CrossEntropy(mean) forward/backward on features f [tasks, n, F]. */
int mi_head_fwd_bwd(void* stream, const float* f, const float* wl, const float* bl, size_t pstride, const int32_t* y,
                    int tasks, int n, int feat, int ways, float* loss, float* acc, float* logits, float* prob, float* dl,
                    float* dwl, float* dbl, size_t gstride, float* df);
size_t mi_kernel_scratch_bytes(int tasks, int n, int h, int w, int c);

/* ------------------------------------------------------------------------------------------------------------------
 * Tangent (R-operator) and fused block-1 kernels: unit-test entry points (tests/test_gpu_tangent_kernels.py).  They are the
 * kernels behind the second-order path -- the double-backward of ConvBlock.forward (vision_models.py:188-193) that the
 * reference gets from autograd with create_graph=True (learn2learn MAML.adapt, call site core_functions/vision.py:13). */

/* zd = conv3x3(x0, w0) [+ conv3x3(x1, w1) when x1 != NULL]  with the tangent-BatchNorm statistics
 *   m1 = mean(zd), m2 = mean(zhat * zd), zhat = (z - mu) * rstd          (all [tasks, co]). */
int mi_conv3x3_tangent(void* stream, const float* x0, const float* w0, const float* x1, const float* w1, size_t pstride,
                       const float* z, const float* mu, const float* rstd, int tasks, int n, int h, int wd, int ci, int co,
                       int stride, float* zd, float* m1, float* m2, void* scratch, size_t scratch_bytes);
/* Two-term conv backward of the tangent pass:  dw9 = wgrad(x0, dz0) + wgrad(x1, dz1);  dx (may be NULL) = dgrad(dz0, w0) +
 * dgrad(dz1, w1). */
int mi_conv3x3_bwd2(void* stream, const float* x0, const float* dz0, const float* x1, const float* dz1, const float* w0,
                    const float* w1, size_t pstride, int tasks, int n, int h, int wd, int ci, int co, int stride, float* dx,
                    float* dw9, size_t gstride, void* scratch, size_t scratch_bytes);

/* Tangent of BN + ReLU + MaxPool (formulas: oracle/kernels_ref.py header).  Per-task vectors advance by their stride. */
typedef struct {
  const float *z, *zd;                  /* conv output and its tangent [tasks, n, ho, wo, c] */
  const float *mu, *rstd, *m1, *m2;     /* [tasks, c] */
  const float *gamma, *beta; size_t pstride;      /* parameters */
  const float *gammad, *betad; size_t vstride;    /* tangent direction */
  const float *dgamma, *dbeta; size_t gstride;    /* primal BatchNorm gradients (backward only) */
  const float *dp, *dpd;                /* cotangent of the block output and its tangent [tasks, n, hp, wp, c] (backward only) */
  int32_t tasks, n, ho, wo, c, pool;
} mi_bn_tangent_args;
int mi_bn_tangent_fwd(void* stream, const mi_bn_tangent_args* a, float* pd);
int mi_bn_tangent_bwd(void* stream, const mi_bn_tangent_args* a, float* rdgamma, float* rdbeta, size_t hstride, float* rdz,
                      void* scratch, size_t scratch_bytes);

/* Fused first ConvBlock (conv recomputed inside every kernel; Ci in {1,3}, stride 1, pooling, even H/W >= 16). */
#define MI_B1_STATS 0        /* -> out0 = mean, out1 = rstd of conv1's output */
#define MI_B1_FWD 1          /* -> p_out (+ zh_out = zhat at each window's argmax, arg_out = argmax position, 4 = ReLU off) */
#define MI_B1_BWD_REDUCE 2   /* -> out0 = dgamma, out1 = dbeta */
#define MI_B1_BWD_WGRAD 3    /* -> out0 = dW [tasks][ostride] */
#define MI_B1_TSTATS 4       /* -> out0 = m1, out1 = m2 */
#define MI_B1_TFWD 5         /* -> p_out = tangent of the block output (+ zh_out = tangent of zhat at the argmax) */
#define MI_B1_TBWD_REDUCE 6  /* -> out0 = R{dgamma}, out1 = R{dbeta} */
#define MI_B1_TBWD_WGRAD 7   /* -> out0 = R{dW} */
#define MI_B1_TFWD_ARG 8     /* as TFWD, from the stored argmax / zhat (arg_in, zh_in) with one conv instead of two */
typedef struct {
  const float* x;                       /* [tasks, n, h, w, ci] NHWC */
  const float *w, *wd;                  /* conv weights [9*ci, co] of theta / of the tangent direction */
  const float *gamma, *beta; size_t pstride;      /* stride of w / gamma / beta */
  const float *gammad, *betad; size_t vstride;    /* stride of wd / gammad / betad */
  const float *mu, *rstd, *m1, *m2;     /* [tasks, co] */
  const float *dgamma, *dbeta; size_t gstride;
  const float *rdgamma, *rdbeta; size_t hstride;
  const float *dp, *dpd;                /* [tasks, n, h/2, w/2, co] */
  const uint8_t* arg_in; const float* zh_in;      /* outputs of MI_B1_FWD (TFWD_ARG, mi_block1_wgrad_gram) */
  int32_t tasks, n, h, w_, ci, co;
} mi_block1_args;
size_t mi_block1_scratch_bytes(int tasks, int n, int h, int w, int ci, int co);
/* mode | 0x100 (forward modes MI_B1_FWD / MI_B1_TFWD_ARG): run the general block1_kernel instead of the lean block1_fwd_kernel the
 * engine uses for tasks of fewer than ~198 84x84x3 images (tests keep the fallback honest). */
int mi_block1_run(void* stream, int mode, const mi_block1_args* a, float* p_out, float* zh_out, uint8_t* arg_out, float* out0,
                  float* out1, size_t ostride, void* scratch, size_t scratch_bytes);
/* dgamma / dbeta of a fused block 1 (zhd, dpd != NULL: their tangents) from pooled-resolution tensors [tasks, rows, c]. */
int mi_pooled_reduce(void* stream, const float* p, const float* zh, const float* zhd, const float* dp, const float* dpd,
                     int tasks, int rows, int c, float* out0, float* out1, size_t ostride, void* scratch, size_t scratch_bytes);
/* Block-1 weight gradient (tangent != 0: its tangent) without conv1: sparse MFMA pass over the pooling argmax + dense parts
 * from the input Gram matrix g of mi_input_gram. */
int mi_block1_wgrad_gram(void* stream, const mi_block1_args* a, const double* g, int tangent, float* dw, size_t ostride,
                         void* scratch, size_t scratch_bytes);

/* Debug/test aids.  mi_debug_conv_tiles_per_wave: the tiles-per-wave split the conv kernels use for this problem (lets a test
 * assert it exercises the multi-tile loop).  mi_debug_set_trace: while buf != NULL every mi_meta_batch_maml call with
 * with_grad != 0 also writes, per task and in the reference's parameter order,
 *   theta_k [K+1][tasks][P] | g_k = grad L_support(theta_k) [K][tasks][P] | lam_{k+1} (the vector fed to the k-th
 *   Hessian-vector product) [K][tasks][P] | H_support(theta_k) lam_{k+1} [K][tasks][P]
 * (the last two only for second-order calls): the per-step state the teacher-forced parity tests check against the oracle. */
int mi_debug_conv_tiles_per_wave(int tasks, int n, int ho, int wo, int co);
int mi_debug_set_trace(mi_engine* e, float* buf, size_t floats);

/* ------------------------------------------------------------------------------------------------------------------
 * Per-launch profiling with HIP events on the caller's stream (the reference has no profiler hooks, SURVEY.md section 5;
 * bench.py uses this for its roofline figures).  kind = op * 8 + layer; kind_filter < 0 profiles every launch. */
int mi_profile_enable(mi_engine* e, int on, int kind_filter);
int mi_profile_kinds(void);
const char* mi_profile_op_name(int op);
int mi_profile_collect(mi_engine* e, double* total_ms, int64_t* count, int n_kinds);

/* ------------------------------------------------------------------------------------------------------------------
 * MAML-TRPO policy path (BASELINE config 5; core_functions/policies.py:30-67, core_functions/rl.py:346-473).
 * Policy parameters are a flat fp32 vector in DiagNormalPolicy's named_parameters() order:
 *   sigma[A], mean.0.weight[H1,S], mean.0.bias[H1], mean.2.weight[H2,H1], mean.2.bias[H2], mean.4.weight[A,H2], mean.4.bias[A].
 * Replays are padded to `batch` rows per task: states [tasks,batch,S], actions [tasks,batch,A], advantages [tasks,batch]
 * (already normalised, rl.py:354-355), count[tasks] = valid rows.  The baseline fit / GAE (rl.py:95-110) stay on the host. */
typedef struct mi_policy mi_policy;
typedef struct {
  int32_t state_size, action_size, hidden1, hidden2;
  int32_t activation; /* 0 = ReLU (DiagNormalPolicy default), 1 = tanh (activation='tanh', policies.py:32-37; the
                         DiagNormalPolicyANIL body, policies.py:76) */
} mi_policy_desc;

int mi_policy_create(const mi_policy_desc* desc, int device, mi_policy** out);
void mi_policy_destroy(mi_policy* p);
const char* mi_policy_last_error(const mi_policy* p);
int mi_policy_param_count(const mi_policy* p, size_t* n);
int mi_trpo_workspace_bytes(const mi_policy* p, int tasks, int batch, size_t* bytes);

/* density(state).loc (policies.py:49-52) for acting; theta shared (tstride 0) or one vector per task (tstride = P). */
int mi_policy_forward(mi_policy* p, void* stream, const float* theta, size_t tstride, const float* states, int tasks, int batch,
                      float* loc_out, void* workspace, size_t workspace_bytes);
/* trpo_update (rl.py:361-374): theta_out[t] = theta[t] - lr * grad_t( a2c.policy_loss = -mean(log_prob * advantages) ).
 * head_only != 0: the hidden layers run under no_grad (DiagNormalPolicyANIL.turn_off_body_grads, policies.py:100-106,
 * rl.py:381-382), so only sigma and the last Linear are updated (learn2learn maml_update skips None gradients). */
int mi_policy_adapt(mi_policy* p, void* stream, const float* theta, size_t tstride, const float* states, const float* actions,
                    const float* adv, const int32_t* count, int tasks, int batch, float lr, int head_only, float* theta_out,
                    float* loss_out, void* workspace, size_t workspace_bytes);
/* meta_surrogate_loss (rl.py:441-473) with one second-order inner step per task: mean surrogate loss, mean KL(new||old),
 * and (grad_out != NULL) the gradient w.r.t. theta (rl.py:413-416).  old_loc [tasks,batch,A], old_scale [tasks,A] are the
 * stored adapted policies' densities on the query states.  Leaves the context mi_trpo_fvp needs in `workspace`. */
int mi_trpo_surrogate(mi_policy* p, void* stream, const float* theta, const float* s_states, const float* s_actions,
                      const float* s_adv, const int32_t* s_count, const float* q_states, const float* q_actions,
                      const float* q_adv, const int32_t* q_count, const float* old_loc, const float* old_scale, int tasks,
                      int batch, float inner_lr, float* loss_out, float* kl_out, float* grad_out, void* workspace,
                      size_t workspace_bytes);
/* trpo.hessian_vector_product(old_kl, params, damping)(v) (rl.py:417) at the parameters mi_trpo_surrogate was last called
 * with on this workspace (where the adapted policies equal the stored old policies).
 * Calls on ONE mi_policy are ordered by the caller (one stream at a time): the fused product keeps a few arrival counters in device
 * memory owned by the policy object (its last fold also takes the mean over tasks; they are zero between launches). */
int mi_trpo_fvp(mi_policy* p, void* stream, const float* theta, const float* s_states, const float* s_actions,
                const int32_t* s_count, const float* q_states, const int32_t* q_count, int tasks, int batch, float inner_lr,
                float damping, const float* v, float* out, void* workspace, size_t workspace_bytes);
/* 1 (default): mi_trpo_fvp of a supported policy (ReLU, 100-wide hidden layers: the reference's DiagNormalPolicy defaults,
 * policies.py:30-37) runs as three fused sweeps over the stored passes + three folds (csrc/policy_sweep.h) instead of ~34 per-layer
 * launches; 0: the per-layer path.  Process-wide ablation / test switch; results agree to fp32 rounding. */
int mi_policy_set_fused_fvp(int on);
/* Debug aid: shader-clock stamps (stage id << 56 | s_memtime) of workgroup 0 of every fused sweep into buf (>= 256 u64; the three sweeps of a
 * product overwrite each other: read after the call, last sweep wins); NULL switches it off. */
int mi_debug_policy_sweep_stamps(void* buf);
/* Debug aid: s_memtime stamps {start, weights staged, first tile done, tiles done, epilogue done} of workgroup (0,0,0) of every stride-1
 * conv launch (csrc/conv_mfma.hip) into buf (>= 8 u64, the last launch wins); NULL switches it off. */
int mi_debug_conv_stamps(void* buf);
/* Operand form of the stride-1 hidden -> hidden 3x3 convolutions with 32 filters (forward, dgrad, their two-term tangent forms and the
 * weight gradient on maps at least 16 wide) and with 64 filters (one-term forward and dgrad, weight gradient on maps at least 16 wide) --
 * the implicit ATen conv2d launches behind
 * ConvBlock.conv, reference core_functions/vision_models.py:177-185,189.  1: split-bf16 -- every fp32 operand as the
 * exact sum of three bf16 pieces, six v_mfma_f32_32x32x16_bf16 products per K = 16 accumulated in fp32 (the three dropped cross terms
 * are <= 2^-24 of a product: one fp32 rounding).  2: two scaled fp16 planes -- x s = h + l with s a power of two chosen per task and
 * tensor from the tensor's largest magnitude (the producing kernels record it; standalone operator entries run a reduction launch),
 * three v_mfma_f32_32x32x16_f16 products per K = 16, the scales multiplied out of the fp32 sums exactly: 22 bits of every operand, an
 * absolute floor 2^-39 below the tensor's maximum, per-kernel errors against fp64 at or below those of the fp32 pipe; a launch whose
 * operands come without a recorded magnitude takes form 1.  Form 2 is an OPT-IN: its operands are narrower than the reference's fp32
 * (22 bits), so benchmark lines that use it say so and it is never the default.  0: the fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
 * Default 1 (environment variable MI_CONV_BF16X3=0 / 2 starts with another); returns the previous setting.  All three forms meet the
 * same fp32 parity bars (tests run all three).  Values above 0xff (0x100 * variant mask + form) select single kernel variants for bisecting
 * (tools/wgrad_probe.py, csrc/conv_mfma.hip) and are not part of the interface. */
int mi_conv_set_split_bf16(int on);
/* The operand form in force (2 scaled fp16 planes, 1 split-bf16, 0 fp32 matrix pipe) read without side effects; mask_out (may be NULL)
 * receives the variant mask a bisecting run set through MI_CONV_BF16X3_MASK / mi_conv_set_split_bf16(0x100 * mask + form). */
int mi_conv_get_split_bf16(unsigned* mask_out);
/* Which kernel runs the split-bf16 form (form 1) of the stride-1 hidden convolutions (forward, dgrad and their two-term tangent forms;
 * ConvBlock.conv of blocks >= 2, reference core_functions/vision_models.py:177-185,189): 16x16x32 MFMAs with one accumulator per horizontal
 * tap (csrc/conv_b16.h) or the 32x32x16 kernel with lane-shifted operands of rounds 3-4.  0 = always the latter, 2 = always the former,
 * 1 (default; MI_CONV_B16 starts with another) = the former for launches of at least 6 tiles per wave, rounded up (MI_CONV_B16_MIN_TPW), where its
 * longer pipeline fill is amortised.  Same operands, products and tiles; the two kernels differ in summation order only and meet the same
 * parity bars (the kernel tests run both on every case).  on < 0 only reads.  Returns the previous setting. */
int mi_conv_set_b16(int on);
/* Operand form of conv1 inside the two lean block-1 forward kernels (ConvBlock 1 of a three-channel net: conv + BatchNorm + ReLU + pool
 * with the conv output never stored, and its tangent from the stored argmax; reference core_functions/vision_models.py:188-193).
 * 0: the fp32 matrix pipe, bit-identical to the general block-1 kernel.  1: split bf16 with all
 * eight products down to 2^-24 (raw-pixel inputs: the six-product form of the hidden blocks is measurably noisier here) and the
 * BatchNorm normalisation folded into the product.  2: the split form in the tangent-forward kernel only -- it takes no pooling / ReLU
 * decisions (the argmax is the forward pass's stored one), so its rounding cannot re-route anything; the forward kernel stays on the fp32 pipe.
 * Default (on < 0, or never set; MI_B1_BF16X3 in the environment sets it): 1 since round 6, in every pass whose later kernels read the pooling /
 * ReLU decisions the forward stored (the Gram-matrix path of block 1: support passes and, since round 6, the query pass); passes whose backward
 * recomputes conv1 on the fp32 pipe and re-derives the decisions keep the forward on the fp32 pipe, so the two always agree.  (Rounds 4 - 5: 2.
 * Post-adaptation accuracy over 1024 tasks cannot tell forms 1 and 2 from fp64 or from each other,
 * profiles/r6/accuracy_parity_cfg2_1024tasks_b1forms.md; form 1 failed decision-level bars only while the query pass's backward re-derived its
 * decisions: csrc/block1.hip.)  With the hidden blocks on the fp32 pipe (mi_conv_set_split_bf16(0)) the default is 2.
 * Returns the form in force before the call. */
int mi_block1_set_split_bf16(int on);
/* Operand form of the sparse part of block 1's weight gradient on 84-wide three-channel inputs (dW1 = sum over pooling windows of the input
 * patch at the window's argmax times the cotangent: what autograd's conv2d backward computes for ConvBlock 1, reference
 * core_functions/vision_models.py:188-193, after max-pool and ReLU have zeroed three of every four positions).  1: split-bf16 operands on
 * the 16-bit matrix pipe with the six partial products of an fp32 product laid out along K (csrc/gram.hip) -- fp32-equivalent, same parity
 * bars; 0: fp32-input MFMAs.  on < 0 (default; MI_SPARSE_WGRAD_BF16 in the environment sets it): follow mi_conv_set_split_bf16 (fp32 pipe
 * there -> fp32 here).  Returns the form in force before the call. */
int mi_sparse_wgrad_set_split_bf16(int on);

/* ANIL-TRPO (rl/anil_trpo.py:104-129, core_functions/rl.py:409-473 with anil=True): the stored old policies were adapted with
 * the body under no_grad (rl.py:381-382) while meta_surrogate_loss re-adapts clone_module(policy) with every parameter
 * (rl.py:447-453), so at the current parameters new != old and trpo.hessian_vector_product(kl) (rl.py:417) is the EXACT Hessian
 * of the mean KL:  mean_t [ J_t^T Hess KL_t(theta'_t) J_t v - lr T_t[v, grad KL_t(theta'_t)] ] + damping v,  T_t = the third
 * derivative of the inner loss (second-order tangent sweep, policy.hip).
 *   mi_trpo_general_workspace_bytes: workspace for the three calls below (a superset of mi_trpo_workspace_bytes; use it for
 *     mi_trpo_surrogate as well).
 *   mi_trpo_kl_prepare: after mi_trpo_surrogate(theta, ...) on the same workspace; kl_grad_out (or NULL) [P] = d mean KL / d theta.
 *   mi_trpo_fvp_general: the product, any number of times after the two calls above. */
int mi_trpo_general_workspace_bytes(const mi_policy* p, int tasks, int batch, size_t* bytes);
int mi_trpo_kl_prepare(mi_policy* p, void* stream, const float* theta, const float* s_states, const float* s_actions,
                       const int32_t* s_count, const float* q_states, const int32_t* q_count, const float* old_loc,
                       const float* old_scale, int tasks, int batch, float inner_lr, float* kl_grad_out, void* workspace,
                       size_t workspace_bytes);
int mi_trpo_fvp_general(mi_policy* p, void* stream, const float* theta, const float* s_states, const float* s_actions,
                        const int32_t* s_count, const float* q_states, const int32_t* q_count, const float* old_scale, int tasks,
                        int batch, float inner_lr, float damping, const float* v, float* out, void* workspace,
                        size_t workspace_bytes);

/* MAML inner loop of the policy with K updates and the VPG / PPO losses, all tasks per call (core_functions/rl.py:
 * fast_adapt_vpg :231-255 with vpg_a2c_loss :209-228 (dice=False), fast_adapt_ppo :267-318; drivers rl/maml_ppo.py, anil_ppo.py).
 *   theta_{k+1} = theta_k - inner_lr * grad L_k(theta_k), k < steps; update k replays support batch step_batch[k] (host array):
 *     VPG: one batch per adapt step; PPO: ppo_epochs consecutive updates share a batch, step_new_old[k] = 1 on the first of them
 *     (old_log_probs are taken at theta_k under no_grad, rl.py:282-283).
 *   support batches [n_batches, tasks, batch, ...] padded like the TRPO replays; count [n_batches, tasks] (NULL = all valid).
 *   loss_out[t] = the validation loss of task t at theta_steps: VPG a2c.policy_loss, PPO ppo.policy_loss against the adapted
 *     policy itself (ratio 1).  theta_out [tasks, P] (or NULL) = adapted parameters.
 *   with_grad: grad_out [P] = SUM over tasks of d loss_out[t] / d theta -- what `av_loss.backward()` leaves in .grad after the
 *     caller's 1/meta_batch_size; second_order = learn2learn's default adapt (0: first-order MAML).
 *   head_only: updates touch only sigma and the last Linear (ANIL, body under no_grad). */
#define MI_PLOSS_A2C 0
#define MI_PLOSS_PPO 1
#define MI_PLOSS_DICE 2   /* vpg_a2c_loss(dice=True), rl.py:219-226: needs mi_policy_meta_batch_dones */
int mi_policy_meta_workspace_bytes(const mi_policy* p, int tasks, int batch, int steps, int n_batches, int second_order,
                                   size_t* bytes);
int mi_policy_meta_batch(mi_policy* p, void* stream, const float* theta, int steps, const int32_t* step_batch,
                         const int32_t* step_new_old, int n_batches, const float* s_states, const float* s_actions,
                         const float* s_adv, const int32_t* s_count, const float* q_states, const float* q_actions,
                         const float* q_adv, const int32_t* q_count, int tasks, int batch, int loss_kind, float clip,
                         float inner_lr, int head_only, int second_order, int with_grad, float* loss_out, float* theta_out,
                         float* grad_out, void* workspace, size_t workspace_bytes);
/* The same with the replays' episode-end flags (cherry `dones`; s_done [n_batches, tasks, batch], q_done [tasks, batch], 1.0 at
 * the last step of every episode).  Required by loss_kind MI_PLOSS_DICE -- vpg_a2c_loss(dice=True), rl.py:219-226: the adapt
 * losses AND the validation loss use  a2c.policy_loss(magic_box(weighted_cumsum(log_probs, weights)), advantages)  with
 * weights = (1 - dones shifted by one) / dones.sum(), including the reference's wrap-around at the first sample. */
int mi_policy_meta_batch_dones(mi_policy* p, void* stream, const float* theta, int steps, const int32_t* step_batch,
                               const int32_t* step_new_old, int n_batches, const float* s_states, const float* s_actions,
                               const float* s_adv, const int32_t* s_count, const float* s_done, const float* q_states,
                               const float* q_actions, const float* q_adv, const int32_t* q_count, const float* q_done, int tasks,
                               int batch, int loss_kind, float clip, float inner_lr, int head_only, int second_order,
                               int with_grad, float* loss_out, float* theta_out, float* grad_out, void* workspace,
                               size_t workspace_bytes);

/* MAML-TRPO with `steps` >= 1 inner updates (params['adapt_steps'], one support replay per update, rl.py:447-453): the
 * generalisation of mi_trpo_surrogate / mi_trpo_fvp.  Support arrays carry a leading [steps] axis:
 * s_states [steps,tasks,batch,S], s_actions [steps,tasks,batch,A], s_adv [steps,tasks,batch], s_count [steps,tasks].
 * mi_trpo_fvp_steps must follow mi_trpo_surrogate_steps on the same workspace and replays (it re-uses the saved passes). */
int mi_trpo_steps_workspace_bytes(const mi_policy* p, int tasks, int batch, int steps, size_t* bytes);
int mi_trpo_surrogate_steps(mi_policy* p, void* stream, const float* theta, int steps, const float* s_states,
                            const float* s_actions, const float* s_adv, const int32_t* s_count, const float* q_states,
                            const float* q_actions, const float* q_adv, const int32_t* q_count, const float* old_loc,
                            const float* old_scale, int tasks, int batch, float inner_lr, float* loss_out, float* kl_out,
                            float* grad_out, void* workspace, size_t workspace_bytes);
int mi_trpo_fvp_steps(mi_policy* p, void* stream, int steps, const float* s_states, const float* s_actions, const int32_t* s_count,
                      const float* q_states, const int32_t* q_count, int tasks, int batch, float inner_lr, float damping,
                      const float* v, float* out, void* workspace, size_t workspace_bytes);

#ifdef __cplusplus
}
#endif
#endif /* MI_MAML_H */

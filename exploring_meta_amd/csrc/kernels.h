// Internal launcher interface between the kernel files and the engine (not part of the C ABI).
#pragma once
#include "mi_common.h"
#include "finalize.h"

struct ConvArgs {
  const float* in[2];   // [T][n][h][w][ci] per term
  const float* wt[2];   // per-task weights per term (forward layout [9][Ci_fwd][Co_fwd]), advance wstride floats per task
  size_t wstride;
  float* out;           // [T][n][ho][wo][co]
  const float* z;       // EPI_TSTATS: primal conv output (same shape as out)
  const float* mu;      // EPI_TSTATS: [T][co]
  const float* rstd;    // EPI_TSTATS: [T][co]
  double* partial;      // EPI_*STATS: [T][blocks_per_task][2][co]
  FinArgs fin;          // EPI_*STATS: fold the partials in the last workgroup of each task (counter != nullptr)
  // EPI_BRED (stride-1 dgrad of block 2 only): the BatchNorm-backward sums of a fused block 1 ride in the epilogue -- the output
  // tile IS the cotangent of block 1's pooled output, and block 1 left p, zhat (tangent: zhat-dot) at the same positions:
  //   1 term : dgamma = sum [p>0] out zh,                    dbeta = sum [p>0] out
  //   2 terms: R{dgamma} = sum [p>0] (out zh + dp zhd),      R{dbeta} = sum [p>0] out        (out = R{dp}, dp = primal cotangent)
  const float *bp, *bzh, *bzhd, *bdp;
  const uint8_t* barg;  // optional (replaces the read of bp): the argmax byte block 1's forward kernel stores per pooled element, 4 = ReLU off <=> p == 0
  ConvGeom g;
  int mpix;             // n*ho*wo
  int ntiles, tiles_per_wave;
  int split_bf16;       // set by launch_conv3x3: the operand form of this launch -- 0 fp32 pipe, 1 split-bf16, 2 scaled fp16 planes (30-pixel tiles)
  int stagger;          // set by launch_conv3x3 (conv_b16.h): delay the second wave of every SIMD by half a tile once, so that the partners' epilogues do not coincide
  const unsigned* amax[2];   // fp16 form: cells [T][MI_CELL_WORDS] (mi_common.h) = the exponent field (fp32 bits) of max |in[term]| per task (written by the tensor's producer, launch_amax otherwise)
};

struct WgradArgs {
  const float* x[2];    // [T][n][h][w][ci] per term
  const float* dz[2];   // [T][n][ho][wo][co] per term
  float* partial;       // [T][nchunks][9][ci][co]
  ConvGeom g;
  int mpix, chunk_pix, nchunks, nterms;
  const unsigned* amax_x[2]; const unsigned* amax_dz[2];   // fp16 form: cells (as ConvArgs::amax) of x[term] / dz[term] (all given, or none)
  int form;             // set by launch_wgrad3x3 (as ConvArgs::split_bf16)
};

struct BnArgs {
  const float* z;       // [T][n][ho][wo][c]
  const float* zd;      // tangent of z
  const float* mu;      // [T][c]
  const float* rstd;
  const float* m1;      // [T][c] mean(zd)
  const float* m2;      // [T][c] mean(zh*zd)
  const float* gamma; const float* beta; size_t pstride;     // theta (per task)
  const float* gammad; const float* betad; size_t vstride;   // tangent direction (per task)
  const float* dgamma; const float* dbeta; size_t gstride;   // primal gradient (per task)
  const float* rdgamma; const float* rdbeta; size_t hstride; // tangent gradient (per task)
  const float* dp;      // [T][n][hp][wp][c]
  const float* dpd;
  float* out;
  unsigned* amax_out;   // optional cells [T][MI_CELL_WORDS] (mi_common.h): the kernels that write `out` fold the exponent of max |out| per task into them (zeroed by the
                        // caller; the fp16 operand form of the convolution that reads `out` next takes its scale from there)
  float* zh_out;        // forward kernels, optional [T][n][hp][wp][c]: zhat (tangent forward: its tangent) at every pooled output's argmax --
                        // what the BatchNorm-backward sums need besides p and dp, so they can ride in the next block's dgrad epilogue
  double* partial;      // [T][nblk][2][c]
  FinArgs fin;          // reduction kernels: fold the partials in the last workgroup of each task (counter != nullptr)
  int n, ho, wo, c;
  float inv_m;
};


// BatchNorm-backward reductions of a fused block 1 from pooled-resolution tensors (bn_pool.hip): the forward kernels leave
// zhat (and its tangent) at every window's argmax next to the pooled output, so dgamma/dbeta need no conv recompute.
struct PoolRedArgs {
  const float* p;       // [T][rows][c] pooled output (p > 0 <=> the window's maximum passed the ReLU)
  const float* zh;      // zhat at the argmax
  const float* zhd;     // tangent of zhat at the argmax (tangent mode)
  const float* dp;      // cotangent of p
  const float* dpd;     // tangent of that cotangent (tangent mode)
  double* partial;      // [T][nblk][2][c]
  int rows, c;
  FinArgs fin;
};
int pooled_reduce_blocks(int rows, int c, int tasks);
hipError_t launch_pooled_reduce(hipStream_t st, const PoolRedArgs& a, int tasks, int tangent, int* nblk);

// conv_mfma.hip
int conv_operand_form();   // 0 fp32 pipe, 1 split-bf16, 2 scaled fp16 planes (mi_conv_set_split_bf16)
int conv_b16();            // which kernel runs form 1 (mi_conv_set_b16)
int block1_split_form();   // block1.hip: mi_block1_set_split_bf16
int sparse_wgrad_split_form();   // gram.hip: mi_sparse_wgrad_set_split_bf16 (1 = the 84-wide rows kernel on the split-bf16 form)
hipError_t launch_conv3x3(hipStream_t st, ConvArgs a, int tasks, int nterms, int epi, int mode, int* blocks_per_task);
hipError_t launch_wgrad3x3(hipStream_t st, WgradArgs a, int tasks, int nterms, int* nchunks_out);
hipError_t launch_wgrad_reduce(hipStream_t st, const float* partial, int nchunks, int nelem, int tasks, float* out, size_t ostride);
size_t wgrad_partial_floats(const ConvGeom& g, int tasks);
int conv_max_blocks_per_task(const ConvGeom& g);
int conv_tiles_per_wave(int mpix, int tasks, int cot);

// bn_pool.hip
int bn_blocks_per_task(int n, int ho, int wo, int c, int pool, int tasks);
hipError_t launch_bn_finalize(hipStream_t st, const double* partial, int nblk, int tasks, int c, double inv_m, int mode,
                              float* out0, size_t stride0, float* out1, size_t stride1);
hipError_t launch_bn_fwd(hipStream_t st, const BnArgs& a, int tasks, int pool);
hipError_t launch_bn_bwd_reduce(hipStream_t st, const BnArgs& a, int tasks, int pool, int* nblk);
hipError_t launch_bn_bwd_apply(hipStream_t st, const BnArgs& a, int tasks, int pool);
hipError_t launch_bn_tan_fwd(hipStream_t st, const BnArgs& a, int tasks, int pool);
hipError_t launch_bn_tan_bwd_reduce(hipStream_t st, const BnArgs& a, int tasks, int pool, int* nblk);
hipError_t launch_bn_tan_bwd_apply(hipStream_t st, const BnArgs& a, int tasks, int pool);

// block1.hip -- fused first ConvBlock with conv recompute
struct B1Args {
  const float* x;            // [T][n][H][W][Ci0]
  const float* w; const float* wd;        // conv weights [9*Ci0][Co] of theta / of the tangent direction
  size_t wstride, vstride;   // per-task strides of theta-shaped / direction-shaped vectors
  const float *mu, *rstd, *m1, *m2;       // [T][Co]
  const float *gamma, *beta; size_t pstride;
  const float *gammad, *betad;            // direction (stride vstride)
  const float *dgamma, *dbeta; size_t gstride;
  const float *rdgamma, *rdbeta; size_t hstride;
  const float* dp; const float* dpd;      // [T][n][H/2][W/2][Co]
  float* out;                // p or pd
  unsigned* amax_out;        // optional (forward modes): cells of max |out| per task, as BnArgs::amax_out
  float* zh_out;             // optional: zhat (FWD) / its tangent (TFWD) at each window's argmax, same shape as out
  uint8_t* arg_out;          // optional (FWD): argmax position 0..3 of every window, 4 where the maximum did not pass the ReLU
  const uint8_t* arg_in; const float* zh_in;   // TFWD_ARG: the two tensors FWD stored (no primal conv recompute)
  double* partial;           // [T][blocks][2][Co]
  FinArgs fin;               // reduction modes: fold the partials in the last workgroup of each task (counter != nullptr)
  float* wpartial;           // [T][blocks][9*Ci0][Co]
  int n, hh, ww, co;         // images per task, conv output height / width (= input, stride 1), filters
  float inv_m;
  int ntiles, tiles_per_wave;
  int fwd_fp32;              // FWD: keep conv1 on the fp32 pipe whatever the block-1 form (the pass's backward will RE-DERIVE the pooling / ReLU decisions with the
                             // general kernel's fp32 conv instead of reading the stored argmax: the two must round alike)
};
enum { B1_STATS = 0, B1_FWD = 1, B1_BWD_REDUCE = 2, B1_BWD_WGRAD = 3, B1_TSTATS = 4, B1_TFWD = 5, B1_TBWD_REDUCE = 6, B1_TBWD_WGRAD = 7,
       B1_TFWD_ARG = 8, B1_FORCE_GENERAL = 0x100 };   // tangent forward from the stored argmax / zhat: one conv (with the direction's weights) instead of two
bool block1_supported(int ci, int stride, int pool, int h, int w, int co);
int block1_blocks_per_task(int n, int h, int w, int co, int tasks);
hipError_t launch_block1(hipStream_t st, B1Args a, int tasks, int ci, int mode, int* blocks_per_task);

// Block-1 weight gradient without recomputing conv1 (gram.hip): dz = gr (du - mean(du) - zhat mean(du zhat)) splits into a
// SPARSE part (du lives at one position per pooling window: S = sum patch(argmax) x cot, sparse_wgrad kernel, fp32 MFMA)
// and DENSE parts that are products of the input Gram matrix with the weights (gram_wgrad kernel, fp64).
struct SparseWgArgs {
  const float* x;            // [T][n][H][W][Ci0]
  const uint8_t* arg;        // [T][rows][Co] from block1 FWD (rows = n*H/2*W/2)
  const float* dp; const float* dpd;             // cotangent of p and its tangent (tangent mode)
  const float *mu_unused, *rstd, *m2;            // [T][Co]   (tangent mode: c1 = gammad r - gamma r^2 m2)
  const float* gamma; size_t pstride;
  const float* gammad; size_t vstride;
  float* wpartial;           // [T][blocks][9*Ci0][Co]
  int n, hh, ww, co;
  int ntiles, tiles_per_wave, row_pitch;
};
bool sparse_wgrad_supported(int w, int ci, int co);
int sparse_wgrad_blocks_per_task(int n, int h, int w, int co, int tasks);
hipError_t launch_sparse_wgrad(hipStream_t st, SparseWgArgs a, int tasks, int ci, int tangent, int* blocks_per_task);
struct GramWgArgs {
  const double* g;           // [T][NG][NG]
  const float* spartial; int nblk;               // sparse partials [T][nblk][9*Ci0][Co]
  const float* w; size_t wstride; const float* wd; size_t vstride;
  const float *mu, *rstd, *m1, *m2;              // [T][Co]
  const float* gamma; size_t pstride; const float* gammad;
  const float *dgamma, *dbeta; size_t gstride;   // primal BN gradients (sums)
  const float *rdgamma, *rdbeta; size_t hstride; // their tangents (tangent mode)
  float* out; size_t ostride;                    // dW (primal) or R{dW} (tangent): [T][.. 9*Ci0*Co]
  int ci, co; double inv_m;
};
hipError_t launch_gram_wgrad(hipStream_t st, const GramWgArgs& a, int tasks, int tangent);

// gram.hip: the tail of a pass as ONE launch ("advance").  A forward/backward (or Hessian-vector) pass leaves its gradient-shaped
// vector g [T][gstride] unfinished: the conv-weight segments of blocks >= 2 are still per-workgroup partials of the weight-gradient
// kernels, block 1's weight gradient still needs its Gram-matrix assembly (GramWgArgs), conv biases / padding are unwritten.  The
// advance kernel finishes g (same fold order as reduce_partials_kernel, same arithmetic as gram_wgrad_kernel), applies
// out = a - alpha * g (learn2learn maml_update / the adjoint recursion; out == nullptr: finish g only) and, from the freshly written
// block-1 weights, forms the BatchNorm statistics the NEXT pass over the support images needs from the Gram matrix (gram_stats) --
// replacing 3 reduce_partials + gram_wgrad + axpy + gram_stats + 1 memset launches per pass.
struct AdvanceSeg { unsigned off, nelem; const float* partial; int nchunks; };
struct AdvanceArgs {
  float* g; size_t gstride;
  AdvanceSeg seg[8]; int nseg;               // g[off .. off+nelem) = sum_chunk partial[task][chunk][.]
  unsigned zoff[10], zlen[10]; int nzero;    // g = 0 there (conv biases: train-mode BatchNorm cancels them; padding P..PS)
  const float* a; float* out; size_t ostride; float alpha;
  unsigned n;                                // elements per task
  int b1_wgrad, gw_tangent; GramWgArgs gw;   // block 1's weight gradient from the Gram matrix (gw.out points into g)
  unsigned off_w1;                           // offset of block 1's conv weights in a parameter vector
  int stats;                                 // 0: none; 1: primal statistics with weights = out; 2: tangent statistics, direction = out (or g when out == nullptr)
  const double* gram; int ci, co;
  const float* sw; size_t swstride;          // stats == 2: the weights of the pass the statistics are for
  const float *mu_in, *rstd_in;              // stats == 2
  float *out0, *out1; double inv_m;          // mean / rstd  (stats == 2: m1 / m2), [T][co]
  unsigned* counter;                         // stats != 0: one zero-initialised arrival counter per task (left at zero)
};
hipError_t launch_advance(hipStream_t st, const AdvanceArgs& a, int tasks);

// gram.hip: input Gram matrix of block 1 (BatchNorm statistics of conv1 as quadratic forms of the weights)
int gram_blocks_per_task(int n, int h);
size_t gram_partial_doubles(int tasks, int n, int h, int ci);
size_t gram_doubles(int tasks, int ci);
bool gram_supported(int w, int ci);     // the kernel stages three fp64 input rows per wave in LDS
hipError_t launch_input_gram(hipStream_t st, const float* x, int tasks, int n, int h, int w, int ci, double* partial, double* g);
hipError_t launch_gram_stats(hipStream_t st, const double* g, int tasks, int ci, int co, const float* w, size_t wstride,
                             const float* wd, size_t vstride, double inv_m, int tangent, float* out0, float* out1,
                             const float* mu, const float* rstd);

// head.hip
struct HeadArgs {
  const float* f;       // [T][n][F]
  const float* fd;      // tangent
  const float* wl; const float* bl; size_t pstride;      // theta
  const float* wld; const float* bld; size_t vstride;    // direction
  const int32_t* y;     // [T][n]
  float* loss; float* acc;          // [T]
  float* logits;        // [T][n][ways] (may be null)
  float* prob; float* dl;           // [T][n][ways] saved (tangent pass reads them)
  float* rdl;                       // [T][n][ways] scratch: R{dl} (tangent)
  float* ld_out = nullptr;          // tangent: also store the logit tangents J v [T][n][ways]
  int fixed_dl = 0;                 // tangent: dl is a given cotangent, not the cross-entropy's -- R{dl} = 0 (mi_learner_hvp)
  float* rowloss; float* rowhit;    // [T][n] scratch: per-row loss / hit
  float* dwl; float* dbl; size_t gstride;   // outputs (primal grads or tangent grads)
  float* df;            // [T][n][F] output (df or R{df}); may be null
  int n, feat, ways;
};
hipError_t launch_head_fwd_bwd(hipStream_t st, const HeadArgs& a, int tasks, int with_grad);
hipError_t launch_head_tangent(hipStream_t st, const HeadArgs& a, int tasks);
hipError_t launch_head_grads(hipStream_t st, const HeadArgs& a, int tasks);   // backward only, a.dl supplied by the caller
hipError_t launch_spatial_mean(hipStream_t st, const float* p, float* f, int rows, int hw, int c);
hipError_t launch_spatial_mean_bwd(hipStream_t st, const float* df, float* dp, int rows, int hw, int c);

// tail.hip: the last ConvBlock's BatchNorm + ReLU + MaxPool, the head, the head's backward and that block's BatchNorm-backward SUMS (or their
// tangents) in one launch, four workgroups (row groups) per task; cross-workgroup sums by the last arriver (finalize.h protocol).
struct TailArgs {
  BnArgs bn;            // z (zd), mu, rstd (m1, m2), gamma / beta (gammad / betad), n, ho, wo, c; tangent: dp = the primal cotangent of p
  HeadArgs hd;          // primal: y, loss, acc, logits, prob, dl, rowloss, rowhit, dwl, dbl, df; tangent: f = the stored primal features, prob, dl, rdl, wld, bld
  float* pooled;        // [T][n][hp][wp][c]: p (primal) or pd (tangent), written by this launch
  float* sum0; float* sum1; size_t sum_stride;   // dgamma / dbeta (tangent: R{dgamma} / R{dbeta}) [T][..c], written by this launch
  int with_grad;        // primal: 0 = forward, loss and accuracy only
  int bwd_tasks;        // primal: tasks >= bwd_tasks stop after the loss (validation tasks of a fused train + validation call)
  float* wpart;         // [T][4][ways][feat] row-group partials of dWl (tail_wpart_floats)
  double* bpart;        // [T][4][2][c] row-group partials of the BatchNorm-backward sums (tail_bpart_doubles)
  float* scr;           // [T][4][ceil(n/4)][ways + 2] per-row dlogits / loss / hit for the folding workgroup (tail_scr_floats)
  unsigned* counter;    // [T] arrival counters, zero on entry, left at zero
  unsigned long long* stamps;   // debug (mi_debug_tail_stamps): [T][4][16] wall_clock64() at the stage boundaries, thread 0 of every workgroup; or nullptr
};
bool tail_supported(int n, int ho, int wo, int c, int pool, int feat, int ways);
size_t tail_wpart_floats(int tasks, int feat, int ways);
size_t tail_bpart_doubles(int tasks, int c);
size_t tail_scr_floats(int tasks, int n, int ways);
hipError_t launch_tail(hipStream_t st, const TailArgs& t, int tasks, int pool, int tangent);

// misc.hip
hipError_t launch_prepare_batch(hipStream_t st, const float* data, const int64_t* labels, int tasks, int n2, int c, int h,
                                int w, float* xs, float* xq, int32_t* ys, int32_t* yq);
hipError_t launch_nchw_to_nhwc(hipStream_t st, const float* src, size_t images, int c, int h, int w, float* dst);
hipError_t launch_sample_tasks(hipStream_t st, const void* dataset, int u8, const int64_t* index, const uint8_t* rot, size_t rows,
                               int c, int h, int w, float* out);
hipError_t launch_split_rows(hipStream_t st, const float* src, int tasks, int n2, int f, float* even, float* odd);
hipError_t launch_interleave_rows(hipStream_t st, const float* even, const float* odd, int tasks, int n, int f, float* dst);
hipError_t launch_split_labels(hipStream_t st, const int64_t* labels, int tasks, int n2, int32_t* ys, int32_t* yq);
hipError_t launch_gather_params(hipStream_t st, const float* theta_ref, size_t src_stride, const int32_t* perm, int p, int pstride,
                                int tasks, float* theta_eng);   // src_stride 0 = one shared theta
hipError_t launch_scatter_tasks(hipStream_t st, const float* g, const int32_t* perm, int p, int pstride, int tasks, float* out_ref);
hipError_t launch_nhwc_to_nchw(hipStream_t st, const float* src, size_t images, int c, int h, int w, float* dst);
hipError_t launch_scatter_sum(hipStream_t st, const float* lam, const int32_t* perm, int p, int pstride, int tasks, float* out_ref);
hipError_t launch_axpy(hipStream_t st, const float* a, const float* b, float alpha, size_t n, float* out);
hipError_t launch_stream_copy(hipStream_t st, const void* src, void* dst, size_t bytes);
hipError_t launch_amax(hipStream_t st, const float* x, size_t per_task, int tasks, unsigned* cell);   // cells [T][MI_CELL_WORDS] (mi_common.h) |= exponent bits of max |x[task]|
const unsigned* standalone_amax(hipStream_t st, int slot, const float* x, size_t per_task, int tasks, hipError_t* err);
struct BnExportArgs {
  const float* mu[8]; const float* rstd[8];     // [T][c_l] per block
  int c[8], off[8];
  int nl, ctot;
  float* out;                                   // [T][2][ctot]
};
hipError_t launch_bn_export(hipStream_t st, const BnExportArgs& a, int tasks);
hipError_t launch_adam(hipStream_t st, float* theta, const float* grad, float* m, float* v, size_t n, int step, float lr,
                       float b1, float b2, float eps, float gscale);

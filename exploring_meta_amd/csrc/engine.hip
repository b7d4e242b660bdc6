// Engine: orchestrates one whole meta-batch (all tasks, all inner steps, query pass, second-order adjoint recursion) as
// a fixed sequence of batched-over-tasks kernel launches on the caller's stream, and exports the C ABI (include/mi_maml.h).
//
// Algorithm (per task t, identical to the reference's clone -> fast_adapt -> eval_loss.backward()):
//   theta_0 = theta;  for k < K:  g_k = grad L_support(theta_k);  theta_{k+1} = theta_k - alpha g_k      (l2l MAML.adapt)
//   lam = grad L_query(theta_K);   second order:  for k = K-1..0:  lam <- lam - alpha * H_support(theta_k) lam
//   meta_grad = sum_t lam_t.        H v is a forward-over-reverse tangent sweep over the SAVED step-k activations.
#include <string>
#include <vector>
#include <cstdio>
#include <cmath>
#include <cstring>
#include "mi_common.h"
#include "kernels.h"
#include "../../include/mi_maml.h"

#define EPI_NONE 0
#define EPI_STATS 1
#define EPI_TSTATS 2
#define EPI_BRED 3

static thread_local std::string g_err;

struct Layer {
  int ci, co, h, w, ho, wo, stride, pool, hp, wp;
  size_t off_gamma, off_beta, off_w, off_b;
};

// Round 6: the query pass's block 1 through a Gram matrix of the QUERY images too (statistics as quadratic forms, weight gradient as sparse
// part + assembly) instead of the two conv-recompute kernels (0.14 + 0.44 ms per cfg2 iteration): the fp64 Gram launch (0.23 ms) runs on the side
// stream beside the inner loop.  cfg2 15.84 -> 15.60 ms in alternating pairs (on the caller's stream: 15.71); rounds 2 - 5 had measured no gain
// from it -- the sparse weight gradient was slower then and the Gram launch sat on the critical path (profiles/r6/ab_gram_query.txt).
static int gram_query_env() { static const int v = getenv("MI_GRAM_QUERY") ? atoi(getenv("MI_GRAM_QUERY")) : 2; return v; }      // 0: off; 1: on the caller's stream; 2 (default): on the side stream, beside the inner loop
static bool fork_once_env() { static const bool v = getenv("MI_FORK_ONCE") && atoi(getenv("MI_FORK_ONCE")) != 0; return v; }
struct mi_engine {
  mi_model_desc d;
  int device;
  std::vector<Layer> L;
  int head_c, head_hw, feat;
  size_t off_wl, off_bl, P;
  size_t PS;  // per-task stride of parameter-shaped buffers (P padded so every task's vectors stay 16-B aligned)
  int32_t* perm_dev;
  bool fuse1 = false;   // block 1 runs through the conv-recompute kernels of block1.hip
  bool gramq = true;    // ... of the QUERY images too (round 6; the backward half's query pass: statistics + weight gradient of block 1)
  bool gram1 = true;    // ... with the statistics of repeated passes from the input Gram matrix (gram.hip) and the BN-backward
                        // reductions from zhat stored at the pooling argmax, instead of further conv-recompute passes
  // Weight gradients of blocks >= 2 run on a side stream: they depend only on dz_l and the block input, nothing downstream of
  // them until the parameter update, and they are matrix-bound while the BatchNorm kernels of the next block are HBM-bound.
  bool overlap = true;
  bool fork_once = fork_once_env();   // false (default): one fork / join of the side stream per hidden block; true: ONE per backward pass (mi_engine_set_overlap(e, 3), MI_FORK_ONCE=1; measured slower)
  // BatchNorm statistic / reduction partials are folded by the last workgroup of the producing kernel (finalize.h) instead of a
  // bn_finalize launch; counters: one zero-initialised arrival counter per task, owned by the engine.
  bool fuse_fin = true;
  // block 1's BatchNorm-backward sums (dgamma, dbeta and their tangents) ride in the epilogue of block 2's dgrad instead of a
  // streaming pooled_reduce pass over p, zhat, dp (needs the fused block 1 with stored zhat and a stride-1 hidden block 2)
  bool fuse_b1red = true;
  // the tail of every pass of mi_meta_batch_maml as ONE launch (gram.hip, advance_kernel): weight-gradient partial folds, block 1's
  // Gram-matrix assembly, the fast-weight / adjoint update and the next pass's Gram statistics -- instead of 3 reduce_partials +
  // gram_wgrad + axpy + gram_stats launches and a memset per pass.  Same arithmetic in the same order: bit-identical results.
  bool bred_arg = true;     // block 2's dgrad epilogue reads block 1's argmax byte instead of p (MI_BRED_ARG=0: p, for A/B runs; same results)
  bool fuse_tail = true;
  // the last ConvBlock's BatchNorm + pooling, the head, its backward and that block's BatchNorm-backward sums as ONE launch, four workgroups per
  // task (tail.hip), instead of four launches per pass (MI_FUSE_LAST=0 / mi_engine_set_fused_last_block(e, 0): the separate launches)
  bool fuse_last = !(getenv("MI_FUSE_LAST") && atoi(getenv("MI_FUSE_LAST")) == 0);
  unsigned long long* tail_stamps = nullptr;   // debug (mi_debug_tail_stamps)
  unsigned zoff[10] = {}, zlen[10] = {};   // conv-bias segments and the padding P..PS of a parameter-shaped vector (never written by a kernel)
  int nzero = 0;
  unsigned* counters = nullptr;
  static constexpr int kMaxCounterTasks = 65536;
  struct SideCtx { hipStream_t side = nullptr; hipEvent_t fork = nullptr, join = nullptr; } sc[1];
  std::string err;
  // Graph replay (mi_engine_set_graph): the launch sequence of mi_meta_batch_maml / _anil is a pure function of its arguments, so a
  // call whose arguments (every pointer, size and scalar) equal an earlier call's is replayed as one hipGraphLaunch -- what the
  // launch-bound few-image configurations need (cfg1: 75 launches in 0.6 ms).  First sight of a signature runs eagerly (kernel
  // attributes, lazily created streams), the second is captured, later ones replay.
  bool graph_on = false;
  struct GraphEntry { std::vector<unsigned long long> key; hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr; int seen = 0; };
  std::vector<GraphEntry> graphs;
  // BatchNorm batch statistics of every forward pass (mi_engine_set_bn_export): what torch.nn.BatchNorm2d's running_mean / running_var
  // update consumes.  Layout [pass][T][2][sum of channels over the blocks]: batch mean, biased batch variance.
  float* bn_export = nullptr;
  size_t bn_export_floats = 0;
  int export_pass = 0;
  // debug trace (mi_debug_set_trace): per-step theta_k / g_k / lam fed to the k-th Hessian-vector product / H lam, reference order
  float* trace = nullptr;
  size_t trace_floats = 0;
  // optional per-launch HIP-event profiling (bench.py's roofline leg): kind = op*8 + layer
  int prof_on = 0, prof_filter = -1;
  std::vector<hipEvent_t> ev0, ev1;
  std::vector<int> ev_kind;
  size_t ev_used = 0;
};

enum ProfOp { OP_CONV_FWD = 0, OP_BN_FINALIZE, OP_BN_FWD, OP_HEAD, OP_BN_BWD_REDUCE, OP_BN_BWD_APPLY, OP_WGRAD, OP_WGRAD_REDUCE,
              OP_DGRAD, OP_TAN_CONV, OP_BN_TAN_FWD, OP_HEAD_TAN, OP_BN_TAN_BWD_REDUCE, OP_BN_TAN_BWD_APPLY, OP_TAN_WGRAD,
              OP_TAN_DGRAD, OP_MISC, OP_GRAM, OP_GRAM_STATS, OP_COUNT };
static const char* kOpNames[OP_COUNT] = {"conv_fwd_stats", "bn_finalize", "bn_relu_pool_fwd", "head_fwd_bwd", "bn_bwd_reduce",
                                         "bn_bwd_apply", "wgrad", "wgrad_reduce", "dgrad", "tangent_conv_fwd", "bn_tangent_fwd",
                                         "head_tangent", "bn_tangent_bwd_reduce", "bn_tangent_bwd_apply", "tangent_wgrad",
                                         "tangent_dgrad", "misc", "input_gram", "gram_stats"};

static bool prof_begin(mi_engine* e, hipStream_t st, int kind) {
  if (!e || !e->prof_on || (e->prof_filter >= 0 && e->prof_filter != kind)) return false;
  if (e->ev_used == e->ev0.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return false;
    e->ev0.push_back(a); e->ev1.push_back(b); e->ev_kind.push_back(0);
  }
  e->ev_kind[e->ev_used] = kind;
  (void)hipEventRecord(e->ev0[e->ev_used], st);
  return true;
}
static void prof_end(mi_engine* e, hipStream_t st, bool active) {
  if (!active) return;
  (void)hipEventRecord(e->ev1[e->ev_used], st);
  e->ev_used++;
}

static int fail(mi_engine* e, int code, const std::string& msg) {
  if (e) e->err = msg;
  g_err = msg;
  return code;
}
#define LAUNCH(e, st, op, layer, call)                                                             \
  do {                                                                                             \
    const bool _pa = prof_begin(e, st, (op) * 8 + (layer));                                        \
    hipError_t _s = (call);                                                                        \
    prof_end(e, st, _pa);                                                                          \
    if (_s != hipSuccess)                                                                          \
      return fail(e, MI_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_s) + " @" + std::to_string(__LINE__)); \
  } while (0)
#define HIPCHK(e, call)                                                                            \
  do {                                                                                             \
    hipError_t _s = (call);                                                                        \
    if (_s != hipSuccess)                                                                          \
      return fail(e, MI_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_s) + " @" + std::to_string(__LINE__)); \
  } while (0)

static FinArgs fin_of(const mi_engine* e, int T, double inv_m, int mode, float* o0, size_t s0, float* o1, size_t s1) {
  const bool on = e->fuse_fin && e->counters && T <= mi_engine::kMaxCounterTasks;
  return FinArgs{on ? e->counters : nullptr, o0, o1, s0, s1, inv_m, mode};
}

static ConvGeom geom(const Layer& l, int n) { return ConvGeom{n, l.h, l.w, l.ho, l.wo, l.ci, l.co, l.stride}; }
static ConvGeom geom_dgrad(const Layer& l, int n) {  // op input = dz (ho,wo,co), op output = dx (h,w,ci)
  return ConvGeom{n, l.ho, l.wo, l.h, l.w, l.co, l.ci, l.stride};
}

// The dgrad of block l (l >= 1) produces the cotangent of block l-1's pooled output.  If block l-1 left zhat at the argmax next to
// p (fused block 1: zhm; hidden blocks: zhl), its BatchNorm-backward sums are an epilogue of that dgrad (EPI_BRED) -- available
// for the geometry conv3x3_s1 covers: stride-1 hidden -> hidden conv of 32 or 64 channels.
static bool dgrad_carries_reduce(const mi_engine* e, int l) {
  if (!e->fuse_b1red || l < 1 || l >= (int)e->L.size()) return false;
  const Layer& L = e->L[l];
  return L.stride == 1 && L.ci == L.co && L.co == e->L[l - 1].co && L.h == L.ho && L.w == L.wo && (L.ci == 32 || L.ci == 64);
}

// ---------------------------------------------------------------------------------------------------------------------
extern "C" const char* mi_version(void) { return "mi_maml 0.1 (gfx950)"; }
extern "C" const char* mi_last_error(const mi_engine* e) { return e ? e->err.c_str() : g_err.c_str(); }

extern "C" int mi_engine_create(const mi_model_desc* d, int device, mi_engine** out) {
  if (!d || !out) return fail(nullptr, MI_ERR_ARG, "null argument");
  if (d->n_layers < 1 || d->n_layers > 8) return fail(nullptr, MI_ERR_ARG, "n_layers must be 1..8");
  if (d->hidden != 32 && d->hidden != 64) return fail(nullptr, MI_ERR_ARG, "hidden must be 32 or 64");
  if (d->in_channels != 1 && d->in_channels != 3 && d->in_channels != 32 && d->in_channels != 64)
    return fail(nullptr, MI_ERR_ARG, "in_channels must be 1, 3, 32 or 64");
  if (d->ways < 1 || d->ways > 64) return fail(nullptr, MI_ERR_ARG, "ways must be 1..64");
  mi_engine* e = new mi_engine();
  e->d = *d;
  e->device = device;
  e->perm_dev = nullptr;
  int ci = d->in_channels, h = d->in_h, w = d->in_w;
  size_t off = 0;
  std::vector<int32_t> perm;
  for (int i = 0; i < d->n_layers; ++i) {
    Layer l;
    l.ci = ci; l.co = d->hidden; l.h = h; l.w = w;
    l.stride = d->max_pool ? 1 : 2;
    l.pool = d->max_pool ? 1 : 0;
    l.ho = (h + 2 - 3) / l.stride + 1;
    l.wo = (w + 2 - 3) / l.stride + 1;
    l.hp = l.pool ? l.ho / 2 : l.ho;
    l.wp = l.pool ? l.wo / 2 : l.wo;
    if (l.hp < 1 || l.wp < 1) { delete e; return fail(nullptr, MI_ERR_ARG, "input too small for this many blocks"); }
    l.off_gamma = off; off += l.co;
    l.off_beta = off; off += l.co;
    l.off_w = off; off += (size_t)9 * l.ci * l.co;
    l.off_b = off; off += l.co;
    for (size_t k = perm.size(); k < l.off_w; ++k) perm.push_back((int32_t)k);
    for (int tap = 0; tap < 9; ++tap)
      for (int c1 = 0; c1 < l.ci; ++c1)
        for (int c2 = 0; c2 < l.co; ++c2) perm.push_back((int32_t)(l.off_w + ((size_t)c2 * l.ci + c1) * 9 + tap));
    for (size_t k = perm.size(); k < off; ++k) perm.push_back((int32_t)k);
    e->L.push_back(l);
    ci = l.co; h = l.hp; w = l.wp;
  }
  e->fuse1 = block1_supported(e->L[0].ci, e->L[0].stride, e->L[0].pool, e->L[0].ho, e->L[0].wo, e->L[0].co);
  if (const char* ba = getenv("MI_BRED_ARG")) e->bred_arg = atoi(ba) != 0;
  e->head_c = ci;
  e->head_hw = h * w;
  e->feat = d->head_mean_pool ? ci : ci * h * w;
  e->off_wl = off; off += (size_t)d->ways * e->feat;
  e->off_bl = off; off += d->ways;
  e->P = off;
  e->PS = align_up(off, 64);
  for (const Layer& l : e->L) { e->zoff[e->nzero] = (unsigned)l.off_b; e->zlen[e->nzero] = (unsigned)l.co; e->nzero++; }
  if (e->PS > e->P) { e->zoff[e->nzero] = (unsigned)e->P; e->zlen[e->nzero] = (unsigned)(e->PS - e->P); e->nzero++; }
  for (int wy = 0; wy < d->ways; ++wy)
    for (int i = 0; i < e->feat; ++i) {
      size_t ref;
      if (d->head_mean_pool) ref = (size_t)wy * e->feat + i;
      else { const int s = i / e->head_c, c = i % e->head_c; ref = (size_t)wy * e->feat + (size_t)c * e->head_hw + s; }
      perm.push_back((int32_t)(e->off_wl + ref));
    }
  for (size_t k = perm.size(); k < off; ++k) perm.push_back((int32_t)k);
  // the caller's current device is left as it was (torch tracks it on its own)
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (hipSetDevice(device) != hipSuccess) { delete e; return fail(nullptr, MI_ERR_HIP, "hipSetDevice failed"); }
  const bool ok = hipMalloc(&e->perm_dev, perm.size() * sizeof(int32_t)) == hipSuccess &&
                  hipMemcpy(e->perm_dev, perm.data(), perm.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMalloc(&e->counters, mi_engine::kMaxCounterTasks * sizeof(unsigned)) == hipSuccess &&
                  hipMemset(e->counters, 0, mi_engine::kMaxCounterTasks * sizeof(unsigned)) == hipSuccess;
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  if (!ok) {
    if (e->perm_dev) (void)hipFree(e->perm_dev);
    if (e->counters) (void)hipFree(e->counters);
    delete e;
    return fail(nullptr, MI_ERR_HIP, "allocating the parameter permutation table / arrival counters failed");
  }
  *out = e;
  return MI_OK;
}

extern "C" void mi_engine_destroy(mi_engine* e) {
  if (!e) return;
  if (e->perm_dev) (void)hipFree(e->perm_dev);
  if (e->counters) (void)hipFree(e->counters);
  for (auto& g : e->graphs) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
  }
  for (auto& c : e->sc) {
    if (c.fork) (void)hipEventDestroy(c.fork);
    if (c.join) (void)hipEventDestroy(c.join);
  }
  for (auto ev : e->ev0) (void)hipEventDestroy(ev);
  for (auto ev : e->ev1) (void)hipEventDestroy(ev);
  delete e;
}

// Ablation / test switch: run block 1 through the generic conv + BN kernels (z stored) instead of the conv-recompute kernels.
extern "C" int mi_engine_set_fused_block1(mi_engine* e, int on) {
  if (!e) return MI_ERR_ARG;
  e->fuse1 = on && block1_supported(e->L[0].ci, e->L[0].stride, e->L[0].pool, e->L[0].ho, e->L[0].wo, e->L[0].co);
  e->gram1 = on != 2;      // 2 = fused kernels, statistics by conv-recompute passes (no Gram matrix)
  e->gramq = on != 3;      // 3 = the Gram matrix for the support passes only (the query pass by conv-recompute kernels: rounds 2 - 5)
  return MI_OK;
}

// Ablation / test switch: 1 (default) = BatchNorm partials folded by the last workgroup of the producing kernel, 0 = separate
// bn_finalize launches.  Results are bit-identical either way (same fold order).
extern "C" int mi_engine_set_fused_finalize(mi_engine* e, int on) {
  if (!e) return MI_ERR_ARG;
  e->fuse_fin = on != 0;
  return MI_OK;
}
// Ablation / test switch: 1 (default) = block 1's BatchNorm-backward sums computed in the epilogue of block 2's dgrad kernel,
// 0 = by a separate streaming pass (pooled_reduce).  Same fp64 sums in a different order: results agree to fp32 rounding.
extern "C" int mi_engine_set_fused_block1_reduce(mi_engine* e, int on) {
  if (!e) return MI_ERR_ARG;
  e->fuse_b1red = on != 0;
  return MI_OK;
}

// 1 = replay repeated identical calls of mi_meta_batch_maml / mi_meta_batch_anil as a captured hipGraph (default 0).  Identical
// means identical ARGUMENTS: the caller must pass the same device buffers (parameters, data, outputs, workspace) again; results
// are those of the eager call (same kernels, same order).  Profiling and the debug trace switch replay off for their calls.
extern "C" int mi_engine_set_graph(mi_engine* e, int on) {
  if (!e) return MI_ERR_ARG;
  e->graph_on = on != 0;
  return MI_OK;
}

// Ablation / test switch: 1 (default) = one "advance" launch ends every pass of mi_meta_batch_maml (weight-gradient folds, block 1's
// Gram assembly, the update, the next pass's Gram statistics); 0 = the separate launches.  Bit-identical results.
extern "C" int mi_engine_set_fused_tail(mi_engine* e, int on) {
  if (!e) return MI_ERR_ARG;
  e->fuse_tail = on != 0;
  return MI_OK;
}

// Ablation / test switch: 1 (default) = the last block's BatchNorm + pooling, the head, its backward and that block's BatchNorm-backward sums (or
// their tangents) in one launch, four workgroups per task (tail.hip); 0 = the four separate launches per pass.  Same arithmetic in the same
// order: p, logits, loss, accuracy, the head's gradients and df are bit-identical; the BatchNorm-backward sums are the same fp64 terms folded
// in a different (fixed) order.
extern "C" int mi_engine_set_fused_last_block(mi_engine* e, int on) {
  if (!e) return MI_ERR_ARG;
  e->fuse_last = on != 0;
  return MI_OK;
}

// 0 = every kernel on the caller's stream; 1 (default) = weight gradients of blocks >= 2 on an engine-owned side stream,
// forked after dz_l is written and joined before the call returns control of the gradients to the caller's stream.
extern "C" int mi_engine_set_overlap(mi_engine* e, int on) {
  if (!e) return MI_ERR_ARG;
  e->overlap = on != 0;
  e->fork_once = on == 3 || fork_once_env();
  return MI_OK;
}

// Engine-owned streams: one pool per device for the life of the process, shared by every engine on it (an engine is driven by
// one host thread and fork / join order every use; creating a stream is slow on this stack while engines are created freely).
static hipStream_t pool_stream(const mi_engine* e, int which) {
  static hipStream_t g_pool[64][1] = {};
  const int d = (e->device >= 0 && e->device < 64) ? e->device : 0;
  if (!g_pool[d][which]) {          // created on the ENGINE's device, whatever device is current in the calling thread
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != e->device) (void)hipSetDevice(e->device);
    if (hipStreamCreateWithFlags(&g_pool[d][which], hipStreamNonBlocking) != hipSuccess) g_pool[d][which] = nullptr;
    if (prev >= 0 && prev != e->device) (void)hipSetDevice(prev);
  }
  return g_pool[d][which];
}
// (hipEventReleaseToDevice instead of the default system-scope release was measured in round 5: no difference -- the ~7 us the caller's stream
// idles behind an event record are the marker packet itself, not its cache write-back; profiles/r5/overlap_fork_ab.txt)
static bool make_event(hipEvent_t* ev) { return *ev || hipEventCreateWithFlags(ev, hipEventDisableTiming) == hipSuccess; }

// fork = the side stream waits for everything issued on `st` so far; join = `st` waits for everything issued on the side stream
static hipStream_t side_fork(mi_engine* e, hipStream_t st, int half) {
  if (!e->overlap) return st;
  mi_engine::SideCtx& c = e->sc[half];
  if (!c.side) {
    c.side = pool_stream(e, 0);
    if (!c.side || !make_event(&c.fork) || !make_event(&c.join)) { e->overlap = false; return st; }
  }
  // (stream memory operations instead of the event -- hipStreamWriteValue32 on the caller's stream, hipStreamWaitValue32 on the side stream --
  // were tried in round 5 as a cheaper fork: they fail on this stack with "invalid argument")
  if (hipEventRecord(c.fork, st) != hipSuccess || hipStreamWaitEvent(c.side, c.fork, 0) != hipSuccess) return st;
  return c.side;
}
static int side_join(mi_engine* e, hipStream_t st, int half, bool used) {
  if (!used) return MI_OK;
  mi_engine::SideCtx& c = e->sc[half];
  if (hipEventRecord(c.join, c.side) != hipSuccess || hipStreamWaitEvent(st, c.join, 0) != hipSuccess)
    return fail(e, MI_ERR_HIP, "side-stream join failed");
  return MI_OK;
}

// Debug/test aid: while set, every second-order mi_meta_batch_maml call with with_grad != 0 also writes, per task and in the
// reference's parameter order, theta_k (k = 0..K), g_k = grad L_support(theta_k), the vector lam_{k+1} fed to the k-th
// Hessian-vector product and H_support(theta_k) lam_{k+1} into `buf`:  [K+1][T][P] | [K][T][P] | [K][T][P] | [K][T][P] floats.
// While set, every forward pass of mi_meta_batch_maml (support steps 0..K-1, then the query pass) / mi_meta_batch_anil (the one trunk
// pass) also writes the BatchNorm batch statistics of every block: buf [passes][tasks][2][C_total] floats (C_total = sum of the blocks'
// filters, block-major): [0] = batch mean, [1] = biased batch variance (1/rstd^2 - eps).  The reference's BatchNorm2d layers update
// their running_mean / running_var buffers from exactly these on every learner(x) (torch momentum 0.1, unbiased variance; learn2learn's
// clone shares the buffers with the base model), and utils/experiment.py:85-90 saves them: the host side folds them into the
// model's buffers in the reference's call order (core_functions/vision_models.py::fold_running_stats).
extern "C" int mi_engine_set_bn_export(mi_engine* e, float* buf, size_t floats) {
  if (!e) return MI_ERR_ARG;
  e->bn_export = buf;
  e->bn_export_floats = buf ? floats : 0;
  return MI_OK;
}

extern "C" int mi_debug_set_trace(mi_engine* e, float* buf, size_t floats) {
  if (!e) return MI_ERR_ARG;
  e->trace = buf;
  e->trace_floats = buf ? floats : 0;
  return MI_OK;
}

int mi_internal_fail(int code, const char* msg) { return fail(nullptr, code, msg); }

extern "C" int mi_param_count(const mi_engine* e, size_t* n) {
  if (!e || !n) return MI_ERR_ARG;
  *n = e->P;
  return MI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
struct Bump {
  char* base;
  size_t off;
  template <class T> T* take(size_t count) {
    off = align_up(off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

struct ActSet {
  float *z[8], *p[8], *dz[8], *dp[8], *mu[8], *rstd[8];
  float* zhl[8];  // blocks >= 2 whose BatchNorm-backward sums ride in the next block's dgrad epilogue: zhat at the argmax of every
                  // pooled output (same shape as p[l]), else nullptr
  float* zhm;     // fused block 1 with backward: zhat at every pooling window's argmax (same shape as p[0]), else nullptr
  uint8_t* arg0;  // ... and the argmax position itself (4 = did not pass the ReLU)
  float *f, *df, *prob, *dl;
};
struct TanSet {
  float *zd[8], *pd[8], *m1[8], *m2[8];
  float* zhdl[8]; // tangent of ActSet::zhl
  float* zhdm;    // fused block 1: tangent of zhat at the argmax
  float* rdz[8];   // R{dz} per block (kept per block: the side-stream weight gradient of block l reads it while block l-1 is written)
  float *dpd[2], *fd, *rdf;
};
struct Plan {
  float *theta, *g, *lam, *hv;
  float* lam2;          // fused tail: the adjoint recursion ping-pongs between lam and lam2 (the advance launch must not update in place)
  float *xs, *xq;
  int32_t *ys, *yq;
  std::vector<ActSet> sup;
  ActSet qry;
  TanSet tan;
  double* bnpart;
  float* wgpart;
  float* wgpart_side;   // partials of the weight gradients that run on the side stream
  float* wgpart_l[8];   // fused tail: one partial buffer per block (the folds wait for the pass's advance launch)
  double *gram_part, *gram_s;   // input Gram matrix of the support images (block 1 statistics), or nullptr
  double *gram_q, *gram_part_q;   // ... of the query images (experiment: MI_GRAM_QUERY), or nullptr
  int half = 0;                 // stream context (engine SideCtx) this plan's side work uses
  float *tmp_loss, *tmp_acc;
  float* hscr;   // head scratch: R{dl} [T][n][ways], row loss [T][n], row hit [T][n]
  float* tail_wpart = nullptr; double* tail_bpart = nullptr; float* tail_scr = nullptr;   // one-launch tail (tail.hip): row-group partials
  // fp16 operand form (bf16_split.h): largest-magnitude cells [slot][T], one slot per tensor a convolution reads, handed out in launch
  // order (cell_bind: the tensor's producer is about to run; cell_of: a consumer asks) and zeroed once per call (plan_begin)
  unsigned* cells;
  int cell_cap, cell_used = 0, cell_T = 0;
  bool f16 = false;
  std::vector<std::pair<const void*, unsigned*>> cellmap;
  size_t bytes;
};

static void plan_actset(const mi_engine* e, Bump& b, ActSet& A, int T, int n, bool with_bwd) {
  const int nl = (int)e->L.size();
  A.zhm = nullptr;
  A.arg0 = nullptr;
  for (int l = 0; l < 8; ++l) A.zhl[l] = nullptr;
  for (int l = 0; l < nl; ++l) {
    const Layer& L = e->L[l];
    const size_t zs = (size_t)T * n * L.ho * L.wo * L.co, ps = (size_t)T * n * L.hp * L.wp * L.co;
    const bool fused = (l == 0 && e->fuse1);   // conv output / its gradient are recomputed, never stored
    A.z[l] = fused ? nullptr : b.take<float>(zs);
    A.p[l] = b.take<float>(ps);
    A.dz[l] = (with_bwd && !fused) ? b.take<float>(zs) : nullptr;
    A.dp[l] = with_bwd ? b.take<float>(ps) : nullptr;
    if (fused && with_bwd && e->gram1) { A.zhm = b.take<float>(ps); A.arg0 = b.take<uint8_t>(ps); }
    A.zhl[l] = (!fused && with_bwd && l >= 1 && L.pool && dgrad_carries_reduce(e, l + 1)) ? b.take<float>(ps) : nullptr;
    A.mu[l] = b.take<float>((size_t)T * L.co);
    A.rstd[l] = b.take<float>((size_t)T * L.co);
  }
  if (e->d.head_mean_pool) {
    A.f = b.take<float>((size_t)T * n * e->feat);
    A.df = with_bwd ? b.take<float>((size_t)T * n * e->feat) : nullptr;
  } else {
    A.f = A.p[nl - 1];
    A.df = A.dp[nl - 1];
  }
  A.prob = b.take<float>((size_t)T * n * e->d.ways);
  A.dl = b.take<float>((size_t)T * n * e->d.ways);
}

// Everything that decides WHICH kernels a fused call launches besides the engine's own switches: the operand form of the hidden
// convolutions, its bisecting mask, the kernel of that form (16x16x32 / 32x32x16), block 1's operand form and its sparse weight gradient's.  Part of the hipGraph cache
// key: a graph captured under one selection must not be replayed under another.
static unsigned long long kernel_selection_key() {
  unsigned mask = 0;
  const unsigned form = (unsigned)mi_conv_get_split_bf16(&mask);
  return (unsigned long long)form + 4ull * (unsigned)conv_b16() + 16ull * (unsigned)block1_split_form() + 64ull * (unsigned)sparse_wgrad_split_form() + 128ull * (unsigned long long)mask;
}

static void make_plan(const mi_engine* e, void* ws, int T, int ns, int nq, int K, int second_order, Plan& pl) {
  Bump b{reinterpret_cast<char*>(ws), 0};
  const int nl = (int)e->L.size();
  const size_t TP = (size_t)T * e->PS;
  pl.theta = b.take<float>(TP * (K + 1));
  pl.g = b.take<float>(TP * (K > 0 ? K : 1));
  pl.lam = b.take<float>(TP);
  pl.hv = b.take<float>(TP);
  pl.lam2 = (second_order && K > 0) ? b.take<float>(TP) : nullptr;
  const size_t img = (size_t)e->d.in_h * e->d.in_w * e->d.in_channels;
  pl.xs = b.take<float>((size_t)T * ns * img);
  pl.xq = b.take<float>((size_t)T * nq * img);
  pl.ys = b.take<int32_t>((size_t)T * ns);
  pl.yq = b.take<int32_t>((size_t)T * nq);
  pl.tmp_loss = b.take<float>(T);
  pl.tmp_acc = b.take<float>(T);
  pl.hscr = b.take<float>((size_t)T * (ns > nq ? ns : nq) * (e->d.ways + 2));
  {
    const int nmx = ns > nq ? ns : nq;
    pl.tail_wpart = b.take<float>(tail_wpart_floats(T, e->feat, e->d.ways));
    pl.tail_bpart = b.take<double>(tail_bpart_doubles(T, e->L.back().co));
    pl.tail_scr = b.take<float>(tail_scr_floats(T, nmx, e->d.ways));
  }
  const int nsets = (second_order && K > 0) ? K : 1;
  pl.sup.resize(nsets);
  for (int k = 0; k < nsets; ++k) plan_actset(e, b, pl.sup[k], T, ns, true);
  plan_actset(e, b, pl.qry, T, nq, true);
  // partial buffers (sized for the larger of support / query passes)
  const int nmax = ns > nq ? ns : nq;
  size_t bnp = 0, wgp = 0, zmax = 0, pmax = 0;
  for (int l = 0; l < nl; ++l) {
    const Layer& L = e->L[l];
    const ConvGeom gg = geom(L, nmax);
    int blk = conv_max_blocks_per_task(gg);
    const int bb = bn_blocks_per_task(nmax, L.ho, L.wo, L.co, L.pool, 1);
    if (bb > blk) blk = bb;
    const size_t need = (size_t)T * blk * 2 * L.co;
    if (need > bnp) bnp = need;
    size_t w = wgrad_partial_floats(gg, T);
    if (l == 0 && e->fuse1) {
      int bpt = block1_blocks_per_task(nmax, L.ho, L.wo, L.co, T);
      const int sb = sparse_wgrad_blocks_per_task(nmax, L.ho, L.wo, L.co, T);
      if (sb > bpt) bpt = sb;
      w = (size_t)T * bpt * 9 * L.ci * L.co;
    }
    if (w > wgp) wgp = w;
    const size_t zs = (l == 0 && e->fuse1) ? 0 : (size_t)T * nmax * L.ho * L.wo * L.co, ps = (size_t)T * nmax * L.hp * L.wp * L.co;
    if (zs > zmax) zmax = zs;
    if (ps > pmax) pmax = ps;
  }
  pl.bnpart = b.take<double>(bnp);
  pl.wgpart = b.take<float>(wgp);
  pl.wgpart_side = b.take<float>(wgp);
  for (int l = 0; l < 8; ++l) pl.wgpart_l[l] = nullptr;
  for (int l = 0; l < nl; ++l) {
    if (l == 0 && e->fuse1) { pl.wgpart_l[0] = pl.wgpart; continue; }     // block 1's sparse partials: consumed inside the same advance launch
    pl.wgpart_l[l] = b.take<float>(wgrad_partial_floats(geom(e->L[l], nmax), T));
  }
  pl.gram_part = pl.gram_s = pl.gram_q = pl.gram_part_q = nullptr;
  if (e->fuse1 && e->gram1 && gram_supported(e->L[0].w, e->L[0].ci) && (K >= 2 || (K >= 1 && second_order))) {   // the support set is swept at least twice
    pl.gram_part = b.take<double>(gram_partial_doubles(T, ns, e->L[0].h, e->L[0].ci));
    pl.gram_s = b.take<double>(gram_doubles(T, e->L[0].ci));
    if (e->gramq && gram_query_env() && nq == ns) {
      pl.gram_q = b.take<double>(gram_doubles(T, e->L[0].ci));
      pl.gram_part_q = b.take<double>(gram_partial_doubles(T, nq, e->L[0].h, e->L[0].ci));
    }
  }
  if (second_order && K > 0) {
    TanSet& X = pl.tan;
    for (int l = 0; l < nl; ++l) {
      const Layer& L = e->L[l];
      X.zd[l] = (l == 0 && e->fuse1) ? nullptr : b.take<float>((size_t)T * ns * L.ho * L.wo * L.co);
      X.pd[l] = b.take<float>((size_t)T * ns * L.hp * L.wp * L.co);
      if (l == 0) X.zhdm = e->fuse1 ? b.take<float>((size_t)T * ns * L.hp * L.wp * L.co) : nullptr;
      X.zhdl[l] = (!(l == 0 && e->fuse1) && l >= 1 && L.pool && dgrad_carries_reduce(e, l + 1)) ? b.take<float>((size_t)T * ns * L.hp * L.wp * L.co) : nullptr;
      X.m1[l] = b.take<float>((size_t)T * L.co);
      X.m2[l] = b.take<float>((size_t)T * L.co);
    }
    for (int l = 0; l < nl; ++l) {
      const Layer& L = e->L[l];
      X.rdz[l] = (l == 0 && e->fuse1) ? nullptr : b.take<float>((size_t)T * ns * L.ho * L.wo * L.co);
    }
    X.dpd[0] = b.take<float>(pmax);
    X.dpd[1] = b.take<float>(pmax);
    if (e->d.head_mean_pool) {
      X.fd = b.take<float>((size_t)T * ns * e->feat);
      X.rdf = b.take<float>((size_t)T * ns * e->feat);
    } else {
      X.fd = nullptr;
      X.rdf = nullptr;
    }
  }
  pl.cell_cap = 8 + (K + 1) * 2 * nl + K * 4 * nl + 4 * nl;      // p and dz per block and pass, pd and R{dz} per Hessian-vector pass, spare
  // (reserved only while the fp16 operand form is selected -- 4 KB per tensor and task, ~20 MB at cfg2; the workspace size is recomputed by
  // every call with the form then in force, so a caller that switches forms is asked for the larger workspace by the size check)
  pl.cells = conv_operand_form() == 2 ? b.take<unsigned>((size_t)pl.cell_cap * T * MI_CELL_WORDS) : nullptr;
  pl.cell_T = T;
  pl.bytes = align_up(b.off, 256);
}

// ---- fp16 operand form: the largest-magnitude cell of every tensor a convolution reads
static int plan_begin(mi_engine* e, hipStream_t st, Plan& pl) {
  pl.f16 = conv_operand_form() == 2;
  pl.cell_used = 0;
  pl.cellmap.clear();
  if (pl.f16) HIPCHK(e, hipMemsetAsync(pl.cells, 0, (size_t)pl.cell_cap * pl.cell_T * MI_CELL_WORDS * sizeof(unsigned), st));
  return MI_OK;
}
// the producer of `tensor` is about to be launched: a fresh cell for it to fold max |tensor| into (nullptr: no fp16 form, or no slot
// left -- its consumers then ask launch_amax, or fail there)
static unsigned* cell_bind(mi_engine* e, Plan& pl, const void* tensor) {
  static const bool no_hooks = getenv("MI_F16_NO_PRODUCER_AMAX") && atoi(getenv("MI_F16_NO_PRODUCER_AMAX")) != 0;   // debug: every cell from launch_amax
  if (!pl.f16 || !tensor) return nullptr;
  for (auto it = pl.cellmap.begin(); it != pl.cellmap.end(); ++it)
    if (it->first == tensor) { pl.cellmap.erase(it); break; }      // the tensor is being rewritten: its old cell is stale
  if (no_hooks || pl.cell_used >= pl.cell_cap) return nullptr;
  unsigned* c = pl.cells + (size_t)(pl.cell_used++) * pl.cell_T * MI_CELL_WORDS;
  pl.cellmap.emplace_back(tensor, c);
  return c;
}
// a convolution is about to read `tensor` ([T][per_task] floats): its cell -- from its producer, else a reduction launch here
static int cell_of(mi_engine* e, hipStream_t st, Plan& pl, const float* tensor, size_t per_task, int T, const unsigned** out) {
  *out = nullptr;
  if (!pl.f16 || !tensor) return MI_OK;
  for (const auto& kv : pl.cellmap)
    if (kv.first == tensor) { *out = kv.second; return MI_OK; }
  if (pl.cell_used >= pl.cell_cap) return fail(e, MI_ERR_WORKSPACE, "fp16 operand form: no largest-magnitude cell left for this call");
  unsigned* c = pl.cells + (size_t)(pl.cell_used++) * pl.cell_T * MI_CELL_WORDS;
  pl.cellmap.emplace_back(tensor, c);
  LAUNCH(e, st, OP_MISC, 5, launch_amax(st, tensor, per_task, T, c));
  *out = c;
  return MI_OK;
}

// Debug/test aid: byte offsets inside the workspace of a mi_meta_batch_maml call with these sizes.
// out = {theta, g, xs, sup[0].p[0], sup[0].dp[0], sup[0].mu[0], sup[0].rstd[0], sup[0].p[1], qry.p[0], total bytes}
extern "C" int mi_debug_plan_offsets(const mi_engine* e, int tasks, int ways, int shots, int adapt_steps, int second_order,
                                     size_t* out) {
  if (!e || !out) return MI_ERR_ARG;
  Plan pl;
  char* base = reinterpret_cast<char*>(4096);
  make_plan(e, base, tasks, ways * shots, ways * shots, adapt_steps, second_order, pl);
  auto off = [&](const void* p) { return (size_t)(reinterpret_cast<const char*>(p) - base); };
  out[0] = off(pl.theta); out[1] = off(pl.g); out[2] = off(pl.xs); out[3] = off(pl.sup[0].p[0]); out[4] = off(pl.sup[0].dp[0]);
  out[5] = off(pl.sup[0].mu[0]); out[6] = off(pl.sup[0].rstd[0]); out[7] = off(pl.sup[0].p[1]); out[8] = off(pl.qry.p[0]);
  out[9] = pl.bytes;
  return MI_OK;
}

extern "C" int mi_workspace_bytes(const mi_engine* e, int tasks, int ways, int shots, int adapt_steps, int second_order,
                                  size_t* bytes) {
  if (!e || !bytes || tasks < 1 || ways < 1 || shots < 1 || adapt_steps < 0) return MI_ERR_ARG;
  Plan pl;
  make_plan(e, nullptr, tasks, ways * shots, ways * shots, adapt_steps, second_order, pl);
  *bytes = pl.bytes;
  return MI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
static B1Args b1_args(const mi_engine* e, Plan& pl, ActSet& A, const float* x0, int n, const float* theta) {
  const Layer& L = e->L[0];
  B1Args a{};
  a.x = x0;
  a.w = theta + L.off_w; a.wstride = e->PS;
  a.mu = A.mu[0]; a.rstd = A.rstd[0];
  a.gamma = theta + L.off_gamma; a.beta = theta + L.off_beta; a.pstride = e->PS;
  a.partial = pl.bnpart; a.wpartial = pl.wgpart;
  a.n = n; a.hh = L.ho; a.ww = L.wo; a.co = L.co;
  a.inv_m = 1.f / (float)(n * L.ho * L.wo);
  return a;
}

// (fp16 operand form) the largest-magnitude cell of a tensor a convolution is about to read, on the caller's stream `st`
#define CELL(ptr, per_task, out)                                                   \
  do {                                                                             \
    const int _crc = cell_of(e, st, pl, ptr, per_task, T, &(out));                 \
    if (_crc) return _crc;                                                         \
  } while (0)

// The one-launch tail (tail.hip) serves a generic last block (not the conv-recompute block 1) feeding a flattened head, outside the fp16
// operand form (its producers would have to fold largest-magnitude cells).
static bool fused_last_ok(const mi_engine* e, const Plan& pl, int n, int T) {
  const int nl = (int)e->L.size();
  if (!e->fuse_last || e->d.head_mean_pool || pl.f16 || (nl == 1 && e->fuse1) || !e->counters || T > mi_engine::kMaxCounterTasks || T > 65535 ||
      !pl.tail_wpart)
    return false;
  const Layer& L = e->L[nl - 1];
  return tail_supported(n, L.ho, L.wo, L.co, L.pool, e->feat, e->d.ways);
}
static void tail_common(const mi_engine* e, const Plan& pl, TailArgs& ta) {
  ta.wpart = pl.tail_wpart; ta.bpart = pl.tail_bpart; ta.scr = pl.tail_scr; ta.counter = e->counters; ta.stamps = e->tail_stamps;
}
// Debug aid: while set, every launch of the one-launch tail (tail.hip) writes the 100 MHz wall clock at its stage boundaries, thread 0 of each of
// its workgroups: buf [tasks][4][16] 64-bit words (device memory; the LAST tail launch of a call is what remains).  NULL switches it off.
extern "C" int mi_debug_tail_stamps(mi_engine* e, unsigned long long* buf) {
  if (!e) return MI_ERR_ARG;
  e->tail_stamps = buf;
  return MI_OK;
}

// Trunk forward: ConvBlocks on n images per task (conv + BN-stat epilogue, finalize, BN+ReLU+pool).
// skip_last_bn: the last block's BatchNorm + pooling belongs to the caller's tail launch.
static int trunk_forward(mi_engine* e, hipStream_t st, Plan& pl, ActSet& A, const float* x0, int n, int T, const float* theta,
                         const double* gram = nullptr, bool stats_ready = false, bool skip_last_bn = false) {
  const int nl = (int)e->L.size();
  const size_t P = e->PS;  // task stride
  for (int l = 0; l < nl; ++l) {
    const Layer& L = e->L[l];
    if (l == 0 && e->fuse1) {
      B1Args ba = b1_args(e, pl, A, x0, n, theta);
      int blk = 0;
      if (gram && stats_ready) {   // ... already formed by the advance launch that ended the previous pass
      } else if (gram) {   // mean / variance of conv1's output as quadratic forms of this step's weights (gram.hip)
        LAUNCH(e, st, OP_GRAM_STATS, 0, launch_gram_stats(st, gram, T, L.ci, L.co, theta + L.off_w, P, nullptr, 0, 1.0 / ((double)n * L.ho * L.wo), 0, A.mu[0], A.rstd[0], nullptr, nullptr));
      } else {
        ba.fin = fin_of(e, T, 1.0 / ((double)n * L.ho * L.wo), FIN_STATS, A.mu[0], L.co, A.rstd[0], L.co);
        LAUNCH(e, st, OP_CONV_FWD, 0, launch_block1(st, ba, T, L.ci, B1_STATS, &blk));
        if (!ba.fin.counter)
          LAUNCH(e, st, OP_BN_FINALIZE, 0, launch_bn_finalize(st, pl.bnpart, blk, T, L.co, 1.0 / ((double)n * L.ho * L.wo), FIN_STATS, A.mu[0], L.co, A.rstd[0], L.co));
        ba.fin = FinArgs{};
      }
      ba.out = A.p[0];
      ba.zh_out = A.zhm;
      ba.arg_out = A.arg0;
      ba.amax_out = cell_bind(e, pl, ba.out);
      // Without the Gram path this pass's backward (or tangent) kernels recompute conv1 on the fp32 pipe and re-derive the pooling / ReLU decisions
      // from it: the forward must then round the same way.  With it every later kernel reads the decisions this launch stores.
      ba.fwd_fp32 = (gram && A.arg0) ? 0 : 1;
      LAUNCH(e, st, OP_BN_FWD, 0, launch_block1(st, ba, T, L.ci, B1_FWD, nullptr));
      continue;
    }
    ConvArgs ca{};
    ca.in[0] = l == 0 ? x0 : A.p[l - 1];
    ca.wt[0] = theta + L.off_w;
    ca.wstride = P;
    ca.out = A.z[l];
    ca.partial = pl.bnpart;
    ca.g = geom(L, n);
    ca.mpix = n * L.ho * L.wo;
    int blk = 0;
    ca.fin = fin_of(e, T, 1.0 / (double)ca.mpix, FIN_STATS, A.mu[l], L.co, A.rstd[l], L.co);
    if (l >= 1) CELL(ca.in[0], (size_t)n * L.h * L.w * L.ci, ca.amax[0]);
    LAUNCH(e, st, OP_CONV_FWD, l, launch_conv3x3(st, ca, T, 1, EPI_STATS, 0, &blk));
    if (!ca.fin.counter)
      LAUNCH(e, st, OP_BN_FINALIZE, l, launch_bn_finalize(st, pl.bnpart, blk, T, L.co, 1.0 / (double)ca.mpix, FIN_STATS, A.mu[l], L.co, A.rstd[l], L.co));
    if (skip_last_bn && l == nl - 1) break;
    BnArgs ba{};
    ba.z = A.z[l]; ba.mu = A.mu[l]; ba.rstd = A.rstd[l];
    ba.gamma = theta + L.off_gamma; ba.beta = theta + L.off_beta; ba.pstride = P;
    ba.out = A.p[l];
    ba.zh_out = A.zhl[l];
    ba.n = n; ba.ho = L.ho; ba.wo = L.wo; ba.c = L.co;
    ba.amax_out = cell_bind(e, pl, ba.out);
    LAUNCH(e, st, OP_BN_FWD, l, launch_bn_fwd(st, ba, T, L.pool));
  }
  return MI_OK;
}

// zhat at the argmax of block `lower`'s pooled outputs, if the forward pass kept it (fused block 1: zhm; hidden blocks: zhl)
static const float* zh_at_argmax(const mi_engine* e, const ActSet& A, int lower) {
  return (lower == 0 && e->fuse1) ? A.zhm : A.zhl[lower];
}

// Trunk backward from A.dp[last] (gradient w.r.t. the last block's output): writes gamma/beta/conv-weight gradients into g.
// adv != nullptr (fused tail): weight-gradient folds and block 1's Gram assembly are recorded in *adv for the caller's advance launch
// instead of being launched here; every block then writes its own partial buffer (pl.wgpart_l).
static void adv_add_seg(AdvanceArgs* adv, size_t off, int nelem, const float* partial, int nchunks) {
  AdvanceSeg& sg = adv->seg[adv->nseg++];
  sg.off = (unsigned)off; sg.nelem = (unsigned)nelem; sg.partial = partial; sg.nchunks = nchunks;
}
// last_bn_done: the caller's tail launch already formed the last block's dgamma / dbeta (the apply is launched here).
static int trunk_backward(mi_engine* e, hipStream_t st, Plan& pl, ActSet& A, const float* x0, int n, int T, const float* theta,
                          float* g, const double* gram = nullptr, AdvanceArgs* adv = nullptr, bool last_bn_done = false) {
  const int nl = (int)e->L.size();
  const size_t P = e->PS;
  bool forked = false, red_done[9] = {false, false, false, false, false, false, false, false, false};
  WgradArgs held[9]; int held_l[9], nheld = 0;                  // weight gradients held back for the pass's one fork of the side stream   // [l]: block l's sums rode in block l+1's dgrad
  for (int l = nl - 1; l >= 0; --l) {
    const Layer& L = e->L[l];
    const int mpix = n * L.ho * L.wo;
    const bool b1red_done = red_done[0];
    if (l == 0 && e->fuse1) {
      B1Args b1 = b1_args(e, pl, A, x0, n, theta);
      b1.dp = A.dp[0];
      int blk = 0;
      const FinArgs fin = fin_of(e, T, 1.0, FIN_SUMS, g + L.off_gamma, P, g + L.off_beta, P);
      if (b1red_done) {   // already summed in the epilogue of block 2's dgrad
      } else if (A.zhm) {   // dgamma / dbeta from pooled-resolution tensors (no conv recompute)
        PoolRedArgs pr{A.p[0], A.zhm, nullptr, A.dp[0], nullptr, pl.bnpart, n * L.hp * L.wp, L.co, fin};
        LAUNCH(e, st, OP_BN_BWD_REDUCE, 0, launch_pooled_reduce(st, pr, T, 0, &blk));
      } else {
        b1.fin = fin;
        LAUNCH(e, st, OP_BN_BWD_REDUCE, 0, launch_block1(st, b1, T, L.ci, B1_BWD_REDUCE, &blk));
        b1.fin = FinArgs{};
      }
      if (!fin.counter && !b1red_done)
        LAUNCH(e, st, OP_BN_FINALIZE, 0, launch_bn_finalize(st, pl.bnpart, blk, T, L.co, 1.0, FIN_SUMS, g + L.off_gamma, P, g + L.off_beta, P));
      b1.dgamma = g + L.off_gamma; b1.dbeta = g + L.off_beta; b1.gstride = P;
      if (gram && A.arg0 && sparse_wgrad_supported(L.wo, L.ci, L.co)) {   // sparse part on the matrix pipe, dense parts from the Gram matrix: no conv recompute
        SparseWgArgs sw{};
        sw.x = x0; sw.arg = A.arg0; sw.dp = A.dp[0]; sw.wpartial = pl.wgpart; sw.n = n; sw.hh = L.ho; sw.ww = L.wo; sw.co = L.co;
        LAUNCH(e, st, OP_WGRAD, 0, launch_sparse_wgrad(st, sw, T, L.ci, 0, &blk));
        GramWgArgs gw{};
        gw.g = gram; gw.spartial = pl.wgpart; gw.nblk = blk; gw.w = theta + L.off_w; gw.wstride = P;
        gw.mu = A.mu[0]; gw.rstd = A.rstd[0]; gw.gamma = theta + L.off_gamma; gw.pstride = P;
        gw.dgamma = g + L.off_gamma; gw.dbeta = g + L.off_beta; gw.gstride = P;
        gw.out = g + L.off_w; gw.ostride = P; gw.ci = L.ci; gw.co = L.co; gw.inv_m = 1.0 / (double)mpix;
        if (adv) { adv->b1_wgrad = 1; adv->gw_tangent = 0; adv->gw = gw; }
        else LAUNCH(e, st, OP_WGRAD_REDUCE, 0, launch_gram_wgrad(st, gw, T, 0));
        continue;
      }
      LAUNCH(e, st, OP_WGRAD, 0, launch_block1(st, b1, T, L.ci, B1_BWD_WGRAD, &blk));
      if (adv) adv_add_seg(adv, L.off_w, 9 * L.ci * L.co, pl.wgpart, blk);
      else LAUNCH(e, st, OP_WGRAD_REDUCE, 0, launch_wgrad_reduce(st, pl.wgpart, blk, 9 * L.ci * L.co, T, g + L.off_w, P));
      continue;
    }
    BnArgs ba{};
    ba.z = A.z[l]; ba.mu = A.mu[l]; ba.rstd = A.rstd[l];
    ba.gamma = theta + L.off_gamma; ba.beta = theta + L.off_beta; ba.pstride = P;
    ba.dp = A.dp[l];
    ba.partial = pl.bnpart;
    ba.n = n; ba.ho = L.ho; ba.wo = L.wo; ba.c = L.co;
    ba.inv_m = 1.f / (float)mpix;
    int blk = 0;
    const bool bn_done = last_bn_done && l == nl - 1;
    if (!red_done[l] && !bn_done) {
      ba.fin = fin_of(e, T, 1.0, FIN_SUMS, g + L.off_gamma, P, g + L.off_beta, P);
      LAUNCH(e, st, OP_BN_BWD_REDUCE, l, launch_bn_bwd_reduce(st, ba, T, L.pool, &blk));
      if (!ba.fin.counter)
        LAUNCH(e, st, OP_BN_FINALIZE, l, launch_bn_finalize(st, pl.bnpart, blk, T, L.co, 1.0, FIN_SUMS, g + L.off_gamma, P, g + L.off_beta, P));
    }
    ba.dgamma = g + L.off_gamma; ba.dbeta = g + L.off_beta; ba.gstride = P;
    ba.out = A.dz[l];
    ba.amax_out = cell_bind(e, pl, ba.out);
    LAUNCH(e, st, OP_BN_BWD_APPLY, l, launch_bn_bwd_apply(st, ba, T, L.pool));
    WgradArgs wa{};
    wa.x[0] = l == 0 ? x0 : A.p[l - 1];
    wa.dz[0] = A.dz[l];
    if (l >= 1) {
      CELL(wa.x[0], (size_t)n * L.h * L.w * L.ci, wa.amax_x[0]);
      CELL(wa.dz[0], (size_t)n * L.ho * L.wo * L.co, wa.amax_dz[0]);
    }
    wa.g = geom(L, n);
    wa.mpix = mpix;
    // The weight gradients of the hidden blocks run on the side stream beside the dgrad chain.  One fork per BLOCK is the default
    // (fork_once = false); with one fork per PASS (mi_engine_set_overlap(e, 3); measured slower, profiles/r5/overlap_fork_ab.txt) they are held
    // back until the last hidden block's dz exists and then issued together -- they overlap that block's dgrad and block 1's kernels --
    // because every fork is an event record on the caller's stream, which idles it for ~7 us (tools/launch_floor.py: 25 such gaps per
    // meta-iteration with one fork per block, at every task count; profiles/r5/launch_floor_cfg2_T*.txt).
    auto issue_wgrad = [&](hipStream_t ws, WgradArgs& w, int wl) -> int {
      const Layer& Lw = e->L[wl];
      w.partial = adv ? pl.wgpart_l[wl] : (ws != st ? pl.wgpart_side : pl.wgpart);
      int nch = 0;
      LAUNCH(e, ws, OP_WGRAD, wl, launch_wgrad3x3(ws, w, T, 1, &nch));
      if (adv) adv_add_seg(adv, Lw.off_w, 9 * Lw.ci * Lw.co, w.partial, nch);
      else LAUNCH(e, ws, OP_WGRAD_REDUCE, wl, launch_wgrad_reduce(ws, w.partial, nch, 9 * Lw.ci * Lw.co, T, g + Lw.off_w, P));
      return MI_OK;
    };
    if (l > 0 && e->overlap && e->fork_once) {
      held[nheld] = wa; held_l[nheld++] = l;
      if (l == 1) {
        hipStream_t ws = side_fork(e, st, pl.half);
        if (ws != st) forked = true;
        for (int q = 0; q < nheld; ++q) { const int rc = issue_wgrad(ws, held[q], held_l[q]); if (rc) return rc; }
        nheld = 0;
      }
    } else {
      hipStream_t ws = l > 0 ? side_fork(e, st, pl.half) : st;
      if (ws != st) forked = true;
      const int rc = issue_wgrad(ws, wa, l);
      if (rc) return rc;
    }
    if (l > 0) {
      ConvArgs ca{};
      ca.in[0] = A.dz[l];
      ca.wt[0] = theta + L.off_w;
      ca.wstride = P;
      ca.out = A.dp[l - 1];
      ca.amax[0] = wa.amax_dz[0];
      ca.g = geom_dgrad(L, n);
      ca.mpix = n * L.h * L.w;
      const float* zh_lo = dgrad_carries_reduce(e, l) ? zh_at_argmax(e, A, l - 1) : nullptr;
      if (zh_lo) {   // dgamma / dbeta of the block below in this kernel's epilogue
        const Layer& L0 = e->L[l - 1];
        ca.bp = A.p[l - 1]; ca.bzh = zh_lo; ca.partial = pl.bnpart;
        ca.barg = (l - 1 == 0 && e->fuse1 && e->bred_arg) ? A.arg0 : nullptr;
        ca.fin = fin_of(e, T, 1.0, FIN_SUMS, g + L0.off_gamma, P, g + L0.off_beta, P);
        int blk0 = 0;
        LAUNCH(e, st, OP_DGRAD, l, launch_conv3x3(st, ca, T, 1, EPI_BRED, 1, &blk0));
        if (!ca.fin.counter)
          LAUNCH(e, st, OP_BN_FINALIZE, l - 1, launch_bn_finalize(st, pl.bnpart, blk0, T, L0.co, 1.0, FIN_SUMS, g + L0.off_gamma, P, g + L0.off_beta, P));
        red_done[l - 1] = true;
      } else {
        LAUNCH(e, st, OP_DGRAD, l, launch_conv3x3(st, ca, T, 1, EPI_NONE, 1, nullptr));
      }
    }
  }
  return side_join(e, st, pl.half, forked);
}

static void head_scratch(const mi_engine* e, HeadArgs& ha, float* hscr, int T, int n) {
  ha.rdl = hscr;
  ha.rowloss = hscr + (size_t)T * n * e->d.ways;
  ha.rowhit = ha.rowloss + (size_t)T * n;
}

// Linear + CE on features f [T][n][feat]: loss/acc/logits, prob & dl saved, and (with_grad) dwl/dbl into g, df.
static int head_pass(mi_engine* e, hipStream_t st, float* hscr, const float* f, const int32_t* y, int n, int T, const float* theta,
                     float* g, float* loss, float* acc, float* logits, float* prob, float* dl, float* df, bool with_grad) {
  const size_t P = e->PS;
  HeadArgs ha{};
  ha.f = f;
  ha.wl = theta + e->off_wl; ha.bl = theta + e->off_bl; ha.pstride = P;
  ha.y = y; ha.loss = loss; ha.acc = acc; ha.logits = logits; ha.prob = prob; ha.dl = dl;
  ha.dwl = with_grad ? g + e->off_wl : nullptr; ha.dbl = with_grad ? g + e->off_bl : nullptr; ha.gstride = P;
  ha.df = with_grad ? df : nullptr;
  ha.n = n; ha.feat = e->feat; ha.ways = e->d.ways;
  head_scratch(e, ha, hscr, T, n);
  LAUNCH(e, st, OP_HEAD, 0, launch_head_fwd_bwd(st, ha, T, with_grad ? 1 : 0));
  return MI_OK;
}

// mi_engine_set_bn_export: batch mean / biased variance of every block of the pass just run, into slot e->export_pass
static int export_bn_stats(mi_engine* e, hipStream_t st, const ActSet& A, int T) {
  if (!e->bn_export) return MI_OK;
  const int nl = (int)e->L.size();
  int ctot = 0;
  for (const Layer& L : e->L) ctot += L.co;
  const size_t need = (size_t)(e->export_pass + 1) * T * 2 * ctot;
  if (need > e->bn_export_floats) return fail(e, MI_ERR_WORKSPACE, "BatchNorm export buffer too small: need " + std::to_string(need) + " floats");
  BnExportArgs ea{};
  ea.nl = nl; ea.ctot = ctot;
  int off = 0;
  for (int l = 0; l < nl; ++l) { ea.mu[l] = A.mu[l]; ea.rstd[l] = A.rstd[l]; ea.c[l] = e->L[l].co; ea.off[l] = off; off += e->L[l].co; }
  ea.out = e->bn_export + (size_t)e->export_pass * T * 2 * ctot;
  HIPCHK(e, launch_bn_export(st, ea, T));
  e->export_pass += 1;
  return MI_OK;
}

// The BatchNorm export covers the fused calls only (mi_meta_batch_maml / mi_meta_batch_anil number their passes from 0): the
// step-wise entry points run with the export paused, so a learner(x) inside an export window neither writes a stale slot nor
// fails on the buffer size.
struct ExportPause {
  mi_engine* e; float* buf;
  explicit ExportPause(mi_engine* e_) : e(e_), buf(e_ ? e_->bn_export : nullptr) { if (e) e->bn_export = nullptr; }
  ~ExportPause() { if (e) e->bn_export = buf; }
};

// One forward (+ backward) pass of the whole net on n images per task.
// Tb: tasks that take the backward half (the first Tb of the T; -1 = all): the validation tasks of a fused train + validation call
// run the forward half only.  adv: fused tail (trunk_backward); the caller's advance launch then also zeroes what no kernel writes.
static int pass_fwd_bwd(mi_engine* e, hipStream_t st, Plan& pl, ActSet& A, const float* x0, const int32_t* y, int n, int T,
                        const float* theta, float* g, float* loss, float* acc, float* logits, bool with_grad,
                        const double* gram = nullptr, AdvanceArgs* adv = nullptr, bool stats_ready = false, int Tb = -1) {
  const int nl = (int)e->L.size();
  if (Tb < 0) Tb = T;
  const bool fused_last = fused_last_ok(e, pl, n, T);
  int rc = trunk_forward(e, st, pl, A, x0, n, T, theta, gram, stats_ready, fused_last);
  if (rc) return rc;
  rc = export_bn_stats(e, st, A, T);
  if (rc) return rc;
  if (e->d.head_mean_pool) HIPCHK(e, launch_spatial_mean(st, A.p[nl - 1], A.f, T * n, e->head_hw, e->head_c));
  if (with_grad && !adv) HIPCHK(e, hipMemsetAsync(g, 0, (size_t)T * e->PS * sizeof(float), st));
  if (fused_last) {   // the last block's BatchNorm + pooling, the head, its backward and that block's BatchNorm backward: one launch (tail.hip)
    const Layer& L = e->L[nl - 1];
    const size_t P = e->PS;
    TailArgs ta{};
    BnArgs& ba = ta.bn;
    ba.z = A.z[nl - 1]; ba.mu = A.mu[nl - 1]; ba.rstd = A.rstd[nl - 1];
    ba.gamma = theta + L.off_gamma; ba.beta = theta + L.off_beta; ba.pstride = P;
    ba.dp = A.dp[nl - 1]; ba.out = A.dz[nl - 1];
    ba.n = n; ba.ho = L.ho; ba.wo = L.wo; ba.c = L.co;
    ba.inv_m = 1.f / (float)(n * L.ho * L.wo);
    HeadArgs& ha = ta.hd;
    ha.f = A.f;
    ha.wl = theta + e->off_wl; ha.bl = theta + e->off_bl; ha.pstride = P;
    ha.y = y; ha.loss = loss; ha.acc = acc; ha.logits = logits; ha.prob = A.prob; ha.dl = A.dl;
    ha.dwl = with_grad ? g + e->off_wl : nullptr; ha.dbl = with_grad ? g + e->off_bl : nullptr; ha.gstride = P;
    ha.df = with_grad ? A.df : nullptr;
    ha.n = n; ha.feat = e->feat; ha.ways = e->d.ways;
    head_scratch(e, ha, pl.hscr, T, n);
    ta.pooled = A.p[nl - 1];
    ta.sum0 = with_grad ? g + L.off_gamma : nullptr; ta.sum1 = with_grad ? g + L.off_beta : nullptr; ta.sum_stride = P;
    ta.with_grad = with_grad ? 1 : 0; ta.bwd_tasks = Tb;
    tail_common(e, pl, ta);
    LAUNCH(e, st, OP_HEAD, 0, launch_tail(st, ta, T, L.pool, 0));
    if (!with_grad || Tb == 0) return MI_OK;
    return trunk_backward(e, st, pl, A, x0, n, Tb, theta, g, gram, adv, true);
  }
  rc = head_pass(e, st, pl.hscr, A.f, y, n, T, theta, g, loss, acc, logits, A.prob, A.dl, A.df, with_grad);
  if (rc || !with_grad || Tb == 0) return rc;
  if (e->d.head_mean_pool) HIPCHK(e, launch_spatial_mean_bwd(st, A.df, A.dp[nl - 1], Tb * n, e->head_hw, e->head_c));
  return trunk_backward(e, st, pl, A, x0, n, Tb, theta, g, gram, adv);
}

// hv = H(theta) v for the saved support pass A (activations) / g (its gradient): forward-over-reverse.
// dl_fixed != nullptr: the loss is sum(logits * dl_fixed) with a GIVEN cotangent (the step-wise learner's double backward): no
// cross-entropy curvature at the head, and the logit tangents J v go to ld_out.
static int pass_hvp(mi_engine* e, hipStream_t st, Plan& pl, ActSet& A, const float* x0, int n, int T, const float* theta,
                    const float* g, const float* v, float* hv, const double* gram = nullptr, const float* dl_fixed = nullptr,
                    float* ld_out = nullptr, AdvanceArgs* adv = nullptr, bool stats_ready = false) {
  const int nl = (int)e->L.size();
  const size_t P = e->PS;  // task stride
  TanSet& X = pl.tan;
  const bool fused_last = !dl_fixed && !ld_out && fused_last_ok(e, pl, n, T);   // tangent tail in one launch (tail.hip)
  if (!adv) HIPCHK(e, hipMemsetAsync(hv, 0, (size_t)T * P * sizeof(float), st));
  for (int l = 0; l < nl; ++l) {
    const Layer& L = e->L[l];
    const int mpix = n * L.ho * L.wo;
    if (l == 0 && e->fuse1) {
      B1Args b1 = b1_args(e, pl, A, x0, n, theta);
      b1.wd = v + L.off_w; b1.vstride = P;
      int blk = 0;
      if (gram && stats_ready) {   // ... already formed by the advance launch that ended the previous pass
      } else if (gram) {
        LAUNCH(e, st, OP_GRAM_STATS, 0, launch_gram_stats(st, gram, T, L.ci, L.co, theta + L.off_w, P, v + L.off_w, P, 1.0 / (double)mpix, 1, X.m1[0], X.m2[0], A.mu[0], A.rstd[0]));
      } else {
        b1.fin = fin_of(e, T, 1.0 / (double)mpix, FIN_TSTATS, X.m1[0], L.co, X.m2[0], L.co);
        LAUNCH(e, st, OP_TAN_CONV, 0, launch_block1(st, b1, T, L.ci, B1_TSTATS, &blk));
        if (!b1.fin.counter)
          LAUNCH(e, st, OP_BN_FINALIZE, 0, launch_bn_finalize(st, pl.bnpart, blk, T, L.co, 1.0 / (double)mpix, FIN_TSTATS, X.m1[0], L.co, X.m2[0], L.co));
        b1.fin = FinArgs{};
      }
      b1.m1 = X.m1[0]; b1.m2 = X.m2[0];
      b1.gammad = v + L.off_gamma; b1.betad = v + L.off_beta;
      b1.out = X.pd[0];
      b1.zh_out = A.zhm ? X.zhdm : nullptr;
      if (A.zhm && A.arg0) {   // the forward pass kept zhat and the argmax: one conv with the direction's weights suffices
        b1.arg_in = A.arg0; b1.zh_in = A.zhm;
        b1.amax_out = cell_bind(e, pl, b1.out);
        LAUNCH(e, st, OP_BN_TAN_FWD, 0, launch_block1(st, b1, T, L.ci, B1_TFWD_ARG, nullptr));
      } else {
        b1.amax_out = cell_bind(e, pl, b1.out);
        LAUNCH(e, st, OP_BN_TAN_FWD, 0, launch_block1(st, b1, T, L.ci, B1_TFWD, nullptr));
      }
      continue;
    }
    ConvArgs ca{};
    ca.in[0] = l == 0 ? x0 : A.p[l - 1];
    ca.wt[0] = v + L.off_w;
    if (l > 0) { ca.in[1] = X.pd[l - 1]; ca.wt[1] = theta + L.off_w; }
    ca.wstride = P;
    ca.out = X.zd[l];
    ca.z = A.z[l]; ca.mu = A.mu[l]; ca.rstd = A.rstd[l];
    ca.partial = pl.bnpart;
    ca.g = geom(L, n);
    ca.mpix = mpix;
    int blk = 0;
    ca.fin = fin_of(e, T, 1.0 / (double)mpix, FIN_TSTATS, X.m1[l], L.co, X.m2[l], L.co);
    if (l >= 1) {
      CELL(ca.in[0], (size_t)n * L.h * L.w * L.ci, ca.amax[0]);
      CELL(ca.in[1], (size_t)n * L.h * L.w * L.ci, ca.amax[1]);
    }
    LAUNCH(e, st, OP_TAN_CONV, l, launch_conv3x3(st, ca, T, l > 0 ? 2 : 1, EPI_TSTATS, 0, &blk));
    if (!ca.fin.counter)
      LAUNCH(e, st, OP_BN_FINALIZE, l, launch_bn_finalize(st, pl.bnpart, blk, T, L.co, 1.0 / (double)mpix, FIN_TSTATS, X.m1[l], L.co, X.m2[l], L.co));
    if (fused_last && l == nl - 1) break;
    BnArgs ba{};
    ba.z = A.z[l]; ba.zd = X.zd[l]; ba.mu = A.mu[l]; ba.rstd = A.rstd[l]; ba.m1 = X.m1[l]; ba.m2 = X.m2[l];
    ba.gamma = theta + L.off_gamma; ba.beta = theta + L.off_beta; ba.pstride = P;
    ba.gammad = v + L.off_gamma; ba.betad = v + L.off_beta; ba.vstride = P;
    ba.out = X.pd[l];
    ba.zh_out = A.zhl[l] ? X.zhdl[l] : nullptr;
    ba.n = n; ba.ho = L.ho; ba.wo = L.wo; ba.c = L.co;
    ba.amax_out = cell_bind(e, pl, ba.out);
    LAUNCH(e, st, OP_BN_TAN_FWD, l, launch_bn_tan_fwd(st, ba, T, L.pool));
  }
  const float* fd = X.pd[nl - 1];
  if (e->d.head_mean_pool) {
    HIPCHK(e, launch_spatial_mean(st, X.pd[nl - 1], X.fd, T * n, e->head_hw, e->head_c));
    fd = X.fd;
  }
  int cur = 0;
  bool forked = false, red_done[9] = {false, false, false, false, false, false, false, false, false};
  WgradArgs held[9]; int held_l[9], nheld = 0;                  // weight gradients held back for the pass's one fork of the side stream
  HeadArgs ha{};
  ha.f = A.f; ha.fd = fd;
  ha.wl = theta + e->off_wl; ha.bl = theta + e->off_bl; ha.pstride = P;
  ha.wld = v + e->off_wl; ha.bld = v + e->off_bl; ha.vstride = P;
  ha.prob = A.prob; ha.dl = A.dl;
  if (dl_fixed) { ha.dl = const_cast<float*>(dl_fixed); ha.fixed_dl = 1; ha.ld_out = ld_out; }
  ha.dwl = hv + e->off_wl; ha.dbl = hv + e->off_bl; ha.gstride = P;
  ha.df = e->d.head_mean_pool ? X.rdf : X.dpd[cur];
  ha.n = n; ha.feat = e->feat; ha.ways = e->d.ways;
  head_scratch(e, ha, pl.hscr, T, n);
  if (fused_last) {   // tangent of the last block's BatchNorm + pooling, the head's tangent and the tangent of that block's BatchNorm backward
    const Layer& L = e->L[nl - 1];
    TailArgs ta{};
    BnArgs& ba = ta.bn;
    ba.z = A.z[nl - 1]; ba.zd = X.zd[nl - 1]; ba.mu = A.mu[nl - 1]; ba.rstd = A.rstd[nl - 1]; ba.m1 = X.m1[nl - 1]; ba.m2 = X.m2[nl - 1];
    ba.gamma = theta + L.off_gamma; ba.beta = theta + L.off_beta; ba.pstride = P;
    ba.gammad = v + L.off_gamma; ba.betad = v + L.off_beta; ba.vstride = P;
    ba.dgamma = g + L.off_gamma; ba.dbeta = g + L.off_beta; ba.gstride = P;
    ba.dp = A.dp[nl - 1]; ba.dpd = X.dpd[cur]; ba.out = X.rdz[nl - 1];
    ba.n = n; ba.ho = L.ho; ba.wo = L.wo; ba.c = L.co;
    ba.inv_m = 1.f / (float)(n * L.ho * L.wo);
    ta.hd = ha;
    ta.pooled = X.pd[nl - 1];
    ta.sum0 = hv + L.off_gamma; ta.sum1 = hv + L.off_beta; ta.sum_stride = P;
    ta.with_grad = 1; ta.bwd_tasks = T;
    tail_common(e, pl, ta);
    LAUNCH(e, st, OP_HEAD_TAN, 0, launch_tail(st, ta, T, L.pool, 1));
  } else {
    LAUNCH(e, st, OP_HEAD_TAN, 0, launch_head_tangent(st, ha, T));
  }
  if (e->d.head_mean_pool) HIPCHK(e, launch_spatial_mean_bwd(st, X.rdf, X.dpd[cur], T * n, e->head_hw, e->head_c));
  for (int l = nl - 1; l >= 0; --l) {
    const Layer& L = e->L[l];
    const int mpix = n * L.ho * L.wo;
    const bool b1red_done = red_done[0];
    if (l == 0 && e->fuse1) {
      B1Args b1 = b1_args(e, pl, A, x0, n, theta);
      b1.wd = v + L.off_w; b1.vstride = P;
      b1.m1 = X.m1[0]; b1.m2 = X.m2[0];
      b1.gammad = v + L.off_gamma; b1.betad = v + L.off_beta;
      b1.dgamma = g + L.off_gamma; b1.dbeta = g + L.off_beta; b1.gstride = P;
      b1.dp = A.dp[0]; b1.dpd = X.dpd[cur];
      int blk = 0;
      const FinArgs fin = fin_of(e, T, 1.0, FIN_SUMS, hv + L.off_gamma, P, hv + L.off_beta, P);
      if (b1red_done) {
      } else if (A.zhm) {
        PoolRedArgs pr{A.p[0], A.zhm, X.zhdm, A.dp[0], X.dpd[cur], pl.bnpart, n * L.hp * L.wp, L.co, fin};
        LAUNCH(e, st, OP_BN_TAN_BWD_REDUCE, 0, launch_pooled_reduce(st, pr, T, 1, &blk));
      } else {
        b1.fin = fin;
        LAUNCH(e, st, OP_BN_TAN_BWD_REDUCE, 0, launch_block1(st, b1, T, L.ci, B1_TBWD_REDUCE, &blk));
        b1.fin = FinArgs{};
      }
      if (!fin.counter && !b1red_done)
        LAUNCH(e, st, OP_BN_FINALIZE, 0, launch_bn_finalize(st, pl.bnpart, blk, T, L.co, 1.0, FIN_SUMS, hv + L.off_gamma, P, hv + L.off_beta, P));
      b1.rdgamma = hv + L.off_gamma; b1.rdbeta = hv + L.off_beta; b1.hstride = P;
      if (gram && A.arg0 && sparse_wgrad_supported(L.wo, L.ci, L.co)) {
        SparseWgArgs sw{};
        sw.x = x0; sw.arg = A.arg0; sw.dp = A.dp[0]; sw.dpd = X.dpd[cur]; sw.rstd = A.rstd[0]; sw.m2 = X.m2[0];
        sw.gamma = theta + L.off_gamma; sw.pstride = P; sw.gammad = v + L.off_gamma; sw.vstride = P;
        sw.wpartial = pl.wgpart; sw.n = n; sw.hh = L.ho; sw.ww = L.wo; sw.co = L.co;
        LAUNCH(e, st, OP_TAN_WGRAD, 0, launch_sparse_wgrad(st, sw, T, L.ci, 1, &blk));
        GramWgArgs gw{};
        gw.g = gram; gw.spartial = pl.wgpart; gw.nblk = blk; gw.w = theta + L.off_w; gw.wstride = P; gw.wd = v + L.off_w; gw.vstride = P;
        gw.mu = A.mu[0]; gw.rstd = A.rstd[0]; gw.m1 = X.m1[0]; gw.m2 = X.m2[0];
        gw.gamma = theta + L.off_gamma; gw.pstride = P; gw.gammad = v + L.off_gamma;
        gw.dgamma = g + L.off_gamma; gw.dbeta = g + L.off_beta; gw.gstride = P;
        gw.rdgamma = hv + L.off_gamma; gw.rdbeta = hv + L.off_beta; gw.hstride = P;
        gw.out = hv + L.off_w; gw.ostride = P; gw.ci = L.ci; gw.co = L.co; gw.inv_m = 1.0 / (double)mpix;
        if (adv) { adv->b1_wgrad = 1; adv->gw_tangent = 1; adv->gw = gw; }
        else LAUNCH(e, st, OP_WGRAD_REDUCE, 0, launch_gram_wgrad(st, gw, T, 1));
        continue;
      }
      LAUNCH(e, st, OP_TAN_WGRAD, 0, launch_block1(st, b1, T, L.ci, B1_TBWD_WGRAD, &blk));
      if (adv) adv_add_seg(adv, L.off_w, 9 * L.ci * L.co, pl.wgpart, blk);
      else LAUNCH(e, st, OP_WGRAD_REDUCE, 0, launch_wgrad_reduce(st, pl.wgpart, blk, 9 * L.ci * L.co, T, hv + L.off_w, P));
      continue;
    }
    BnArgs ba{};
    ba.z = A.z[l]; ba.zd = X.zd[l]; ba.mu = A.mu[l]; ba.rstd = A.rstd[l]; ba.m1 = X.m1[l]; ba.m2 = X.m2[l];
    ba.gamma = theta + L.off_gamma; ba.beta = theta + L.off_beta; ba.pstride = P;
    ba.gammad = v + L.off_gamma; ba.betad = v + L.off_beta; ba.vstride = P;
    ba.dgamma = g + L.off_gamma; ba.dbeta = g + L.off_beta; ba.gstride = P;
    ba.dp = A.dp[l]; ba.dpd = X.dpd[cur];
    ba.partial = pl.bnpart;
    ba.n = n; ba.ho = L.ho; ba.wo = L.wo; ba.c = L.co;
    ba.inv_m = 1.f / (float)mpix;
    int blk = 0;
    const bool bn_done = fused_last && l == nl - 1;
    if (!red_done[l] && !bn_done) {
      ba.fin = fin_of(e, T, 1.0, FIN_SUMS, hv + L.off_gamma, P, hv + L.off_beta, P);
      LAUNCH(e, st, OP_BN_TAN_BWD_REDUCE, l, launch_bn_tan_bwd_reduce(st, ba, T, L.pool, &blk));
      if (!ba.fin.counter)
        LAUNCH(e, st, OP_BN_FINALIZE, l, launch_bn_finalize(st, pl.bnpart, blk, T, L.co, 1.0, FIN_SUMS, hv + L.off_gamma, P, hv + L.off_beta, P));
    }
    ba.rdgamma = hv + L.off_gamma; ba.rdbeta = hv + L.off_beta; ba.hstride = P;
    ba.out = X.rdz[l];
    ba.amax_out = cell_bind(e, pl, ba.out);
    LAUNCH(e, st, OP_BN_TAN_BWD_APPLY, l, launch_bn_tan_bwd_apply(st, ba, T, L.pool));
    WgradArgs wa{};
    wa.x[0] = l == 0 ? x0 : A.p[l - 1];
    wa.dz[0] = X.rdz[l];
    if (l > 0) { wa.x[1] = X.pd[l - 1]; wa.dz[1] = A.dz[l]; }
    if (l >= 1) {
      CELL(wa.x[0], (size_t)n * L.h * L.w * L.ci, wa.amax_x[0]);
      CELL(wa.dz[0], (size_t)n * L.ho * L.wo * L.co, wa.amax_dz[0]);
      CELL(wa.x[1], (size_t)n * L.h * L.w * L.ci, wa.amax_x[1]);
      CELL(wa.dz[1], (size_t)n * L.ho * L.wo * L.co, wa.amax_dz[1]);
    }
    wa.g = geom(L, n);
    wa.mpix = mpix;
    // (one fork of the side stream per pass, as in the primal backward pass above)
    auto issue_wgrad = [&](hipStream_t ws, WgradArgs& w, int wl) -> int {
      const Layer& Lw = e->L[wl];
      w.partial = adv ? pl.wgpart_l[wl] : (ws != st ? pl.wgpart_side : pl.wgpart);
      int nch = 0;
      LAUNCH(e, ws, OP_TAN_WGRAD, wl, launch_wgrad3x3(ws, w, T, wl > 0 ? 2 : 1, &nch));
      if (adv) adv_add_seg(adv, Lw.off_w, 9 * Lw.ci * Lw.co, w.partial, nch);
      else LAUNCH(e, ws, OP_WGRAD_REDUCE, wl, launch_wgrad_reduce(ws, w.partial, nch, 9 * Lw.ci * Lw.co, T, hv + Lw.off_w, P));
      return MI_OK;
    };
    if (l > 0 && e->overlap && e->fork_once) {
      held[nheld] = wa; held_l[nheld++] = l;
      if (l == 1) {
        hipStream_t ws = side_fork(e, st, pl.half);
        if (ws != st) forked = true;
        for (int q = 0; q < nheld; ++q) { const int rc = issue_wgrad(ws, held[q], held_l[q]); if (rc) return rc; }
        nheld = 0;
      }
    } else {
      hipStream_t ws = l > 0 ? side_fork(e, st, pl.half) : st;
      if (ws != st) forked = true;
      const int rc = issue_wgrad(ws, wa, l);
      if (rc) return rc;
    }
    if (l > 0) {
      ConvArgs ca{};
      ca.in[0] = X.rdz[l]; ca.wt[0] = theta + L.off_w;
      ca.in[1] = A.dz[l]; ca.wt[1] = v + L.off_w;
      ca.amax[0] = wa.amax_dz[0]; ca.amax[1] = wa.amax_dz[1];
      ca.wstride = P;
      ca.out = X.dpd[cur ^ 1];
      ca.g = geom_dgrad(L, n);
      ca.mpix = n * L.h * L.w;
      const float* zh_lo = dgrad_carries_reduce(e, l) ? zh_at_argmax(e, A, l - 1) : nullptr;
      const float* zhd_lo = (l - 1 == 0 && e->fuse1) ? X.zhdm : X.zhdl[l - 1];
      if (zh_lo && zhd_lo) {   // R{dgamma}, R{dbeta} of the block below in this kernel's epilogue
        const Layer& L0 = e->L[l - 1];
        ca.bp = A.p[l - 1]; ca.bzh = zh_lo; ca.bzhd = zhd_lo; ca.bdp = A.dp[l - 1]; ca.partial = pl.bnpart;
        ca.barg = (l - 1 == 0 && e->fuse1 && e->bred_arg) ? A.arg0 : nullptr;
        ca.fin = fin_of(e, T, 1.0, FIN_SUMS, hv + L0.off_gamma, P, hv + L0.off_beta, P);
        int blk0 = 0;
        LAUNCH(e, st, OP_TAN_DGRAD, l, launch_conv3x3(st, ca, T, 2, EPI_BRED, 1, &blk0));
        if (!ca.fin.counter)
          LAUNCH(e, st, OP_BN_FINALIZE, l - 1, launch_bn_finalize(st, pl.bnpart, blk0, T, L0.co, 1.0, FIN_SUMS, hv + L0.off_gamma, P, hv + L0.off_beta, P));
        red_done[l - 1] = true;
      } else {
        LAUNCH(e, st, OP_TAN_DGRAD, l, launch_conv3x3(st, ca, T, 2, EPI_NONE, 1, nullptr));
      }
      cur ^= 1;
    }
  }
  return side_join(e, st, pl.half, forked);
}

// The part of an advance launch that does not depend on the pass: sizes, strides, the never-written elements, block 1's geometry.
static AdvanceArgs advance_base(const mi_engine* e, float* g) {
  AdvanceArgs a{};
  a.g = g; a.gstride = e->PS; a.ostride = e->PS; a.n = (unsigned)e->PS;
  a.nzero = e->nzero;
  for (int z = 0; z < e->nzero; ++z) { a.zoff[z] = e->zoff[z]; a.zlen[z] = e->zlen[z]; }
  a.off_w1 = (unsigned)e->L[0].off_w;
  a.ci = (e->L[0].ci == 1 || e->L[0].ci == 3) ? e->L[0].ci : 0;
  a.co = e->L[0].co;
  a.counter = e->counters;
  return a;
}

// grad_tasks: the first grad_tasks of the `tasks` tasks are TRAIN tasks (query backward, second-order adjoint recursion, summed into
// meta_grad_out); the rest are VALIDATION tasks of the same meta-iteration (reference maml_vision.py:117-124: the same clone +
// fast_adapt without backward), whose K support steps and query forward run in the SAME launches as the train tasks'.
static int meta_batch_maml_impl(mi_engine* e, void* stream, const float* theta, const float* data, const int64_t* labels,
                                int tasks, int ways, int shots, int adapt_steps, float inner_lr, int second_order, int grad_tasks,
                                float* loss_out, float* acc_out, float* meta_grad_out, float* logits_out,
                                void* workspace, size_t workspace_bytes) {
  if (!e) return fail(nullptr, MI_ERR_ARG, "null engine");
  if (!theta || !data || !labels || !loss_out || !acc_out || !workspace) return fail(e, MI_ERR_ARG, "null pointer argument");
  const int with_grad = grad_tasks > 0;
  if (with_grad && !meta_grad_out) return fail(e, MI_ERR_ARG, "meta_grad_out is NULL but a gradient was asked for");
  if (tasks < 1 || shots < 1 || adapt_steps < 0) return fail(e, MI_ERR_ARG, "tasks/shots must be >= 1, adapt_steps >= 0");
  if (grad_tasks < 0 || grad_tasks > tasks) return fail(e, MI_ERR_ARG, "grad_tasks must be in 0..tasks");
  if (ways != e->d.ways) return fail(e, MI_ERR_ARG, "ways differs from the engine's classifier width");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int T = tasks, Tg = grad_tasks, K = adapt_steps, ns = ways * shots, nq = ways * shots;
  const int so = (second_order && with_grad) ? 1 : 0;
  e->export_pass = 0;
  Plan pl;
  make_plan(e, workspace, T, ns, nq, K, so, pl);
  if (pl.bytes > workspace_bytes)
    return fail(e, MI_ERR_WORKSPACE, "workspace too small: need " + std::to_string(pl.bytes) + " bytes");
  const size_t TP = (size_t)T * e->PS;
  const Layer& L0 = e->L[0];
  const bool tail = e->fuse_tail && e->counters && T <= 65535 /* launch_advance puts the tasks on grid.y */ && T <= mi_engine::kMaxCounterTasks && 1024 % L0.co == 0;
  const double inv_m0 = 1.0 / ((double)ns * L0.ho * L0.wo);
  { const int brc = plan_begin(e, st, pl); if (brc) return brc; }
  LAUNCH(e, st, OP_MISC, 0, launch_prepare_batch(st, data, labels, T, 2 * ns, e->d.in_channels, e->d.in_h, e->d.in_w, pl.xs, pl.xq, pl.ys, pl.yq));
  LAUNCH(e, st, OP_MISC, 1, launch_gather_params(st, theta, 0, e->perm_dev, (int)e->P, (int)e->PS, T, pl.theta));
  if (pl.gram_s)
    LAUNCH(e, st, OP_GRAM, 0, launch_input_gram(st, pl.xs, T, ns, e->L[0].h, e->L[0].w, e->L[0].ci, pl.gram_part, pl.gram_s));
  const double* gq = pl.gram_q;          // (also for forward-only calls and tasks: the same query forward whether or not a backward half follows)
  bool gq_side = false;
  if (gq && gram_query_env() == 2 && e->overlap) {   // the query images' Gram matrix beside the inner loop
    hipStream_t ws = side_fork(e, st, pl.half);
    gq_side = ws != st;
    LAUNCH(e, ws, OP_GRAM, 0, launch_input_gram(ws, pl.xq, T, nq, e->L[0].h, e->L[0].w, e->L[0].ci, pl.gram_part_q, pl.gram_q));
  }
  const bool stats0 = tail && pl.gram_s && K > 0;
  if (stats0) {   // block 1's statistics of the first support pass, from theta_0 (nothing folded, nothing written)
    AdvanceArgs a0 = advance_base(e, pl.theta);
    a0.nzero = 0;
    a0.stats = 1; a0.gram = pl.gram_s; a0.out0 = pl.sup[0].mu[0]; a0.out1 = pl.sup[0].rstd[0]; a0.inv_m = inv_m0;
    LAUNCH(e, st, OP_MISC, 2, launch_advance(st, a0, T));
  }
  for (int k = 0; k < K; ++k) {
    ActSet& A = so ? pl.sup[k] : pl.sup[0];
    float* th = pl.theta + (size_t)k * TP;
    float* gk = pl.g + (size_t)k * TP;
    AdvanceArgs adv = advance_base(e, gk);
    int rc = pass_fwd_bwd(e, st, pl, A, pl.xs, pl.ys, ns, T, th, gk, pl.tmp_loss, pl.tmp_acc, nullptr, true, pl.gram_s, tail ? &adv : nullptr,
                          tail && pl.gram_s);
    if (rc) return rc;
    if (tail) {   // g_k finished, theta_{k+1} = theta_k - lr g_k, and block 1's statistics of the next support pass
      adv.a = th; adv.out = th + TP; adv.alpha = inner_lr;
      if (pl.gram_s && k + 1 < K) {
        ActSet& An = so ? pl.sup[k + 1] : pl.sup[0];
        adv.stats = 1; adv.gram = pl.gram_s; adv.out0 = An.mu[0]; adv.out1 = An.rstd[0]; adv.inv_m = inv_m0;
      }
      LAUNCH(e, st, OP_MISC, 2, launch_advance(st, adv, T));
    } else {
      LAUNCH(e, st, OP_MISC, 2, launch_axpy(st, th, gk, inner_lr, TP, th + TP));
    }
  }
  float* thK = pl.theta + (size_t)K * TP;
  AdvanceArgs advq = advance_base(e, pl.lam);
  if (gq && gq_side) { const int jrc = side_join(e, st, pl.half, true); if (jrc) return jrc; }
  else if (gq) LAUNCH(e, st, OP_GRAM, 0, launch_input_gram(st, pl.xq, T, nq, e->L[0].h, e->L[0].w, e->L[0].ci, pl.gram_part_q, pl.gram_q));
  int rc = pass_fwd_bwd(e, st, pl, pl.qry, pl.xq, pl.yq, nq, T, thK, pl.lam, loss_out, acc_out, logits_out, with_grad != 0, gq,
                        (tail && with_grad) ? &advq : nullptr, false, Tg);
  if (rc) return rc;
  if (!with_grad) return MI_OK;
  // debug trace layout (floats): theta [K+1][T][P] | g [K][T][P] | lam_in [K][T][P] | hv [K][T][P], reference parameter order
  const size_t TPr = (size_t)T * e->P;
  const bool tr = e->trace && e->trace_floats >= (size_t)(4 * K + 1) * TPr;
  if (e->trace && !tr) return fail(e, MI_ERR_WORKSPACE, "debug trace buffer too small: need " + std::to_string((size_t)(4 * K + 1) * TPr) + " floats");
  if (tr && Tg != T) return fail(e, MI_ERR_ARG, "the debug trace covers calls whose tasks all take the backward half");
  // tangent statistics of block 1 for the Hessian-vector pass over support pass k (weights theta_k), direction = the finished vector
  auto tangent_stats = [&](AdvanceArgs& a, int k) {
    a.stats = 2; a.gram = pl.gram_s; a.sw = pl.theta + (size_t)k * TP + L0.off_w; a.swstride = e->PS;
    a.mu_in = pl.sup[k].mu[0]; a.rstd_in = pl.sup[k].rstd[0]; a.out0 = pl.tan.m1[0]; a.out1 = pl.tan.m2[0]; a.inv_m = inv_m0;
  };
  if (tail) {   // lam = grad L_query(theta_K) finished (folds only)
    if (so && K > 0 && pl.gram_s) tangent_stats(advq, K - 1);
    LAUNCH(e, st, OP_MISC, 2, launch_advance(st, advq, Tg));
  }
  float *lam_cur = pl.lam, *lam_nxt = pl.lam2;
  if (so) {
    for (int k = K - 1; k >= 0; --k) {
      if (tr) HIPCHK(e, launch_scatter_tasks(st, lam_cur, e->perm_dev, (int)e->P, (int)e->PS, T, e->trace + (size_t)(2 * K + 1 + k) * TPr));
      AdvanceArgs adv = advance_base(e, pl.hv);
      rc = pass_hvp(e, st, pl, pl.sup[k], pl.xs, ns, Tg, pl.theta + (size_t)k * TP, pl.g + (size_t)k * TP, lam_cur, pl.hv, pl.gram_s, nullptr, nullptr,
                    tail ? &adv : nullptr, tail);
      if (rc) return rc;
      if (tail) {   // H lam finished, lam' = lam - lr H lam (into the other buffer: the row workgroups of the advance launch read lam's
                    // block-1 entries while others write lam'), and the tangent statistics of the next Hessian-vector pass
        adv.a = lam_cur; adv.out = lam_nxt; adv.alpha = inner_lr;
        if (k > 0 && pl.gram_s) tangent_stats(adv, k - 1);
        LAUNCH(e, st, OP_MISC, 2, launch_advance(st, adv, Tg));
        float* t = lam_cur; lam_cur = lam_nxt; lam_nxt = t;
        if (tr) HIPCHK(e, launch_scatter_tasks(st, pl.hv, e->perm_dev, (int)e->P, (int)e->PS, T, e->trace + (size_t)(3 * K + 1 + k) * TPr));
      } else {
        if (tr) HIPCHK(e, launch_scatter_tasks(st, pl.hv, e->perm_dev, (int)e->P, (int)e->PS, T, e->trace + (size_t)(3 * K + 1 + k) * TPr));
        LAUNCH(e, st, OP_MISC, 2, launch_axpy(st, lam_cur, pl.hv, inner_lr, (size_t)Tg * e->PS, lam_cur));
      }
    }
  }
  if (tr) {
    for (int k = 0; k <= K; ++k)
      HIPCHK(e, launch_scatter_tasks(st, pl.theta + (size_t)k * TP, e->perm_dev, (int)e->P, (int)e->PS, T, e->trace + (size_t)k * TPr));
    for (int k = 0; k < K; ++k)
      HIPCHK(e, launch_scatter_tasks(st, pl.g + (size_t)k * TP, e->perm_dev, (int)e->P, (int)e->PS, T, e->trace + (size_t)(K + 1 + k) * TPr));
  }
  LAUNCH(e, st, OP_MISC, 3, launch_scatter_sum(st, lam_cur, e->perm_dev, (int)e->P, (int)e->PS, Tg, meta_grad_out));
  return MI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// ANIL (reference vision/anil_vision.py:86-99,116-122 + utils/data_pre.py:118-119): the conv trunk runs ONCE per task on all
// 2*shots*ways images (BatchNorm statistics over support and query together), the inner loop adapts only the Linear head,
// the outer gradient reaches the trunk through the query features (directly) and through the support features (second-
// order path:  d L_q / d f_s = -lr * sum_k  R_{lam_{k+1}}{ dL_s/df_s }(w_k), the head tangent kernel's R{df} output).
struct AnilPlan {
  float *theta, *g, *lam, *hv;
  float* x;                 // [T][2n][H][W][C] NHWC
  int32_t *ys, *yq;
  ActSet act;               // trunk activations over 2n images
  float *fs, *fq, *dfs, *dfq, *rdf;   // [T][n][feat]
  float *prob, *dl;         // [K+1][T][n][ways]  (support steps 0..K-1, query at K)
  float *tmp_loss, *tmp_acc;
  float* hscr;
  Plan scratch;             // bnpart / wgpart live here
  size_t bytes;
};

static void make_anil_plan(const mi_engine* e, void* ws, int T, int n, int K, AnilPlan& ap) {
  Bump b{reinterpret_cast<char*>(ws), 0};
  const size_t TP = (size_t)T * e->PS;
  ap.theta = b.take<float>(TP * (K + 1));
  ap.g = b.take<float>(TP * (K > 0 ? K : 1));
  ap.lam = b.take<float>(TP);
  ap.hv = b.take<float>(TP);
  const size_t img = (size_t)e->d.in_h * e->d.in_w * e->d.in_channels;
  ap.x = b.take<float>((size_t)T * 2 * n * img);
  ap.ys = b.take<int32_t>((size_t)T * n);
  ap.yq = b.take<int32_t>((size_t)T * n);
  plan_actset(e, b, ap.act, T, 2 * n, true);
  const size_t fsz = (size_t)T * n * e->feat;
  ap.fs = b.take<float>(fsz); ap.fq = b.take<float>(fsz); ap.dfs = b.take<float>(fsz); ap.dfq = b.take<float>(fsz);
  ap.rdf = b.take<float>(fsz);
  ap.prob = b.take<float>((size_t)(K + 1) * T * n * e->d.ways);
  ap.dl = b.take<float>((size_t)(K + 1) * T * n * e->d.ways);
  ap.tmp_loss = b.take<float>(T);
  ap.tmp_acc = b.take<float>(T);
  ap.hscr = b.take<float>((size_t)T * n * (e->d.ways + 2));
  size_t bnp = 0, wgp = 0;
  for (const Layer& L : e->L) {
    const ConvGeom gg = geom(L, 2 * n);
    int blk = conv_max_blocks_per_task(gg);
    const int bb = bn_blocks_per_task(2 * n, L.ho, L.wo, L.co, L.pool, 1);
    if (bb > blk) blk = bb;
    const size_t need = (size_t)T * blk * 2 * L.co;
    if (need > bnp) bnp = need;
    size_t w = wgrad_partial_floats(gg, T);
    if (&L == &e->L[0] && e->fuse1) {
      int bpt = block1_blocks_per_task(2 * n, L.ho, L.wo, L.co, T);
      const int sb = sparse_wgrad_blocks_per_task(2 * n, L.ho, L.wo, L.co, T);
      if (sb > bpt) bpt = sb;
      w = (size_t)T * bpt * 9 * L.ci * L.co;
    }
    if (w > wgp) wgp = w;
  }
  ap.scratch.bnpart = b.take<double>(bnp);
  ap.scratch.wgpart = b.take<float>(wgp);
  ap.scratch.wgpart_side = b.take<float>(wgp);
  ap.scratch.gram_part = ap.scratch.gram_s = nullptr;
  ap.scratch.cell_cap = 8 + 4 * (int)e->L.size();
  ap.scratch.cells = conv_operand_form() == 2 ? b.take<unsigned>((size_t)ap.scratch.cell_cap * T * MI_CELL_WORDS) : nullptr;
  ap.scratch.cell_T = T;
  if (e->fuse1 && e->gram1 && gram_supported(e->L[0].w, e->L[0].ci)) {   // statistics + weight gradient of block 1 from the Gram matrix
    ap.scratch.gram_part = b.take<double>(gram_partial_doubles(T, 2 * n, e->L[0].h, e->L[0].ci));
    ap.scratch.gram_s = b.take<double>(gram_doubles(T, e->L[0].ci));
  }
  ap.bytes = align_up(b.off, 256);
}

extern "C" int mi_anil_workspace_bytes(const mi_engine* e, int tasks, int ways, int shots, int adapt_steps, size_t* bytes) {
  if (!e || !bytes || tasks < 1 || ways < 1 || shots < 1 || adapt_steps < 0) return MI_ERR_ARG;
  AnilPlan ap;
  make_anil_plan(e, nullptr, tasks, ways * shots, adapt_steps, ap);
  *bytes = ap.bytes;
  return MI_OK;
}

static int meta_batch_anil_impl(mi_engine* e, void* stream, const float* theta, const float* data, const int64_t* labels,
                                int tasks, int ways, int shots, int adapt_steps, float inner_lr, int second_order,
                                int with_grad, float* loss_out, float* acc_out, float* meta_grad_out, float* logits_out,
                                void* workspace, size_t workspace_bytes) {
  if (!e) return fail(nullptr, MI_ERR_ARG, "null engine");
  if (!theta || !data || !labels || !loss_out || !acc_out || !workspace) return fail(e, MI_ERR_ARG, "null pointer argument");
  if (with_grad && !meta_grad_out) return fail(e, MI_ERR_ARG, "meta_grad_out is NULL but with_grad != 0");
  if (e->d.head_mean_pool) return fail(e, MI_ERR_ARG, "ANIL features are flattened (view(-1, fc_neurons)), not mean-pooled");
  if (tasks < 1 || shots < 1 || adapt_steps < 0 || ways != e->d.ways) return fail(e, MI_ERR_ARG, "bad tasks/shots/steps/ways");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int T = tasks, K = adapt_steps, n = ways * shots, nl = (int)e->L.size();
  AnilPlan ap;
  make_anil_plan(e, workspace, T, n, K, ap);
  if (ap.bytes > workspace_bytes)
    return fail(e, MI_ERR_WORKSPACE, "workspace too small: need " + std::to_string(ap.bytes) + " bytes");
  const size_t TP = (size_t)T * e->PS, fsz = (size_t)T * n * e->feat, pw = (size_t)T * n * e->d.ways;
  LAUNCH(e, st, OP_MISC, 0, launch_nchw_to_nhwc(st, data, (size_t)T * 2 * n, e->d.in_channels, e->d.in_h, e->d.in_w, ap.x));
  LAUNCH(e, st, OP_MISC, 0, launch_split_labels(st, labels, T, 2 * n, ap.ys, ap.yq));
  LAUNCH(e, st, OP_MISC, 1, launch_gather_params(st, theta, 0, e->perm_dev, (int)e->P, (int)e->PS, T, ap.theta));
  const double* gram = (with_grad && workspace) ? ap.scratch.gram_s : nullptr;       // pays off only with a backward pass
  if (gram)
    LAUNCH(e, st, OP_GRAM, 0, launch_input_gram(st, ap.x, T, 2 * n, e->L[0].h, e->L[0].w, e->L[0].ci, ap.scratch.gram_part, ap.scratch.gram_s));
  int rc = plan_begin(e, st, ap.scratch);
  if (rc) return rc;
  rc = trunk_forward(e, st, ap.scratch, ap.act, ap.x, 2 * n, T, ap.theta, gram);     // features(data) on all rows
  if (rc) return rc;
  e->export_pass = 0;
  rc = export_bn_stats(e, st, ap.act, T);
  if (rc) return rc;
  LAUNCH(e, st, OP_MISC, 4, launch_split_rows(st, ap.act.p[nl - 1], T, 2 * n, e->feat, ap.fs, ap.fq));
  for (int k = 0; k < K; ++k) {                                                    // head-only inner loop
    float* th = ap.theta + (size_t)k * TP;
    float* gk = ap.g + (size_t)k * TP;
    HIPCHK(e, hipMemsetAsync(gk, 0, TP * sizeof(float), st));
    rc = head_pass(e, st, ap.hscr, ap.fs, ap.ys, n, T, th, gk, ap.tmp_loss, ap.tmp_acc, nullptr, ap.prob + k * pw, ap.dl + k * pw,
                   nullptr, true);
    if (rc) return rc;
    LAUNCH(e, st, OP_MISC, 2, launch_axpy(st, th, gk, inner_lr, TP, th + TP));
  }
  float* thK = ap.theta + (size_t)K * TP;
  if (with_grad) HIPCHK(e, hipMemsetAsync(ap.lam, 0, TP * sizeof(float), st));
  rc = head_pass(e, st, ap.hscr, ap.fq, ap.yq, n, T, thK, ap.lam, loss_out, acc_out, logits_out, ap.prob + K * pw, ap.dl + K * pw, ap.dfq,
                 with_grad != 0);
  if (rc || !with_grad) return rc;
  HIPCHK(e, hipMemsetAsync(ap.dfs, 0, fsz * sizeof(float), st));
  if (second_order) {
    for (int k = K - 1; k >= 0; --k) {
      HIPCHK(e, hipMemsetAsync(ap.hv, 0, TP * sizeof(float), st));
      HeadArgs ha{};
      ha.f = ap.fs; ha.fd = nullptr;
      ha.wl = ap.theta + (size_t)k * TP + e->off_wl; ha.bl = ap.theta + (size_t)k * TP + e->off_bl; ha.pstride = e->PS;
      ha.wld = ap.lam + e->off_wl; ha.bld = ap.lam + e->off_bl; ha.vstride = e->PS;
      ha.prob = ap.prob + k * pw; ha.dl = ap.dl + k * pw;
      ha.dwl = ap.hv + e->off_wl; ha.dbl = ap.hv + e->off_bl; ha.gstride = e->PS;
      ha.df = ap.rdf;
      ha.n = n; ha.feat = e->feat; ha.ways = e->d.ways;
      head_scratch(e, ha, ap.hscr, T, n);
      LAUNCH(e, st, OP_HEAD_TAN, 0, launch_head_tangent(st, ha, T));
      LAUNCH(e, st, OP_MISC, 2, launch_axpy(st, ap.dfs, ap.rdf, inner_lr, fsz, ap.dfs));   // dfs -= lr * R{df}
      LAUNCH(e, st, OP_MISC, 2, launch_axpy(st, ap.lam, ap.hv, inner_lr, TP, ap.lam));     // lam -= lr * H lam
    }
  }
  LAUNCH(e, st, OP_MISC, 4, launch_interleave_rows(st, ap.dfs, ap.dfq, T, n, e->feat, ap.act.dp[nl - 1]));
  rc = trunk_backward(e, st, ap.scratch, ap.act, ap.x, 2 * n, T, ap.theta, ap.lam, gram);  // trunk grads join the head part in lam
  if (rc) return rc;
  LAUNCH(e, st, OP_MISC, 3, launch_scatter_sum(st, ap.lam, e->perm_dev, (int)e->P, (int)e->PS, T, meta_grad_out));
  return MI_OK;
}

// ---- graph replay of the two fused calls
typedef int (*MetaBatchFn)(mi_engine*, void*, const float*, const float*, const int64_t*, int, int, int, int, float, int, int, float*,
                           float*, float*, float*, void*, size_t);
static int meta_batch_entry(MetaBatchFn fn, int which, mi_engine* e, void* stream, const float* theta, const float* data,
                            const int64_t* labels, int tasks, int ways, int shots, int adapt_steps, float inner_lr, int second_order,
                            int with_grad, float* loss_out, float* acc_out, float* meta_grad_out, float* logits_out, void* workspace,
                            size_t workspace_bytes) {
  if (!e || !e->graph_on || e->prof_on || e->trace || e->bn_export || !stream)     // (capture is not permitted on the legacy default stream)
    return fn(e, stream, theta, data, labels, tasks, ways, shots, adapt_steps, inner_lr, second_order, with_grad, loss_out, acc_out,
              meta_grad_out, logits_out, workspace, workspace_bytes);
  unsigned lr_bits;
  memcpy(&lr_bits, &inner_lr, sizeof lr_bits);
  const std::vector<unsigned long long> key = {
      (unsigned long long)which, (unsigned long long)(uintptr_t)stream, (unsigned long long)(uintptr_t)theta,
      (unsigned long long)(uintptr_t)data, (unsigned long long)(uintptr_t)labels, (unsigned long long)tasks, (unsigned long long)ways,
      (unsigned long long)shots, (unsigned long long)adapt_steps, (unsigned long long)lr_bits, (unsigned long long)second_order,
      (unsigned long long)with_grad, (unsigned long long)(uintptr_t)loss_out, (unsigned long long)(uintptr_t)acc_out,
      (unsigned long long)(uintptr_t)meta_grad_out, (unsigned long long)(uintptr_t)logits_out, (unsigned long long)(uintptr_t)workspace,
      (unsigned long long)workspace_bytes, (unsigned long long)e->fuse1 + 2ull * e->gram1 + (1ull << 40) * e->gramq + 4ull * e->overlap + 8ull * e->fuse_fin + 16ull * e->fuse_b1red + 32ull * e->fuse_tail + 64ull * e->fork_once + 128ull * e->fuse_last + 256ull * kernel_selection_key()};
  mi_engine::GraphEntry* ent = nullptr;
  for (auto& g : e->graphs)
    if (g.key == key) { ent = &g; break; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (ent && ent->exec) {
    HIPCHK(e, hipGraphLaunch(ent->exec, st));
    return MI_OK;
  }
  if (!ent) {                                   // first sight: run eagerly (kernel attributes, lazily created streams and events)
    if (e->graphs.size() >= 8) {                // small cache: drop the oldest
      if (e->graphs.front().exec) (void)hipGraphExecDestroy(e->graphs.front().exec);
      if (e->graphs.front().graph) (void)hipGraphDestroy(e->graphs.front().graph);
      e->graphs.erase(e->graphs.begin());
    }
    e->graphs.push_back(mi_engine::GraphEntry{});
    e->graphs.back().key = key;
    e->graphs.back().seen = 1;
    return fn(e, stream, theta, data, labels, tasks, ways, shots, adapt_steps, inner_lr, second_order, with_grad, loss_out, acc_out,
              meta_grad_out, logits_out, workspace, workspace_bytes);
  }
  // second sight: capture the launch sequence (nothing executes), instantiate, launch
  HIPCHK(e, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  const int rc = fn(e, stream, theta, data, labels, tasks, ways, shots, adapt_steps, inner_lr, second_order, with_grad, loss_out, acc_out,
                    meta_grad_out, logits_out, workspace, workspace_bytes);
  hipGraph_t graph = nullptr;
  const hipError_t ce = hipStreamEndCapture(st, &graph);
  if (rc != MI_OK || ce != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    e->graph_on = false;                         // do not try again; the caller gets the error of this call
    return rc != MI_OK ? rc : fail(e, MI_ERR_HIP, std::string("graph capture failed: ") + hipGetErrorString(ce));
  }
  hipGraphExec_t exec = nullptr;
  if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGraphDestroy(graph);
    e->graph_on = false;
    return fail(e, MI_ERR_HIP, "hipGraphInstantiate failed");
  }
  ent->graph = graph;
  ent->exec = exec;
  HIPCHK(e, hipGraphLaunch(exec, st));
  return MI_OK;
}

extern "C" int mi_meta_batch_maml(mi_engine* e, void* stream, const float* theta, const float* data, const int64_t* labels,
                                  int tasks, int ways, int shots, int adapt_steps, float inner_lr, int second_order,
                                  int with_grad, float* loss_out, float* acc_out, float* meta_grad_out, float* logits_out,
                                  void* workspace, size_t workspace_bytes) {
  return meta_batch_entry(meta_batch_maml_impl, 0, e, stream, theta, data, labels, tasks, ways, shots, adapt_steps, inner_lr,
                          second_order, with_grad ? tasks : 0, loss_out, acc_out, meta_grad_out, logits_out, workspace, workspace_bytes);
}
// Train and validation halves of one meta-iteration in the same launches (reference maml_vision.py:102-124): the first grad_tasks
// tasks are the train tasks (their summed meta-gradient goes to meta_grad_out), the rest are adapted and scored only.
extern "C" int mi_meta_batch_maml_tv(mi_engine* e, void* stream, const float* theta, const float* data, const int64_t* labels,
                                     int tasks, int grad_tasks, int ways, int shots, int adapt_steps, float inner_lr, int second_order,
                                     float* loss_out, float* acc_out, float* meta_grad_out, float* logits_out,
                                     void* workspace, size_t workspace_bytes) {
  if (grad_tasks < 0 || grad_tasks > tasks) return fail(e, MI_ERR_ARG, "grad_tasks must be in 0..tasks");
  return meta_batch_entry(meta_batch_maml_impl, 0, e, stream, theta, data, labels, tasks, ways, shots, adapt_steps, inner_lr,
                          second_order, grad_tasks, loss_out, acc_out, meta_grad_out, logits_out, workspace, workspace_bytes);
}
extern "C" int mi_meta_batch_anil(mi_engine* e, void* stream, const float* theta, const float* data, const int64_t* labels,
                                  int tasks, int ways, int shots, int adapt_steps, float inner_lr, int second_order,
                                  int with_grad, float* loss_out, float* acc_out, float* meta_grad_out, float* logits_out,
                                  void* workspace, size_t workspace_bytes) {
  return meta_batch_entry(meta_batch_anil_impl, 1, e, stream, theta, data, labels, tasks, ways, shots, adapt_steps, inner_lr,
                          second_order, with_grad, loss_out, acc_out, meta_grad_out, logits_out, workspace, workspace_bytes);
}

// Plain forward of the classifier (BatchNorm in train mode, i.e. statistics of the n images of each task batch), no
// adaptation: `learner(x)` / `model(x)` of the reference (vision_models.py:51-55,107-110).
extern "C" int mi_forward_logits(mi_engine* e, void* stream, const float* theta, const float* x, int tasks, int n,
                                 float* logits_out, void* workspace, size_t workspace_bytes) {
  if (!e) return fail(nullptr, MI_ERR_ARG, "null engine");
  if (!theta || !x || !logits_out || !workspace || tasks < 1 || n < 1) return fail(e, MI_ERR_ARG, "bad forward arguments");
  ExportPause no_export(e);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  Plan pl;
  make_plan(e, workspace, tasks, n, n, 0, 0, pl);
  if (pl.bytes > workspace_bytes)
    return fail(e, MI_ERR_WORKSPACE, "workspace too small: need " + std::to_string(pl.bytes) + " bytes");
  { const int brc = plan_begin(e, st, pl); if (brc) return brc; }
  LAUNCH(e, st, OP_MISC, 0, launch_nchw_to_nhwc(st, x, (size_t)tasks * n, e->d.in_channels, e->d.in_h, e->d.in_w, pl.xq));
  LAUNCH(e, st, OP_MISC, 1, launch_gather_params(st, theta, 0, e->perm_dev, (int)e->P, (int)e->PS, tasks, pl.theta));
  HIPCHK(e, hipMemsetAsync(pl.yq, 0, (size_t)tasks * n * sizeof(int32_t), st));
  return pass_fwd_bwd(e, st, pl, pl.qry, pl.xq, pl.yq, n, tasks, pl.theta, pl.lam, pl.tmp_loss, pl.tmp_acc, logits_out, false);
}

extern "C" int mi_forward_workspace_bytes(const mi_engine* e, int tasks, int n, size_t* bytes) {
  if (!e || !bytes || tasks < 1 || n < 1) return MI_ERR_ARG;
  Plan pl;
  make_plan(e, nullptr, tasks, n, n, 0, 0, pl);
  *bytes = pl.bytes;
  return MI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Step-wise learner (reference: `learner(x)`, `learner.adapt(loss)`, `get_rep`, `get_rep_i` on a learn2learn clone --
// misc_scripts/cl_vision.py:56-66, misc_scripts/rc_vision.py:66-86, core_functions/maml.py:15-19).  The fast weights live
// with the caller (theta [theta_tasks][P], reference order); forward and backward are two stateless calls, the backward
// re-runs the forward into the workspace instead of keeping activations alive between calls.
static int learner_args(mi_engine* e, const float* theta, int theta_tasks, const float* x, int tasks, int n, void* workspace,
                        size_t workspace_bytes, Plan& pl) {
  if (!e) return fail(nullptr, MI_ERR_ARG, "null engine");
  if (!theta || !x || !workspace || tasks < 1 || n < 1) return fail(e, MI_ERR_ARG, "bad learner arguments");
  if (theta_tasks != 1 && theta_tasks != tasks) return fail(e, MI_ERR_ARG, "theta_tasks must be 1 (shared) or == tasks");
  make_plan(e, workspace, tasks, n, n, 0, 0, pl);
  if (pl.bytes > workspace_bytes)
    return fail(e, MI_ERR_WORKSPACE, "workspace too small: need " + std::to_string(pl.bytes) + " bytes");
  return MI_OK;
}

extern "C" int mi_learner_forward(mi_engine* e, void* stream, const float* theta, int theta_tasks, const float* x, int tasks,
                                  int n, float* logits_out, int rep_layer, float* rep_out, void* workspace,
                                  size_t workspace_bytes) {
  Plan pl;
  int rc = learner_args(e, theta, theta_tasks, x, tasks, n, workspace, workspace_bytes, pl);
  if (rc) return rc;
  ExportPause no_export(e);
  const int nl = (int)e->L.size();
  if (rep_out && (rep_layer < 1 || rep_layer > nl)) return fail(e, MI_ERR_ARG, "rep_layer must be in 1..layers");
  if (!logits_out && !rep_out) return fail(e, MI_ERR_ARG, "nothing to compute: logits_out and rep_out are both NULL");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  rc = plan_begin(e, st, pl);
  if (rc) return rc;
  LAUNCH(e, st, OP_MISC, 0, launch_nchw_to_nhwc(st, x, (size_t)tasks * n, e->d.in_channels, e->d.in_h, e->d.in_w, pl.xq));
  LAUNCH(e, st, OP_MISC, 1, launch_gather_params(st, theta, theta_tasks == 1 ? 0 : e->P, e->perm_dev, (int)e->P, (int)e->PS, tasks, pl.theta));
  HIPCHK(e, hipMemsetAsync(pl.yq, 0, (size_t)tasks * n * sizeof(int32_t), st));
  rc = pass_fwd_bwd(e, st, pl, pl.qry, pl.xq, pl.yq, n, tasks, pl.theta, pl.lam, pl.tmp_loss, pl.tmp_acc, logits_out, false);
  if (rc) return rc;
  if (rep_out) {
    const Layer& L = e->L[rep_layer - 1];
    LAUNCH(e, st, OP_MISC, 0, launch_nhwc_to_nchw(st, pl.qry.p[rep_layer - 1], (size_t)tasks * n, L.co, L.hp, L.wp, rep_out));
  }
  return MI_OK;
}

extern "C" int mi_learner_backward(mi_engine* e, void* stream, const float* theta, int theta_tasks, const float* x,
                                   const float* dlogits, int tasks, int n, float* grad_out, void* workspace,
                                   size_t workspace_bytes) {
  Plan pl;
  int rc = learner_args(e, theta, theta_tasks, x, tasks, n, workspace, workspace_bytes, pl);
  if (rc) return rc;
  ExportPause no_export(e);
  if (!dlogits || !grad_out) return fail(e, MI_ERR_ARG, "null dlogits / grad_out");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nl = (int)e->L.size();
  ActSet& A = pl.qry;
  rc = plan_begin(e, st, pl);
  if (rc) return rc;
  LAUNCH(e, st, OP_MISC, 0, launch_nchw_to_nhwc(st, x, (size_t)tasks * n, e->d.in_channels, e->d.in_h, e->d.in_w, pl.xq));
  LAUNCH(e, st, OP_MISC, 1, launch_gather_params(st, theta, theta_tasks == 1 ? 0 : e->P, e->perm_dev, (int)e->P, (int)e->PS, tasks, pl.theta));
  rc = trunk_forward(e, st, pl, A, pl.xq, n, tasks, pl.theta);
  if (rc) return rc;
  if (e->d.head_mean_pool) HIPCHK(e, launch_spatial_mean(st, A.p[nl - 1], A.f, tasks * n, e->head_hw, e->head_c));
  HIPCHK(e, hipMemsetAsync(pl.lam, 0, (size_t)tasks * e->PS * sizeof(float), st));
  HeadArgs ha{};
  ha.f = A.f;
  ha.wl = pl.theta + e->off_wl; ha.bl = pl.theta + e->off_bl; ha.pstride = e->PS;
  ha.dl = const_cast<float*>(dlogits);
  ha.dwl = pl.lam + e->off_wl; ha.dbl = pl.lam + e->off_bl; ha.gstride = e->PS;
  ha.df = A.df;
  ha.n = n; ha.feat = e->feat; ha.ways = e->d.ways;
  LAUNCH(e, st, OP_HEAD, 0, launch_head_grads(st, ha, tasks));
  if (e->d.head_mean_pool) HIPCHK(e, launch_spatial_mean_bwd(st, A.df, A.dp[nl - 1], tasks * n, e->head_hw, e->head_c));
  rc = trunk_backward(e, st, pl, A, pl.xq, n, tasks, pl.theta, pl.lam);
  if (rc) return rc;
  if (theta_tasks == 1) {
    LAUNCH(e, st, OP_MISC, 3, launch_scatter_sum(st, pl.lam, e->perm_dev, (int)e->P, (int)e->PS, tasks, grad_out));
  } else {
    LAUNCH(e, st, OP_MISC, 3, launch_scatter_tasks(st, pl.lam, e->perm_dev, (int)e->P, (int)e->PS, tasks, grad_out));
  }
  return MI_OK;
}

// Double backward of the step-wise learner: with s(theta) = sum(logits(theta) * dlogits) and g = ds/dtheta (mi_learner_backward),
// autograd asks for the vector-Jacobian products of (theta, dlogits) -> g with a cotangent v on g:
//   grad_theta_out = (d^2 s / dtheta^2) v   (forward-over-reverse, dlogits held fixed),   logits_dot_out = J(theta) v.
// This is what makes `learner.adapt(loss)` of a second-order learner differentiable (learn2learn MAML.adapt with create_graph,
// driven step-wise at misc_scripts/rc_vision.py:67-70); the cross-entropy's own curvature reaches theta through logits_dot_out
// and an ordinary mi_learner_backward, by the chain rule autograd already applies.
extern "C" int mi_learner_hvp_workspace_bytes(const mi_engine* e, int tasks, int n, size_t* bytes) {
  if (!e || !bytes || tasks < 1 || n < 1) return MI_ERR_ARG;
  Plan pl;
  make_plan(e, nullptr, tasks, n, n, 1, 1, pl);
  *bytes = pl.bytes;
  return MI_OK;
}

extern "C" int mi_learner_hvp(mi_engine* e, void* stream, const float* theta, int theta_tasks, const float* x, const float* dlogits,
                              const float* v, int tasks, int n, float* grad_theta_out, float* logits_dot_out, void* workspace,
                              size_t workspace_bytes) {
  if (!e) return fail(nullptr, MI_ERR_ARG, "null engine");
  if (!theta || !x || !dlogits || !v || !grad_theta_out || !logits_dot_out || !workspace || tasks < 1 || n < 1)
    return fail(e, MI_ERR_ARG, "bad mi_learner_hvp arguments");
  if (theta_tasks != 1 && theta_tasks != tasks) return fail(e, MI_ERR_ARG, "theta_tasks must be 1 (shared) or == tasks");
  ExportPause no_export(e);
  Plan pl;
  make_plan(e, workspace, tasks, n, n, 1, 1, pl);
  if (pl.bytes > workspace_bytes)
    return fail(e, MI_ERR_WORKSPACE, "workspace too small: need " + std::to_string(pl.bytes) + " bytes");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nl = (int)e->L.size();
  const size_t pstride = theta_tasks == 1 ? 0 : e->P;
  { const int brc = plan_begin(e, st, pl); if (brc) return brc; }
  ActSet& A = pl.sup[0];
  LAUNCH(e, st, OP_MISC, 0, launch_nchw_to_nhwc(st, x, (size_t)tasks * n, e->d.in_channels, e->d.in_h, e->d.in_w, pl.xs));
  LAUNCH(e, st, OP_MISC, 1, launch_gather_params(st, theta, pstride, e->perm_dev, (int)e->P, (int)e->PS, tasks, pl.theta));
  LAUNCH(e, st, OP_MISC, 1, launch_gather_params(st, v, pstride, e->perm_dev, (int)e->P, (int)e->PS, tasks, pl.lam));
  int rc = trunk_forward(e, st, pl, A, pl.xs, n, tasks, pl.theta);
  if (rc) return rc;
  if (e->d.head_mean_pool) HIPCHK(e, launch_spatial_mean(st, A.p[nl - 1], A.f, tasks * n, e->head_hw, e->head_c));
  // primal backward with the given cotangent: the tangent sweep needs its BatchNorm sums (g) and layer cotangents (A.dp)
  HIPCHK(e, hipMemsetAsync(pl.g, 0, (size_t)tasks * e->PS * sizeof(float), st));
  HeadArgs ha{};
  ha.f = A.f;
  ha.wl = pl.theta + e->off_wl; ha.bl = pl.theta + e->off_bl; ha.pstride = e->PS;
  ha.dl = const_cast<float*>(dlogits);
  ha.dwl = pl.g + e->off_wl; ha.dbl = pl.g + e->off_bl; ha.gstride = e->PS;
  ha.df = A.df;
  ha.n = n; ha.feat = e->feat; ha.ways = e->d.ways;
  LAUNCH(e, st, OP_HEAD, 0, launch_head_grads(st, ha, tasks));
  if (e->d.head_mean_pool) HIPCHK(e, launch_spatial_mean_bwd(st, A.df, A.dp[nl - 1], tasks * n, e->head_hw, e->head_c));
  rc = trunk_backward(e, st, pl, A, pl.xs, n, tasks, pl.theta, pl.g);
  if (rc) return rc;
  rc = pass_hvp(e, st, pl, A, pl.xs, n, tasks, pl.theta, pl.g, pl.lam, pl.hv, nullptr, dlogits, logits_dot_out);
  if (rc) return rc;
  if (theta_tasks == 1) {
    LAUNCH(e, st, OP_MISC, 3, launch_scatter_sum(st, pl.hv, e->perm_dev, (int)e->P, (int)e->PS, tasks, grad_theta_out));
  } else {
    LAUNCH(e, st, OP_MISC, 3, launch_scatter_tasks(st, pl.hv, e->perm_dev, (int)e->P, (int)e->PS, tasks, grad_theta_out));
  }
  return MI_OK;
}

extern "C" int mi_adam_step(void* stream, float* theta, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                            int step, float lr, float beta1, float beta2, float eps, float grad_scale) {
  if (!theta || !grad || !exp_avg || !exp_avg_sq || step < 1) return fail(nullptr, MI_ERR_ARG, "bad adam arguments");
  hipError_t s = launch_adam(reinterpret_cast<hipStream_t>(stream), theta, grad, exp_avg, exp_avg_sq, n, step, lr, beta1,
                             beta2, eps, grad_scale);
  return s == hipSuccess ? MI_OK : fail(nullptr, MI_ERR_HIP, hipGetErrorString(s));
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-kernel entry points (unit parity tests).
#define HIPCHK0(call)                                                                     \
  do {                                                                                    \
    hipError_t _s = (call);                                                               \
    if (_s != hipSuccess) return fail(nullptr, MI_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_s)); \
  } while (0)

extern "C" size_t mi_kernel_scratch_bytes(int tasks, int n, int h, int w, int c) {
  // generous: covers bn partials and wgrad partials for one layer with <= 64 input channels
  ConvGeom g{n, h, w, h, w, 64, c, 1};
  ConvGeom g2{n, h, w, (h - 1) / 2 + 1, (w - 1) / 2 + 1, 64, c, 2};      // the stride-2 (Omniglot) weight gradient splits finer
  size_t bn = (size_t)tasks * (size_t)conv_max_blocks_per_task(g) * 2 * c * sizeof(double);
  size_t wg = wgrad_partial_floats(g, tasks) * sizeof(float);
  const size_t wg2 = wgrad_partial_floats(g2, tasks) * sizeof(float);
  if (wg2 > wg) wg = wg2;
  return align_up(bn, 256) + align_up(wg, 256) + 4096;
}

extern "C" int mi_prepare_batch(void* stream, const float* data, const int64_t* labels, int tasks, int n2, int c, int h, int w,
                                float* xs, float* xq, int32_t* ys, int32_t* yq) {
  if (n2 % 2) return fail(nullptr, MI_ERR_ARG, "task batch must hold 2*shots*ways rows");
  HIPCHK0(launch_prepare_batch(reinterpret_cast<hipStream_t>(stream), data, labels, tasks, n2, c, h, w, xs, xq, ys, yq));
  return MI_OK;
}

// Input Gram matrix of block 1 and BatchNorm statistics of conv1 from it (gram.hip); unit-test entry points.
extern "C" size_t mi_input_gram_scratch_bytes(int tasks, int n, int h, int ci) {
  return gram_partial_doubles(tasks, n, h, ci) * sizeof(double);
}
extern "C" int mi_input_gram(void* stream, const float* x, int tasks, int n, int h, int w, int ci, void* scratch,
                             size_t scratch_bytes, double* g_out) {
  if (!x || !scratch || !g_out || (ci != 1 && ci != 3)) return fail(nullptr, MI_ERR_ARG, "mi_input_gram: bad arguments (ci must be 1 or 3)");
  if (scratch_bytes < mi_input_gram_scratch_bytes(tasks, n, h, ci)) return fail(nullptr, MI_ERR_WORKSPACE, "mi_input_gram: scratch too small");
  HIPCHK0(launch_input_gram(reinterpret_cast<hipStream_t>(stream), x, tasks, n, h, w, ci, static_cast<double*>(scratch), g_out));
  return MI_OK;
}
extern "C" int mi_gram_bn_stats(void* stream, const double* g, int tasks, int ci, int co, const float* w9, size_t pstride,
                                const float* w9d, size_t vstride, int pixels, float* out0, float* out1, const float* mu,
                                const float* rstd) {
  if (!g || !w9 || !out0 || !out1 || (w9d && (!mu || !rstd))) return fail(nullptr, MI_ERR_ARG, "mi_gram_bn_stats: bad arguments");
  HIPCHK0(launch_gram_stats(reinterpret_cast<hipStream_t>(stream), g, tasks, ci, co, w9, pstride, w9d, vstride, 1.0 / (double)pixels,
                            w9d ? 1 : 0, out0, out1, mu, rstd));
  return MI_OK;
}

extern "C" int mi_stream_copy(void* stream, const void* src, void* dst, size_t bytes) {
  if (!src || !dst || bytes % 16) return fail(nullptr, MI_ERR_ARG, "mi_stream_copy: null pointer or size not a multiple of 16");
  HIPCHK0(launch_stream_copy(reinterpret_cast<hipStream_t>(stream), src, dst, bytes));
  return MI_OK;
}

extern "C" int mi_sample_tasks(void* stream, const void* dataset, int dataset_is_u8, size_t num_images, int c, int h, int w,
                               const int64_t* index, const uint8_t* rot, int tasks, int n2, float* data_out) {
  if (!dataset || !index || !data_out || tasks < 1 || n2 < 1 || num_images < 1) return fail(nullptr, MI_ERR_ARG, "bad sampler arguments");
  if (((size_t)c * h * w) % 4 != 0) return fail(nullptr, MI_ERR_ARG, "C*H*W must be a multiple of 4");
  if (rot && h != w) return fail(nullptr, MI_ERR_ARG, "quarter-turn rotations need square images");
  HIPCHK0(launch_sample_tasks(reinterpret_cast<hipStream_t>(stream), dataset, dataset_is_u8, index, rot, (size_t)tasks * n2, c, h, w, data_out));
  return MI_OK;
}

extern "C" int mi_conv3x3_bn_stats(void* stream, const float* x, const float* w9, size_t pstride, int tasks, int n, int h,
                                   int wd, int ci, int co, int stride, float* z, float* mu, float* rstd, void* scratch,
                                   size_t scratch_bytes) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  ConvArgs ca{};
  ca.in[0] = x; ca.wt[0] = w9; ca.wstride = pstride; ca.out = z;
  ca.partial = reinterpret_cast<double*>(scratch);
  ca.g = ConvGeom{n, h, wd, (h - 1) / stride + 1, (wd - 1) / stride + 1, ci, co, stride};
  ca.mpix = n * ca.g.ho * ca.g.wo;
  if ((size_t)tasks * conv_max_blocks_per_task(ca.g) * 2 * co * sizeof(double) > scratch_bytes)
    return fail(nullptr, MI_ERR_WORKSPACE, "scratch too small");
  hipError_t aerr = hipSuccess;      // (fp16 operand form: the input's largest magnitude per task, a reduction launch of its own here)
  ca.amax[0] = standalone_amax(st, 0, x, (size_t)n * h * wd * ci, tasks, &aerr);
  HIPCHK0(aerr);
  int blk = 0;
  HIPCHK0(launch_conv3x3(st, ca, tasks, 1, EPI_STATS, 0, &blk));
  HIPCHK0(launch_bn_finalize(st, ca.partial, blk, tasks, co, 1.0 / (double)ca.mpix, FIN_STATS, mu, co, rstd, co));
  return MI_OK;
}

extern "C" int mi_bn_relu_pool(void* stream, const float* z, const float* mu, const float* rstd, const float* gamma,
                               const float* beta, size_t pstride, int tasks, int n, int ho, int wo, int c, int pool, float* p) {
  BnArgs ba{};
  ba.z = z; ba.mu = mu; ba.rstd = rstd; ba.gamma = gamma; ba.beta = beta; ba.pstride = pstride; ba.out = p;
  ba.n = n; ba.ho = ho; ba.wo = wo; ba.c = c;
  HIPCHK0(launch_bn_fwd(reinterpret_cast<hipStream_t>(stream), ba, tasks, pool));
  return MI_OK;
}

extern "C" int mi_bn_relu_pool_bwd(void* stream, const float* z, const float* mu, const float* rstd, const float* gamma,
                                   const float* beta, size_t pstride, const float* dp, int tasks, int n, int ho, int wo, int c,
                                   int pool, float* dgamma, float* dbeta, size_t gstride, float* dz, void* scratch,
                                   size_t scratch_bytes) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  BnArgs ba{};
  ba.z = z; ba.mu = mu; ba.rstd = rstd; ba.gamma = gamma; ba.beta = beta; ba.pstride = pstride; ba.dp = dp;
  ba.partial = reinterpret_cast<double*>(scratch);
  ba.n = n; ba.ho = ho; ba.wo = wo; ba.c = c;
  ba.inv_m = 1.f / (float)(n * ho * wo);
  if ((size_t)tasks * bn_blocks_per_task(n, ho, wo, c, pool, tasks) * 2 * c * sizeof(double) > scratch_bytes)
    return fail(nullptr, MI_ERR_WORKSPACE, "scratch too small");
  int blk = 0;
  HIPCHK0(launch_bn_bwd_reduce(st, ba, tasks, pool, &blk));
  HIPCHK0(launch_bn_finalize(st, ba.partial, blk, tasks, c, 1.0, FIN_SUMS, dgamma, gstride, dbeta, gstride));
  ba.dgamma = dgamma; ba.dbeta = dbeta; ba.gstride = gstride; ba.out = dz;
  HIPCHK0(launch_bn_bwd_apply(st, ba, tasks, pool));
  return MI_OK;
}

extern "C" int mi_conv3x3_bwd(void* stream, const float* x, const float* dz, const float* w9, size_t pstride, int tasks, int n,
                              int h, int wd, int ci, int co, int stride, float* dx, float* dw9, size_t gstride, void* scratch,
                              size_t scratch_bytes) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int ho = (h - 1) / stride + 1, wo = (wd - 1) / stride + 1;
  WgradArgs wa{};
  wa.x[0] = x; wa.dz[0] = dz; wa.partial = reinterpret_cast<float*>(scratch);
  wa.g = ConvGeom{n, h, wd, ho, wo, ci, co, stride};
  wa.mpix = n * ho * wo;
  if (wgrad_partial_floats(wa.g, tasks) * sizeof(float) > scratch_bytes) return fail(nullptr, MI_ERR_WORKSPACE, "scratch too small");
  hipError_t aerr = hipSuccess;
  wa.amax_x[0] = standalone_amax(st, 0, x, (size_t)n * h * wd * ci, tasks, &aerr);
  HIPCHK0(aerr);
  wa.amax_dz[0] = standalone_amax(st, 1, dz, (size_t)n * ho * wo * co, tasks, &aerr);
  HIPCHK0(aerr);
  int nch = 0;
  HIPCHK0(launch_wgrad3x3(st, wa, tasks, 1, &nch));
  HIPCHK0(launch_wgrad_reduce(st, wa.partial, nch, 9 * ci * co, tasks, dw9, gstride));
  if (dx) {
    ConvArgs ca{};
    ca.in[0] = dz; ca.wt[0] = w9; ca.wstride = pstride; ca.out = dx;
    ca.amax[0] = wa.amax_dz[0];
    ca.g = ConvGeom{n, ho, wo, h, wd, co, ci, stride};
    ca.mpix = n * h * wd;
    HIPCHK0(launch_conv3x3(st, ca, tasks, 1, EPI_NONE, 1, nullptr));
  }
  return MI_OK;
}

extern "C" int mi_head_fwd_bwd(void* stream, const float* f, const float* wl, const float* bl, size_t pstride, const int32_t* y,
                               int tasks, int n, int feat, int ways, float* loss, float* acc, float* logits, float* prob,
                               float* dl, float* dwl, float* dbl, size_t gstride, float* df) {
  HeadArgs ha{};
  ha.f = f; ha.wl = wl; ha.bl = bl; ha.pstride = pstride; ha.y = y; ha.loss = loss; ha.acc = acc; ha.logits = logits;
  ha.prob = prob; ha.dl = dl; ha.dwl = dwl; ha.dbl = dbl; ha.gstride = gstride; ha.df = df;
  ha.n = n; ha.feat = feat; ha.ways = ways;
  if (!df || (size_t)feat < 2) return fail(nullptr, MI_ERR_ARG, "mi_head_fwd_bwd needs df (its first 2*tasks*n floats double as row scratch)");
  ha.rowloss = df; ha.rowhit = df + (size_t)tasks * n;   // consumed by the reduce launch before the gradient launch overwrites df
  // (the engine's own passes give the rows kernel separate scratch and let the gradient launch fold loss / acc; here the row
  // scratch aliases df, so the three launches stay separate)
  HIPCHK0(launch_head_fwd_bwd(reinterpret_cast<hipStream_t>(stream), ha, tasks, 0));
  if (dwl) {
    ha.loss = nullptr; ha.acc = nullptr;
    HIPCHK0(launch_head_grads(reinterpret_cast<hipStream_t>(stream), ha, tasks));
  }
  return MI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-launch profiling with HIP events on the caller's stream (used by bench.py for the roofline figures).
extern "C" int mi_profile_enable(mi_engine* e, int on, int kind_filter) {
  if (!e) return MI_ERR_ARG;
  e->prof_on = on;
  e->prof_filter = kind_filter;
  e->ev_used = 0;
  return MI_OK;
}
extern "C" int mi_profile_kinds(void) { return OP_COUNT * 8; }
extern "C" const char* mi_profile_op_name(int op) { return (op >= 0 && op < OP_COUNT) ? kOpNames[op] : "?"; }
// Sums elapsed ms / launch counts per kind (= op*8 + layer) since the last collect; blocks until the events completed.
extern "C" int mi_profile_collect(mi_engine* e, double* total_ms, int64_t* count, int n_kinds) {
  if (!e || !total_ms || !count || n_kinds < OP_COUNT * 8) return MI_ERR_ARG;
  for (int i = 0; i < n_kinds; ++i) { total_ms[i] = 0.0; count[i] = 0; }
  for (size_t i = 0; i < e->ev_used; ++i) {
    if (hipEventSynchronize(e->ev1[i]) != hipSuccess) return fail(e, MI_ERR_HIP, "hipEventSynchronize failed");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->ev0[i], e->ev1[i]) != hipSuccess) return fail(e, MI_ERR_HIP, "hipEventElapsedTime failed");
    total_ms[e->ev_kind[i]] += ms;
    count[e->ev_kind[i]] += 1;
  }
  e->ev_used = 0;
  return MI_OK;
}

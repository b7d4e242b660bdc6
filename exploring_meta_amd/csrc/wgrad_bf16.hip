// Stride-1 3x3 weight gradient of a 32 -> 32 channel block on the split-bf16 operand form (bf16_split.h): the counterpart of
// wgrad3x3_rows_mfma_kernel (conv_mfma.hip) for the implicit ATen conv2d-backward-weight launches behind ConvBlock.conv
// (reference core_functions/vision_models.py:177-185,189), every task of the meta-batch in one launch.
//
// dW[tap][ci][co] = sum over pixels of x[pixel + tap][ci] * dz[pixel][co]: M = ci, N = co, K = pixels.  v_mfma_f32_32x32x16_bf16
// takes K = 16 pixels per instruction: lane half h owns image row y = 2*yp + h, both halves the same 8 columns [8s, 8s + 8) -- a work
// unit is (image, row pair yp, column segment s).  Per unit a lane loads the 8 dz values of its output channel (B operand) and a
// 3 x 10 halo patch of x of its input channel (A operand): 38 coalesced dword loads (32 lanes = the 32 channels of one pixel) feed
// 54 MFMAs (9 taps x 6 plane products).  Every loaded value is split into its three bf16 planes ONCE (4.5 VALU instructions per
// value) and re-used by up to nine taps from registers: the horizontal displacement -1 / +1 is a choice of four of the row's five
// packed register pairs, 0 a v_alignbit_b32 of neighbouring pairs.  The VALU work of a row (57 instructions) is placed between the
// 18 MFMAs of the previous row, the next unit's loads fly under the current unit (two register sets).
// Accumulators (9 x 16) live in AGPRs (this file is built without -amdgpu-mfma-vgpr-form): one workgroup of four waves per CU.
// The four waves of a workgroup interleave units, reduce their accumulators through LDS tap by tap and leave ONE partial per
// workgroup; reduce_partials_kernel folds workgroups in a fixed order (deterministic, no atomics) -- as the fp32 kernel does.
#include "mi_common.h"
#include "kernels.h"
#include "bf16_split.h"

// Ablation builds (timing experiments, not shipped): -DMI_WGRAD_DBG=5 strip kernel with three of its six products per K step (the matrix
// work of a two-plane operand form, DESIGN.md 8c lead 5; wrong results), =1 no MFMAs, 2 no operand preparation, 3 no loads in the loop, 4 loads
// only; build to another file name and select it with MI_MAML_LIB (tools/wgrad_probe.py runs in one process either way).
#ifndef MI_WGRAD_DBG
#define MI_WGRAD_DBG 0
#endif

namespace {

struct WRaw { float xa[3][10]; float b[8]; };                   // one unit's loads: x rows y-1, y, y+1 x columns c0-1 .. c0+8; dz row y
struct RowPl { unsigned h[5], m[5], l[5]; };                    // a row's pairs (v0,v1) .. (v8,v9), three planes
struct OddPl { unsigned h[4], m[4], l[4]; };                    // the odd packing (v1,v2) .. (v7,v8): displacement 0
struct DzPl { unsigned h[4], m[4], l[4]; };

// The loads of a unit in four pieces (piece 0: its 8 dz values, pieces 1..3: the 10 x values of halo row piece - 1), so that they can
// be placed between the MFMAs of an earlier unit.  unit, and everything decoded from it, is wave-uniform (scalar unit); rows are per
// lane half (h).  unit < 0: every load out of range (zeros).
struct WUnitPos { int n, y0, c0; bool ok; };
__device__ __forceinline__ WUnitPos unit_pos(int unit, int hp2, int nseg) {
  WUnitPos p;
  p.ok = unit >= 0;
  const int uu = p.ok ? unit : 0;
  p.n = uu / (hp2 * nseg);
  const int rem = uu - p.n * hp2 * nseg;
  const int yp = rem / nseg;
  p.y0 = 2 * yp;
  p.c0 = (rem - yp * nseg) * 8;
  return p;
}
template <int PIECE, int C>   // C = filters: a pixel is C * 4 bytes; lane_x / lane_dz = this lane's channel inside it
__device__ __forceinline__ void load_piece(WRaw& u, const WUnitPos& p, mi_rsrc rx, mi_rsrc rdz, unsigned lane_x, unsigned lane_dz, int H, int W, int h) {
  const int y = p.y0 + h;
  if (PIECE == 0) {
    const bool rowok = p.ok && y < H;
    const unsigned mask = rowok ? 0u : MI_OOB;                                               // out of range as an OR: no select, no branch
    const unsigned dzoff = (lane_dz + (unsigned)(((p.n * H + y) * W + p.c0) * (C * 4))) | mask;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned col = (p.c0 + i) < W ? (unsigned)(i * C * 4) : MI_OOB;                  // scalar select
      u.b[i] = buf_ld(rdz, dzoff + col);
    }
  } else {
    const int r = PIECE - 1;
    const int iy = y + r - 1;
    const bool rok = p.ok && y < H && (unsigned)iy < (unsigned)H;
    const unsigned mask = rok ? 0u : MI_OOB;
    const unsigned xoff = (lane_x + (unsigned)(((p.n * H + iy) * W + p.c0) * (C * 4))) | mask;   // pixel c0 of the row (offsets < 2^30: launcher)
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      const unsigned col = (unsigned)(p.c0 + c - 1) < (unsigned)W ? (unsigned)((c - 1) * C * 4) : MI_OOB;   // scalar select (c = 0: one pixel back, only where c0 >= 1)
      u.xa[r][c] = buf_ld(rx, xoff + col);
    }
  }
}

template <int P>
__device__ __forceinline__ void split_x_pair(const float* v, RowPl& p) {
  bf16_split2(floatx2{v[2 * P], v[2 * P + 1]}, p.h[P], p.m[P], p.l[P]);
  asm volatile("" : "+v"(p.h[P]), "+v"(p.m[P]), "+v"(p.l[P]));       // computed HERE (instruction selection otherwise sinks the split to its use)
}
template <int P>
__device__ __forceinline__ void split_dz_pair(const float* v, DzPl& p) {
  bf16_split2(floatx2{v[2 * P], v[2 * P + 1]}, p.h[P], p.m[P], p.l[P]);
  asm volatile("" : "+v"(p.h[P]), "+v"(p.m[P]), "+v"(p.l[P]));
}
// F16: the two-plane fp16 form (bf16_split.h) -- planes h and l only, the value scaled by s (wave-uniform) on the way
template <int P, bool F16>
__device__ __forceinline__ void split_x(const float* v, RowPl& p, float s) {
  if constexpr (F16) {
    f16_split2(floatx2{v[2 * P], v[2 * P + 1]}, s, p.h[P], p.l[P]);
    asm volatile("" : "+v"(p.h[P]), "+v"(p.l[P]));
  } else {
    split_x_pair<P>(v, p);
  }
}
template <int P, bool F16>
__device__ __forceinline__ void split_dz(const float* v, DzPl& p, float s) {
  if constexpr (F16) {
    f16_split2(floatx2{v[2 * P], v[2 * P + 1]}, s, p.h[P], p.l[P]);
    asm volatile("" : "+v"(p.h[P]), "+v"(p.l[P]));
  } else {
    split_dz_pair<P>(v, p);
  }
}
// scales of the operands of every term (x, dz) and the factor that takes them out of the sums again; all wave-uniform
struct F16Scales { float sx[2], sd[2], inv; };
// (two steps, so that the cells are requested at the top of the kernel and waited for only after the first operand loads are out)
struct F16Cells { unsigned x[2], d[2]; };
__device__ __forceinline__ F16Cells f16_wgrad_cells(const WgradArgs& a, int task) {
  F16Cells c = {{0u, 0u}, {0u, 0u}};
  for (int t = 0; t < a.nterms; ++t) { c.x[t] = mi_cell_fetch(a.amax_x[t], task); c.d[t] = mi_cell_fetch(a.amax_dz[t], task); }
  return c;
}
__device__ __forceinline__ F16Scales f16_wgrad_scales(const WgradArgs& a, const F16Cells& c) {
  int kk[2][2] = {{0, 0}, {0, 0}};
  for (int t = 0; t < a.nterms; ++t) { kk[t][0] = f16_scale_exp(mi_cell_fold(c.x[t])); kk[t][1] = f16_scale_exp(mi_cell_fold(c.d[t])); }
  if (a.nterms == 2) f16_common_scale(kk);
  F16Scales r;
  for (int t = 0; t < 2; ++t) {
    r.sx[t] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(f16_pow2(kk[t][0]))));
    r.sd[t] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(f16_pow2(kk[t][1]))));
  }
  r.inv = f16_pow2(-(kk[0][0] + kk[0][1]));
  return r;
}
__device__ __forceinline__ void odd_plane(const unsigned* e, unsigned* o) {   // (v1,v2) = high half of (v0,v1) | low half of (v2,v3) ...
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = __builtin_amdgcn_alignbit(e[i + 1], e[i], 16);
  asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
}

}  // namespace

template <int C, bool F16>   // C = filters (32 or 64: blockIdx.z = (ci tile, co tile))
__global__ __launch_bounds__(256) void wgrad3x3_rows_bf16_kernel(WgradArgs a) {
  __shared__ float red[4 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: unit decode runs on the scalar unit
  const int j = lane & 31, h = lane >> 5;
  const int task = blockIdx.y;
  const int H = a.g.h, W = a.g.w;                              // stride 1, 32 -> 32 channels: conv output is H x W as well
  const int hp2 = (H + 1) >> 1, nseg = (W + 7) >> 3;
  const int nunits = a.g.n * hp2 * nseg;                       // per term; the unit stream is [term][unit]
  const int total = nunits * a.nterms;
  const int ub0 = blockIdx.x * a.chunk_pix;                    // chunk_pix = units per workgroup here
  const int ub1 = min(ub0 + a.chunk_pix, total);
  const size_t t_elems = (size_t)a.g.n * H * W * C;
  constexpr int NCT = C / 32;
  const int cit = blockIdx.z / NCT, cot = blockIdx.z - cit * NCT;
  const unsigned tb = (unsigned)(t_elems * 4);
  const mi_rsrc rx0 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x[0] + (size_t)task * t_elems), 0, tb, 0x00020000);
  const mi_rsrc rd0 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dz[0] + (size_t)task * t_elems), 0, tb, 0x00020000);
  const float* x1p = a.nterms > 1 ? a.x[1] : a.x[0];
  const float* d1p = a.nterms > 1 ? a.dz[1] : a.dz[0];
  const mi_rsrc rx1 = __builtin_amdgcn_make_buffer_rsrc((void*)(x1p + (size_t)task * t_elems), 0, tb, 0x00020000);
  const mi_rsrc rd1 = __builtin_amdgcn_make_buffer_rsrc((void*)(d1p + (size_t)task * t_elems), 0, tb, 0x00020000);
  const unsigned lane_x = (unsigned)(cit * 32 + j) * 4u, lane_dz = (unsigned)(cot * 32 + j) * 4u;
  F16Scales fs = {{1.f, 1.f}, {1.f, 1.f}, 1.f};
  F16Cells fcells = {{0u, 0u}, {0u, 0u}};
  if constexpr (F16) fcells = f16_wgrad_cells(a, task);

  floatx16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // unit v of this wave's stream: second term of the stream from nunits on (scalar selects, no branch); v >= ub1: nothing left (zeros)
  auto pos_of = [&](int v) { return unit_pos(v >= ub1 ? -1 : (v >= nunits ? v - nunits : v), hp2, nseg); };
#define WG_LOAD(PIECE, RAW, V, POS) load_piece<PIECE, C>(RAW, POS, (V) >= nunits ? rx1 : rx0, (V) >= nunits ? rd1 : rd0, lane_x, lane_dz, H, W, h)

  WRaw raw[3];                                                  // loads run two units ahead
  RowPl pe[2];
  OddPl po[2];
  DzPl pb[2];
  // six plane products of one tap, the next row's (or unit's) operand preparation spread between them
  // (F16: three products -- l b_h, h b_l, h b_h -- with the same preparation slots between them)
#define WG_MFMA(T, X, Y) if (MI_WGRAD_DBG != 1 && MI_WGRAD_DBG != 4) { if constexpr (F16) acc[T] = MI_F16_MFMA(X, Y, acc[T]); else acc[T] = MI_BF_MFMA(X, Y, acc[T]); }
#define WG_MFMA3(T, X, Y) if constexpr (!F16) { WG_MFMA(T, X, Y); }    /* the three products only the three-plane form has */
#define WG_TAP(T, AH, AM, AL, B, V0, V1, V2, V3)                                       \
  __builtin_amdgcn_sched_barrier(0);                                                   \
  WG_MFMA(T, AL, B.h);                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                   \
  if (MI_WGRAD_DBG != 2 && MI_WGRAD_DBG != 4) { V0; }                                                                \
  __builtin_amdgcn_sched_barrier(0);                                                   \
  WG_MFMA(T, AH, B.l);                                                            \
  __builtin_amdgcn_sched_barrier(0);                                                   \
  if (MI_WGRAD_DBG != 2 && MI_WGRAD_DBG != 4) { V1; }                                                                \
  __builtin_amdgcn_sched_barrier(0);                                                   \
  WG_MFMA3(T, AM, B.m);                                                           \
  __builtin_amdgcn_sched_barrier(0);                                                   \
  if (MI_WGRAD_DBG != 2 && MI_WGRAD_DBG != 4) { V2; }                                                                \
  __builtin_amdgcn_sched_barrier(0);                                                   \
  WG_MFMA3(T, AM, B.h);                                                           \
  __builtin_amdgcn_sched_barrier(0);                                                   \
  if (MI_WGRAD_DBG != 2 && MI_WGRAD_DBG != 4) { V3; }                                                                \
  __builtin_amdgcn_sched_barrier(0);                                                   \
  WG_MFMA3(T, AH, B.m);                                                           \
  WG_MFMA(T, AH, B.h);                                                            \
  __builtin_amdgcn_sched_barrier(0);
  // one row of taps (3r, 3r + 1, 3r + 2) from planes (E, O) against B; meanwhile row NR of raw set NRAW is prepared into (NE, NO) and,
  // where NR == 0 (the next unit), its dz into NB
  // (SXN / SDN: the scales of the unit whose rows / dz are being prepared -- the unit after this one where NR == 0)
#define WG_ROW(R, E, O, B, NRAW, NR, NE, NO, NB, FRAW, SXN, SDN)                                                               \
  WG_TAP(3 * (R) + 0, (E.h), (E.m), (E.l), B, (split_x<0, F16>(NRAW.xa[NR], NE, SXN)), (split_x<1, F16>(NRAW.xa[NR], NE, SXN)),        \
         (split_x<2, F16>(NRAW.xa[NR], NE, SXN)), (split_x<3, F16>(NRAW.xa[NR], NE, SXN)))                                             \
  WG_TAP(3 * (R) + 1, (O.h), (O.m), (O.l), B, (split_x<4, F16>(NRAW.xa[NR], NE, SXN)), odd_plane(NE.h, NO.h), if constexpr (!F16) odd_plane(NE.m, NO.m), \
         odd_plane(NE.l, NO.l))                                                                                          \
  WG_TAP(3 * (R) + 2, (E.h + 1), (E.m + 1), (E.l + 1), B,                                                                \
         if (NR == 0) (split_dz<0, F16>(NRAW.b, NB, SDN)), if (NR == 0) (split_dz<1, F16>(NRAW.b, NB, SDN)),                           \
         if (NR == 0) (split_dz<2, F16>(NRAW.b, NB, SDN)), if (NR == 0) (split_dz<3, F16>(NRAW.b, NB, SDN)))
  // one unit: raw set K % 3 (its row 0 and dz are already in planes, buffers K & 1), the next unit in raw set (K + 1) % 3
#define WG_UNIT(K)                                                                              \
  WG_ROW(0, pe[(K) & 1], po[(K) & 1], pb[(K) & 1], raw[(K) % 3], 1, pe[((K) + 1) & 1], po[((K) + 1) & 1], pb[(K) & 1], raw[((K) + 2) % 3], sx_cur, sd_cur)       \
  WG_ROW(1, pe[((K) + 1) & 1], po[((K) + 1) & 1], pb[(K) & 1], raw[(K) % 3], 2, pe[(K) & 1], po[(K) & 1], pb[(K) & 1], raw[((K) + 2) % 3], sx_cur, sd_cur)       \
  WG_ROW(2, pe[(K) & 1], po[(K) & 1], pb[(K) & 1], raw[((K) + 1) % 3], 0, pe[((K) + 1) & 1], po[((K) + 1) & 1], pb[((K) + 1) & 1], raw[((K) + 2) % 3], sx_nxt, sd_nxt)

  int u = ub0 + wave;
  {
    const WUnitPos p0 = pos_of(u), p1 = pos_of(u + 4);
    WG_LOAD(0, raw[0], u, p0); WG_LOAD(1, raw[0], u, p0); WG_LOAD(2, raw[0], u, p0); WG_LOAD(3, raw[0], u, p0);
    WG_LOAD(0, raw[1], u + 4, p1); WG_LOAD(1, raw[1], u + 4, p1); WG_LOAD(2, raw[1], u + 4, p1); WG_LOAD(3, raw[1], u + 4, p1);
  }
  if constexpr (F16) fs = f16_wgrad_scales(a, fcells);
  // prologue: row 0 and dz of the first unit
  {
    const float sx0 = u >= nunits ? fs.sx[1] : fs.sx[0], sd0 = u >= nunits ? fs.sd[1] : fs.sd[0];
    split_x<0, F16>(raw[0].xa[0], pe[0], sx0); split_x<1, F16>(raw[0].xa[0], pe[0], sx0); split_x<2, F16>(raw[0].xa[0], pe[0], sx0);
    split_x<3, F16>(raw[0].xa[0], pe[0], sx0); split_x<4, F16>(raw[0].xa[0], pe[0], sx0);
    odd_plane(pe[0].h, po[0].h); if constexpr (!F16) odd_plane(pe[0].m, po[0].m); odd_plane(pe[0].l, po[0].l);
    split_dz<0, F16>(raw[0].b, pb[0], sd0); split_dz<1, F16>(raw[0].b, pb[0], sd0); split_dz<2, F16>(raw[0].b, pb[0], sd0); split_dz<3, F16>(raw[0].b, pb[0], sd0);
  }
  // Six units per trip: the raw sets rotate with period three, the plane buffers with period two (a unit has three rows, so the row
  // planes of unit K start in buffer K & 1 and end there, the next unit's row 0 lands in (K + 1) & 1).  No exit and no skip inside
  // a trip (either makes the compiler keep a second copy of the 144 accumulator registers and move between them): a unit past the
  // wave's last one runs on zeros, and the launcher deals units to workgroups in multiples of 24 so that only a task's last
  // workgroup has any.
#define WG_STEP(K)                                                          \
  {                                                                         \
    const int fv = u + 4 * ((K) + 2);      /* the unit two ahead: its loads go out before this unit's MFMAs (between them they were slower) */ \
    const WUnitPos fpos = pos_of(fv);                                       \
    const bool t1c_ = u + 4 * (K) >= nunits, t1n_ = u + 4 * ((K) + 1) >= nunits;   /* second term: this unit / the next one */ \
    const float sx_cur = t1c_ ? fs.sx[1] : fs.sx[0], sd_cur = t1c_ ? fs.sd[1] : fs.sd[0];   \
    const float sx_nxt = t1n_ ? fs.sx[1] : fs.sx[0], sd_nxt = t1n_ ? fs.sd[1] : fs.sd[0];   \
    (void)sd_cur;                                                           \
    if (MI_WGRAD_DBG != 3) {                                                \
      WG_LOAD(0, raw[((K) + 2) % 3], fv, fpos); WG_LOAD(1, raw[((K) + 2) % 3], fv, fpos);   \
      WG_LOAD(2, raw[((K) + 2) % 3], fv, fpos); WG_LOAD(3, raw[((K) + 2) % 3], fv, fpos);   \
    }                                                                       \
    WG_UNIT(K)                                                              \
  }
  for (; u < ub1; u += 24) {
    WG_STEP(0) WG_STEP(1) WG_STEP(2) WG_STEP(3) WG_STEP(4) WG_STEP(5)
  }
#undef WG_STEP
#undef WG_LOAD
#undef WG_UNIT
#undef WG_ROW
#undef WG_TAP
#undef WG_MFMA

  // cross-wave reduction, one tap at a time: red[wave][r*64 + lane]
  float* pt = a.partial + ((size_t)task * gridDim.x + blockIdx.x) * 9 * C * C;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[tap][r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q;
      float v = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
      if constexpr (F16) v *= fs.inv;                          // the operands' scales out of the sums (a power of two: exact)
      const int r = e >> 6, l = e & 63;
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
      pt[((size_t)tap * C + cit * 32 + row) * C + cot * 32 + col] = v;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Strip form (maps at least 32 wide): a work item is a vertical strip of one image -- 16 columns (lane half h = columns [8h, 8h + 8) of
// it), a run of rows -- walked one output row per step with the x rows it needs RESIDENT as split planes: step y multiplies dz row y
// against x rows y - 1, y, y + 1 (9 taps, 54 MFMAs) and meanwhile loads and splits only the NEW x row y + 2 and dz row y + 1: 18 loads and
// 18 split values per 54 MFMAs where the unit form above needs 38 and 38.  Every row has its own buffer descriptor (base = the row's
// first pixel, size = the row): columns outside the image are out of range (zeros) and all loads take immediates -- no address arithmetic
// on the vector unit at all.  Row planes rotate through four register sets, raw rows through four (three steps of load lookahead): four
// steps per loop trip, strips are dealt in runs of a multiple of four rows.
struct StripItem { int n, c0, ya, yb; bool ok; };
struct XRowPl { RowPl e; };                                     // (the odd packing is rebuilt per step: 12 instructions instead of 12 resident registers per row)

// LDSR (round 6; 32 filters, bf16 form): the rows reach the lanes through a per-wave LDS ring filled by LDS-DMA (buffer_load ... lds: no register
// destination, so nothing limits how far ahead they run but the ring): one "bundle" per step -- x row r + 1 as two 1-KB pieces and one 256-B piece
// (18 pixels x 128 B), dz row r as two 1-KB pieces -- issued SEVEN steps ahead into 8 + 8 slots (34 KB per wave), and read back as the same 18 dwords per
// lane (ds_read_b32, conflict-free: a pixel's 32 channels are 32 consecutive words) at the places the register loads used to be.  A wave's own DMA is
// ordered for its own ds_read by a COUNTED s_waitcnt vmcnt (loads retire in order; hipcc does not track LDS-DMA, so every count here is by hand: 5
// operations per bundle, the tables next to the macros) -- never by vmcnt(0), which is what made the register ring stall once per loop trip (hipcc
// turns the first operand wait behind a back edge into vmcnt(0)).  Out-of-range pieces (columns outside the image: offsets past the row descriptor's
// size; rows outside it or past the strip piece: empty descriptors) arrive as ZEROS in LDS (tools/lds_dma_probe.hip), exactly as they did in registers.
#define MI_WG_RING 8
#define MI_WG_XSLOT 576                                          /* floats: 18 pixels x 32 channels */
#define MI_WG_DSLOT 512                                          /* floats: 16 pixels x 32 channels */
__device__ __forceinline__ void wg_dma16(mi_rsrc r, float* lds, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ void wg_dma4(mi_rsrc r, float* lds, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 4, voff, 0, 0, 0);
}

template <int C, bool F16, bool LDSR = false>   // C = filters (32 or 64: blockIdx.z = (ci tile, co tile))
__global__ __launch_bounds__(256) void wgrad3x3_strip_bf16_kernel(WgradArgs a) {
  static_assert(!LDSR || (C == 32 && !F16), "the LDS-ring form serves the 32-filter split-bf16 kernel");
  __shared__ float red[4 * 1024];
  extern __shared__ __attribute__((aligned(16))) float wg_ring[];        // LDSR: [wave][8 x slots | 8 dz slots]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int task = blockIdx.y;
  const int H = a.g.h, W = a.g.w;
  const int nseg = (W + 15) >> 4;
  const int rp = a.mpix;                                       // rows per strip piece (multiple of 4; the launcher passes it here)
  const int nh = (H + rp - 1) / rp;
  const int nitems = a.g.n * nseg * nh;                        // per term; the item stream is [term][item]
  const int total = nitems * a.nterms;
  const int ub0 = blockIdx.x * a.chunk_pix;                    // chunk_pix = items per workgroup here
  const int ub1 = min(ub0 + a.chunk_pix, total);
  const size_t t_elems = (size_t)a.g.n * H * W * C;
  const unsigned rowb = (unsigned)(W * C * 4);                 // bytes of one image row
  constexpr int NCT = C / 32;
  const int cit = blockIdx.z / NCT, cot = blockIdx.z - cit * NCT;
  F16Scales fs = {{1.f, 1.f}, {1.f, 1.f}, 1.f};
  F16Cells fcells = {{0u, 0u}, {0u, 0u}};
  if constexpr (F16) fcells = f16_wgrad_cells(a, task);

  floatx16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  XRowPl xr[4];                                                // x row planes, slot = (row - first row + 1) & 3
  OddPl ot;                                                    // odd packing of the row whose taps are running
  DzPl dzp[2];
  float rawx[4][10], rawd[4][8];

  for (int item = ub0 + wave; item < ub1; item += 4) {
    const int term = item >= nitems ? 1 : 0;
    const int it = item - term * nitems;

    const int n = it / (nseg * nh);
    const int rem = it - n * nseg * nh;
    const int s16 = rem / nh, half = rem - s16 * nh;
    const int ya = half * rp, yb = min(H, ya + rp);
    const int cb = s16 * 16 + 8 * h;                           // this lane half's first column
    const float* xt = (term ? a.x[1] : a.x[0]) + (size_t)task * t_elems + (size_t)n * H * W * C;
    const float* dt = (term ? a.dz[1] : a.dz[0]) + (size_t)task * t_elems + (size_t)n * H * W * C;
    // per-lane offsets inside a row: column cb (x columns cb .. cb + 8 and all dz columns as immediates), and column cb - 1 on its own
    const unsigned vo = (unsigned)((cb * C + cit * 32 + j) * 4), vod = (unsigned)((cb * C + cot * 32 + j) * 4);
    const unsigned vom = cb >= 1 ? vo - (unsigned)(C * 4) : MI_OOB;
    auto xrow = [&](int yy) {                                    // descriptor of x row yy (empty outside the image)
      const bool ok = (unsigned)yy < (unsigned)H;
      return __builtin_amdgcn_make_buffer_rsrc((void*)(xt + (size_t)(ok ? yy : 0) * W * C), 0, ok ? rowb : 0u, 0x00020000);
    };
    auto drow = [&](int yy) {                                    // dz row yy; rows past the strip piece contribute nothing
      const bool ok = yy < yb;
      return __builtin_amdgcn_make_buffer_rsrc((void*)(dt + (size_t)(ok ? yy : 0) * W * C), 0, ok ? rowb : 0u, 0x00020000);
    };
    // LDSR: this wave's ring, the lanes' source offsets of the five pieces of a bundle (x: segment = columns s16 * 16 - 1 .. + 16; a negative
    // offset wraps past the descriptor's size, i.e. reads zeros: the left padding) and this lane's word inside a slot
    float* const xring = wg_ring + wave * (MI_WG_RING * (MI_WG_XSLOT + MI_WG_DSLOT));
    float* const dring = xring + MI_WG_RING * MI_WG_XSLOT;
    const unsigned xseg = (unsigned)((s16 * 16 - 1) * (C * 4));
    const unsigned vx0 = xseg + (unsigned)(lane * 16), vx1 = xseg + 1024u + (unsigned)(lane * 16), vx2 = xseg + 2048u + (unsigned)(lane * 4);
    const unsigned vd0 = (unsigned)(s16 * 16 * (C * 4) + lane * 16), vd1 = vd0 + 1024u;
    const int lane_w = (8 * h) * 32 + j;
#define WG_XISSUE(R) { const mi_rsrc rr_ = xrow(R); float* s_ = xring + (((R) - ya + 1) & (MI_WG_RING - 1)) * MI_WG_XSLOT;        \
      wg_dma16(rr_, s_, vx0); wg_dma16(rr_, s_ + 256, vx1); wg_dma4(rr_, s_ + 512, vx2); }
#define WG_DISSUE(R) { const mi_rsrc rr_ = drow(R); float* s_ = dring + (((R) - ya) & (MI_WG_RING - 1)) * MI_WG_DSLOT;            \
      wg_dma16(rr_, s_, vd0); wg_dma16(rr_, s_ + 256, vd1); }
#define WG_BUNDLE(R) { if constexpr (LDSR) { WG_XISSUE((R) + 1) WG_DISSUE(R) } }                 /* 5 operations */
#define WG_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");
    // ST_LOADX / ST_LOADD(SET, YY, N): row YY into raw set SET.  Register form: the loads themselves.  LDSR: wait until at most N younger DMA
    // operations are outstanding (= row YY has landed), then the lane's dwords from the ring.
#define ST_LOADX(SET, YY, N) { if constexpr (LDSR) { WG_WAIT(N)                                                              \
        const float* p_ = xring + (((YY) - ya + 1) & (MI_WG_RING - 1)) * MI_WG_XSLOT + lane_w;                                     \
        _Pragma("unroll") for (int c = 0; c < 10; ++c) rawx[SET][c] = p_[c * 32];                                                 \
      } else { const mi_rsrc rr_ = xrow(YY); rawx[SET][0] = buf_ld(rr_, vom);                                                      \
        _Pragma("unroll") for (int c = 1; c < 10; ++c) rawx[SET][c] = buf_ld(rr_, vo + (unsigned)((c - 1) * C * 4)); } }
#define ST_LOADD(SET, YY, N) { if constexpr (LDSR) { WG_WAIT(N)                                                              \
        const float* p_ = dring + (((YY) - ya) & (MI_WG_RING - 1)) * MI_WG_DSLOT + lane_w;                                         \
        _Pragma("unroll") for (int c = 0; c < 8; ++c) rawd[SET][c] = p_[c * 32];                                                  \
      } else { const mi_rsrc rr_ = drow(YY);                                                                                        \
        _Pragma("unroll") for (int c = 0; c < 8; ++c) rawd[SET][c] = buf_ld(rr_, vod + (unsigned)(c * C * 4)); } }
#define ST_SPLITX(SET, SLOT) { split_x<0, F16>(rawx[SET], xr[SLOT].e, sxi); split_x<1, F16>(rawx[SET], xr[SLOT].e, sxi); split_x<2, F16>(rawx[SET], xr[SLOT].e, sxi); \
      split_x<3, F16>(rawx[SET], xr[SLOT].e, sxi); split_x<4, F16>(rawx[SET], xr[SLOT].e, sxi); }
#define ST_SPLITD(SET, BUF) { split_dz<0, F16>(rawd[SET], dzp[BUF], sdi); split_dz<1, F16>(rawd[SET], dzp[BUF], sdi); split_dz<2, F16>(rawd[SET], dzp[BUF], sdi); \
      split_dz<3, F16>(rawd[SET], dzp[BUF], sdi); }
    // prologue: x rows ya - 1, ya, ya + 1 (slots 0, 1, 2) and dz row ya; then the ring: x rows ya + 2 .. ya + 4, dz rows ya + 1 .. ya + 3
    // LDSR: the previous item's DMA drained, then x rows ya - 1, ya and bundles ya .. ya + 5 on their way (36 operations: the ring's 8 x slots
    // full); counts below = operations issued AFTER the awaited row's (bundle r = x row r + 1 [3], dz row r [2])
    if constexpr (LDSR) {
      WG_WAIT(0)
      WG_XISSUE(ya - 1) WG_XISSUE(ya)
      WG_BUNDLE(ya) WG_BUNDLE(ya + 1) WG_BUNDLE(ya + 2) WG_BUNDLE(ya + 3) WG_BUNDLE(ya + 4) WG_BUNDLE(ya + 5)
    }
    ST_LOADX(0, ya - 1, 33) ST_LOADX(1, ya, 30) ST_LOADX(2, ya + 1, 27) ST_LOADD(0, ya, 25)
    if constexpr (F16) fs = f16_wgrad_scales(a, fcells);          // (scalar arithmetic on the cells requested at the top of the kernel)
    const float sxi = term ? fs.sx[1] : fs.sx[0], sdi = term ? fs.sd[1] : fs.sd[0];     // (F16) this item's operand scales
    ST_SPLITX(0, 0) ST_SPLITX(1, 1) ST_SPLITX(2, 2) ST_SPLITD(0, 0)
    // The ring.  hipcc turns the FIRST operand wait of every loop trip into vmcnt(0) whatever is in flight across the back edge (its
    // wait-count pass does not carry exact counts around a loop), so the loads are timed such that everything in flight at a trip
    // boundary is at least two steps old: a trip's steps 2 and 3 are loaded at its step 0, the NEXT trip's steps 0 and 1 at its step 2.
    if constexpr (LDSR) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }      // (x row ya - 1 has left its slot: bundle ya + 6 takes it)
    WG_BUNDLE(ya + 6)
    ST_LOADX(3, ya + 2, 27) ST_LOADD(1, ya + 1, 25) ST_LOADX(0, ya + 3, 22) ST_LOADD(2, ya + 2, 20)
    // step T (output row y + T of the trip that starts at row y): x rows in slots T, T+1, T+2 (mod 4), dz planes T & 1; meanwhile x row
    // y + T + 2 (raw set (T + 3) & 3) is split into slot (T + 3) & 3 and dz row y + T + 1 (raw set (T + 1) & 3) into planes (T + 1) & 1
  // (F16: three products -- l b_h, h b_l, h b_h -- with the same preparation slots between them)
#define ST_MFMA(T, X, Y) if (MI_WGRAD_DBG != 1 && MI_WGRAD_DBG != 4) { if constexpr (F16) acc[T] = MI_F16_MFMA(X, Y, acc[T]); else acc[T] = MI_BF_MFMA(X, Y, acc[T]); }
#define ST_MFMA_LOW(T, X, Y) if (MI_WGRAD_DBG != 5) ST_MFMA(T, X, Y)      /* (ablation build: the bf16 form without three of its six products) */
#define ST_MFMA3(T, X, Y) if constexpr (!F16) { ST_MFMA(T, X, Y); }          /* the three products only the three-plane form has */
#define ST_TAP(T, AH, AM, AL, B, V0, V1, V2, V3)                                       \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    if constexpr (F16) { ST_MFMA(T, AL, B.h); } else { ST_MFMA_LOW(T, AL, B.h); }   \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    V0;                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    if constexpr (F16) { ST_MFMA(T, AH, B.l); } else { ST_MFMA_LOW(T, AH, B.l); }   \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    V1;                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    if constexpr (!F16) { ST_MFMA_LOW(T, AM, B.m); }                              \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    V2;                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    ST_MFMA3(T, AM, B.h);                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    V3;                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    ST_MFMA3(T, AH, B.m);                                                         \
    ST_MFMA(T, AH, B.h);                                                          \
    __builtin_amdgcn_sched_barrier(0);
#define ST_ROW(R, X, B, V3, V4, V5, V6, V7, V8, V9, V10, V11)                                                     \
    ST_TAP(3 * (R) + 0, (X.e.h), (X.e.m), (X.e.l), B, ST_PREP(odd_plane(X.e.h, ot.h)), ST_PREP(if constexpr (!F16) odd_plane(X.e.m, ot.m)), ST_PREP(odd_plane(X.e.l, ot.l)), V3)   \
    ST_TAP(3 * (R) + 1, (ot.h), (ot.m), (ot.l), B, V4, V5, V6, V7)                                                \
    ST_TAP(3 * (R) + 2, (X.e.h + 1), (X.e.m + 1), (X.e.l + 1), B, V8, V9, V10, V11)
#define ST_PREP(X) if (MI_WGRAD_DBG != 2 && MI_WGRAD_DBG != 4) { X; }
    // LDSR: step T of the trip that starts at row y puts bundle y + T + 7 on its way (x row y + T + 8 into the slot of row y + T, read a trip ago); the
    // reads of steps 0 / 2 then find: x row y + T + 4 with 22 younger operations (its own bundle's dz [2] + four bundles), dz row y + T + 3 with
    // 20, x row y + T + 5 with 17, dz row y + T + 4 with 15
#define ST_STEP(T)                                                                                                \
    {                                                                                                             \
      WG_BUNDLE(y + (T) + 7)                                                                                      \
      XRowPl& nx = xr[((T) + 3) & 3];                                                                             \
      const float* rx_ = rawx[((T) + 3) & 3];                                                                     \
      const float* rd_ = rawd[((T) + 1) & 3];                                                                     \
      DzPl& nd = dzp[((T) + 1) & 1];                                                                              \
      const DzPl& cd = dzp[(T) & 1];                                                                              \
      ST_ROW(0, xr[(T) & 3], cd, ST_PREP((split_x<0, F16>(rx_, nx.e, sxi))), ST_PREP((split_x<1, F16>(rx_, nx.e, sxi))), ST_PREP((split_x<2, F16>(rx_, nx.e, sxi))),      \
             ST_PREP((split_x<3, F16>(rx_, nx.e, sxi))), ST_PREP((split_x<4, F16>(rx_, nx.e, sxi))), ST_PREP((split_dz<0, F16>(rd_, nd, sdi))), ST_PREP((split_dz<1, F16>(rd_, nd, sdi))), \
             ST_PREP((split_dz<2, F16>(rd_, nd, sdi))), ST_PREP((split_dz<3, F16>(rd_, nd, sdi))))                                \
      ST_ROW(1, xr[((T) + 1) & 3], cd,                                                                            \
             if (MI_WGRAD_DBG == 3) {} else if ((T) == 0) ST_LOADX(1, y + 4, 22) else if ((T) == 2) ST_LOADX(3, y + 6, 22),          \
             if (MI_WGRAD_DBG == 3) {} else if ((T) == 0) ST_LOADD(3, y + 3, 20) else if ((T) == 2) ST_LOADD(1, y + 5, 20),          \
             (void)0, (void)0,                                                                                    \
             if (MI_WGRAD_DBG == 3) {} else if ((T) == 0) ST_LOADX(2, y + 5, 17) else if ((T) == 2) ST_LOADX(0, y + 7, 17),          \
             if (MI_WGRAD_DBG == 3) {} else if ((T) == 0) ST_LOADD(0, y + 4, 15) else if ((T) == 2) ST_LOADD(2, y + 6, 15),          \
             (void)0, (void)0, (void)0)                                                                           \
      ST_ROW(2, xr[((T) + 2) & 3], cd, (void)0, (void)0, (void)0, (void)0, (void)0, (void)0, (void)0, (void)0, (void)0) \
    }
    for (int y = ya; y < yb; y += 4) {
      ST_STEP(0) ST_STEP(1) ST_STEP(2) ST_STEP(3)
    }
#undef ST_STEP
#undef ST_PREP
#undef ST_ROW
#undef ST_TAP
#undef ST_MFMA3
#undef ST_MFMA_LOW
#undef ST_MFMA
#undef ST_SPLITD
#undef ST_SPLITX
#undef ST_LOADD
#undef ST_LOADX
#undef WG_WAIT
#undef WG_BUNDLE
#undef WG_DISSUE
#undef WG_XISSUE
  }
  if constexpr (LDSR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the last item's run-ahead pieces land before LDS is re-used / the wave ends)
  if constexpr (F16) fs = f16_wgrad_scales(a, fcells);           // (a wave without items has not formed them yet; every thread needs fs.inv below)

  // cross-wave reduction, one tap at a time: red[wave][r*64 + lane]
  float* pt = a.partial + ((size_t)task * gridDim.x + blockIdx.x) * 9 * C * C;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[tap][r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q;
      float v = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
      if constexpr (F16) v *= fs.inv;                          // the operands' scales out of the sums (a power of two: exact)
      const int r = e >> 6, l = e & 63;
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
      pt[((size_t)tap * C + cit * 32 + row) * C + cot * 32 + col] = v;
    }
    __syncthreads();
  }
}

// strip form: 32 or 64 filters, maps at least 32 wide whose width wastes at most 15 % of whole 16-column strips (42 -> 48)
bool wgrad_bf16_strips(const ConvGeom& g) {
  return g.stride == 1 && g.ci == g.co && (g.ci == 32 || g.ci == 64) && g.h == g.ho && g.w == g.wo && g.w >= 32 &&
         ((g.w + 15) / 16) * 16 * 100 <= g.w * 115;
}
// Items per task (per term) with pieces of `rows` rows (a multiple of four; the launcher picks it: 24 where the launch has items for every
// wave of the chip, shorter pieces for few-image launches, whose time is otherwise that of ONE 24-row item per wave however little work
// there is).  Pieces are evened out (42 rows as 24 + 18, not 24 + 24).
int wgrad_bf16_strip_rows(const ConvGeom& g, int rows) {
  const int nh = (g.h + rows - 1) / rows;
  return (((g.h + nh - 1) / nh + 3) / 4) * 4;
}
int wgrad_bf16_strip_items(const ConvGeom& g, int rows) {
  const int rp = wgrad_bf16_strip_rows(g, rows);
  return g.n * ((g.w + 15) / 16) * ((g.h + rp - 1) / rp);
}
// the rows-through-LDS form of the strip kernel (32 filters, bf16 form): OPT-IN (MI_WGRAD_LDS=1).  Bit-identical results (same operands, same
// products, same order: the 100 weight-gradient kernel tests pass with it), but no faster: block 2 of cfg2 in isolation 197.8 us against 185.0 us
// for the register ring (tools/wgrad_probe.py, one box), 16.12 / 16.14 against 16.14 / 16.12 ms per cfg2 iteration (profiles/r6/wgrad_lds_ring_ab.txt):
// the rows cost the same whichever way they arrive, so the register ring was never waiting for latency a deeper run-ahead could hide -- the
// launch runs at the socket power cap (and, beside the dgrad on the other stream, at the memory's bandwidth), where bytes moved are clock lost.
static bool wgrad_lds_rows() { static const bool v = getenv("MI_WGRAD_LDS") && atoi(getenv("MI_WGRAD_LDS")) != 0; return v; }
hipError_t launch_wgrad_strips_bf16(hipStream_t st, WgradArgs a, dim3 grid, int rows) {
  a.mpix = wgrad_bf16_strip_rows(a.g, rows);                   // (the kernel takes the rows per piece in this field)
  if (a.form != 2 && a.g.ci == 32 && wgrad_lds_rows()) {
    const size_t lds = (size_t)4 * MI_WG_RING * (MI_WG_XSLOT + MI_WG_DSLOT) * sizeof(float);
    auto k = wgrad3x3_strip_bf16_kernel<32, false, true>;
    static unsigned attr_done = 0;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (!(attr_done & (1u << (dev & 31)))) {
      if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e != hipSuccess) return e;
      attr_done |= 1u << (dev & 31);
    }
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
    return hipGetLastError();
  }
  if (a.form == 2) {
    if (a.g.ci == 64) hipLaunchKernelGGL((wgrad3x3_strip_bf16_kernel<64, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wgrad3x3_strip_bf16_kernel<32, true>), grid, dim3(256), 0, st, a);
  } else {
    if (a.g.ci == 64) hipLaunchKernelGGL((wgrad3x3_strip_bf16_kernel<64, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wgrad3x3_strip_bf16_kernel<32, false>), grid, dim3(256), 0, st, a);
  }
  return hipGetLastError();
}

// units per task for geometry g (per term) and whether this kernel takes it (32 -> 32 channels, stride 1, tensors addressable in 30 bits)
bool wgrad_bf16_ok(const ConvGeom& g) {
  // (w >= 16: on 10 x 10 maps -- two column segments, 20 % of them padding -- the fp32 kernel with its exact segments is faster, 27 vs 30 us)
  return g.stride == 1 && g.ci == g.co && (g.ci == 32 || g.ci == 64) && g.h == g.ho && g.w == g.wo && g.w >= 16 &&
         (size_t)g.n * g.h * g.w * g.ci * 4 < (size_t)MI_OOB - 4096;
}
int wgrad_bf16_units(const ConvGeom& g) { return g.n * ((g.h + 1) / 2) * ((g.w + 7) / 8); }

hipError_t launch_wgrad_rows_bf16(hipStream_t st, const WgradArgs& a, dim3 grid) {
  if (a.form == 2) {
    if (a.g.ci == 64) hipLaunchKernelGGL((wgrad3x3_rows_bf16_kernel<64, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wgrad3x3_rows_bf16_kernel<32, true>), grid, dim3(256), 0, st, a);
  } else {
    if (a.g.ci == 64) hipLaunchKernelGGL((wgrad3x3_rows_bf16_kernel<64, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wgrad3x3_rows_bf16_kernel<32, false>), grid, dim3(256), 0, st, a);
  }
  return hipGetLastError();
}

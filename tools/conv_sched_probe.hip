// Probe (round 5): how should the K loop of the split-bf16 stride-1 convolution (conv_mfma.hip, conv3x3_s1_mfma_kernel<32, NTERMS, .., BF>)
// be SCHEDULED so that its vector work (operand split, lane shifts) runs beside its matrix work instead of before / after it?
// The round-4 counters say the shipped kernel serialises the two (317.6k matrix + 193.8k vector issue cycles per SIMD ~ the 536k-cycle launch).
// This file runs the kernel's own half-group -- 2 x 16-B row loads three half-groups ahead, the split of eight values into three bf16 planes,
// two lane-shifted copies, nine 16-B weight-plane reads from LDS, eighteen v_mfma_f32_32x32x16_bf16 -- on a streamed tensor of cfg2's
// block-2 geometry (800 images x 42 x 42 x 32 channels, tiles of 30 pixels) in several layouts, one 8-wave workgroup per CU (two waves per
// SIMD, as the two-term kernel runs), and prints shader-clock cycles and nanoseconds per MFMA and SIMD for each:
//
//   LAYOUT 0  shipped order: vector work in bursts of 6 / 6 / 13 / 13 between the MFMAs of a unit (MI_UNIT)
//   LAYOUT 1  bunched: a half-group's vector work (split of the NEXT half-group, its shifted copies) first, then its 18 MFMAs back to back
//   LAYOUT 2  bunched + s_setprio 1 around the MFMA run
//   LAYOUT 3  bunched + ping-pong: the two waves of a SIMD alternate between the vector phase and the matrix phase (s_barrier)
//   LAYOUT 4  the MFMAs alone (no vector work; loads and LDS reads kept)      LAYOUT 5  the vector work alone
//   LAYOUT 6  vector work spread evenly: 4 instructions behind every MFMA
//   NACC 3    the horizontal taps as three ACCUMULATORS fed from the unshifted planes (no lane shifts in the loop; the displacement is applied
//             to the sums once per tile): 48 instead of 72 vector instructions per 18 MFMAs
//   FLAGS 1   the lane shifts as plain v_and_b32 (no DPP): isolates the cross-lane path      FLAGS 2   loads from one resident line (no HBM stream)
//   fma sweep: NV independent v_fma_f32 behind each MFMA of a dependent chain, one and two waves per SIMD -- the issue model
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -Iexploring_meta_amd/csrc -Iinclude \
//         tools/conv_sched_probe.hip -o build/conv_sched_probe && build/conv_sched_probe
#include "bf16_split.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

struct Bf16Planes { unsigned h[4], m[4], l[4]; };
template <int P>
__device__ __forceinline__ void split_pair(const floatx4& x, Bf16Planes& p) {
  bf16_split2(floatx2{x[(P & 1) * 2], x[(P & 1) * 2 + 1]}, p.h[P], p.m[P], p.l[P]);
  asm volatile("" : "+v"(p.h[P]), "+v"(p.m[P]), "+v"(p.l[P]));
}

#define SB __builtin_amdgcn_sched_barrier(0)
// SHAPE 0: one v_mfma_f32_32x32x16_bf16; SHAPE 1: the same FLOPs, operand registers and accumulator registers as TWO v_mfma_f32_16x16x32_bf16
// (MI355X_MICROARCH.md, DVFS give-back item 7: the chip holds a higher clock on the 16x16x32 shape; results are not meaningful here)
template <int SHAPE> struct AccT;
template <> struct AccT<0> { floatx16 v; __device__ void zero() { for (int i = 0; i < 16; ++i) v[i] = 0.f; } __device__ float sum() const { float s = 0.f; for (int i = 0; i < 16; ++i) s += v[i]; return s; } };
template <> struct AccT<1> { floatx4 q[4]; __device__ void zero() { for (int i = 0; i < 16; ++i) q[i >> 2][i & 3] = 0.f; } __device__ float sum() const { float s = 0.f; for (int i = 0; i < 16; ++i) s += q[i >> 2][i & 3]; return s; } };
template <int SHAPE> __device__ __forceinline__ void mf(const unsigned (&a)[4], const mi_u32x4& b, AccT<SHAPE>& c, int slot) {
  if constexpr (SHAPE == 0) {
    c.v = MI_BF_MFMA(a, b, c.v);
  } else {
    const int s0 = (slot & 1) * 2;
    c.q[s0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(MI_BF8(a), MI_BF8(b), c.q[s0], 0, 0, 0);
    c.q[s0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(MI_BF8(a), MI_BF8(b), c.q[s0 + 1], 0, 0, 0);
  }
}
#define MF(a, b, c) mf<SHAPE>(a, b, c, __COUNTER__)
// the six products of a unit (small terms first, as the kernel orders them)
#define SIX(ca, cb, c, V1, V2, V3, V4) \
  SB; MF(ca.l, cb[0], c); SB; V1; SB; MF(ca.h, cb[2], c); SB; V2; SB; MF(ca.m, cb[1], c); SB; V3; SB; MF(ca.m, cb[0], c); SB; V4; SB; MF(ca.h, cb[1], c); MF(ca.h, cb[0], c); SB;

template <int LAYOUT, int NACC, int FLAGS, int SHAPE = 0>
__global__ __launch_bounds__(512, 1) void probe(const float* __restrict__ stream, unsigned stream_bytes, const mi_u32x4* __restrict__ wsrc,
                                                float* __restrict__ out, unsigned long long* __restrict__ stamps, int tiles_per_wave) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  mi_u32x4* lw = reinterpret_cast<mi_u32x4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  constexpr int NU = 18 * 2;                                       // weight units of one term: (9 taps) x (k half) x 3 planes x 64 lanes x 16 B = 54 KB
  for (int i = tid; i < NU * 3 * 64; i += (int)blockDim.x) lw[i] = wsrc[i];
  __syncthreads();
  const mi_u32x4* l4 = lw + lane;
  const mi_rsrc rin = __builtin_amdgcn_make_buffer_rsrc((void*)stream, 0, stream_bytes, 0x00020000);
  constexpr int W = 42, CI = 32, NH = 6, HRING = 3;
  const int NW = (int)(blockDim.x >> 6);
  const int wci = W * CI * 4;
  const unsigned lane_in = (unsigned)(h * 64);
  const int tile_base = blockIdx.x * NW * tiles_per_wave;
  auto tile_off = [&](int tl) { return (FLAGS & 2) ? lane_in + (unsigned)(j * 128) + (unsigned)wci : (unsigned)((tl * 30 + j) * (CI * 4)) + lane_in + (unsigned)wci; };
  floatx4 rawc[HRING][2], rawd[HRING][2];                        // (rawd, pd, accd: the second tile of LAYOUT 7)
  auto issue_hg = [&](unsigned base, int i) {
    const int ddy = (i % 6) / 2 - 1;
    const unsigned o = base + (unsigned)(ddy * wci) + (unsigned)((i & 1) * 32);
    rawc[i % HRING][0] = buf_ld16(rin, o);
    rawc[i % HRING][1] = buf_ld16(rin, o + 16);
  };
  auto unit_of = [](int i, int ddx) { return ((((i % 6) / 2) * 3 + (ddx + 1)) * 2 + (i & 1)); };   // 0..17
  int tile = tile_base + wave;
  unsigned cur = tile_off(tile);
  if (LAYOUT != 8) {
#pragma unroll
    for (int i = 0; i < HRING; ++i) issue_hg(cur, i);
  }
  // LAYOUT 7: the wave owns TWO tiles (64 pixel rows) and every weight read serves both -- half the LDS bytes per MFMA
  constexpr unsigned T2 = 30u * 128u * 4096u;                     // the second tile: 4096 tiles further (another part of the tensor)
  auto issue_hg2 = [&](unsigned base, int i) {
    const int ddy = (i % 6) / 2 - 1;
    const unsigned o = base + T2 + (unsigned)(ddy * wci) + (unsigned)((i & 1) * 32);
    rawd[i % HRING][0] = buf_ld16(rin, o);
    rawd[i % HRING][1] = buf_ld16(rin, o + 16);
  };
  if (LAYOUT == 7) {
#pragma unroll
    for (int i = 0; i < HRING; ++i) issue_hg2(cur, i);
  }
  Bf16Planes pc[2], opm[2], opp[2], pd[2];
  // LAYOUT 8: the tile's rows through LDS in FULL lines -- 4 coalesced 16-byte loads per row (1 KB each: 8 pixels x 128 B), stored to a
  // per-wave ring of three 4-KB row images with the 16-byte chunk index XOR-ed with the pixel (conflict-free 16-byte reads at a 128-byte
  // pitch), the A operand of a half-group read back as 2 x ds_read_b128 -- instead of 64 lane-accesses per fragment-shaped load
  floatx4 rrow[2][4];
  unsigned char* stage = smem + 54 * 1024 + wave * 12 * 1024;
  auto row_base = [&](int tl, int ddy) { return (unsigned)(tl * 30 * 128 + 42 * 128) + (unsigned)(ddy * wci) + (unsigned)(lane * 16); };
  auto load_row = [&](unsigned base, int slot) {
#pragma unroll
    for (int q = 0; q < 4; ++q) rrow[slot][q] = buf_ld16(rin, base + (unsigned)(q * 1024));
  };
  auto write_row = [&](int slot, int lslot) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int px = q * 8 + (lane >> 3), c = lane & 7;
      *reinterpret_cast<floatx4*>(stage + lslot * 4096 + px * 128 + ((c ^ (px & 7)) * 16)) = rrow[slot][q];
    }
  };
  auto read_op = [&](int lslot, int kb, floatx4* dst) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = h * 4 + kb * 2 + e;
      dst[e] = *reinterpret_cast<const floatx4*>(stage + lslot * 4096 + j * 128 + ((c ^ (j & 7)) * 16));
    }
  };
  if (LAYOUT == 8) {
    load_row(row_base(tile, -1), 0);
    write_row(0, 0);
    load_row(row_base(tile, 0), 1);
    read_op(0, 0, rawc[0]);
  }
  mi_u32x4 pb[2][3];
  split_pair<0>(rawc[0][0], pc[0]); split_pair<1>(rawc[0][0], pc[0]); split_pair<2>(rawc[0][1], pc[0]); split_pair<3>(rawc[0][1], pc[0]);
  if (LAYOUT == 7) { split_pair<0>(rawd[0][0], pd[0]); split_pair<1>(rawd[0][0], pd[0]); split_pair<2>(rawd[0][1], pd[0]); split_pair<3>(rawd[0][1], pd[0]); }
  unsigned selm = 0xffffffffu, selp = (j == 17) ? 0u : 0xffffffffu;
  asm volatile("" : "+v"(selm), "+v"(selp));
#define SHIFTR(dst, src, P, R, CTRL, SEL) dst.P[R] = (FLAGS & 1) ? (src.P[R] & SEL) : ((unsigned)__builtin_amdgcn_mov_dpp((int)src.P[R], CTRL, 0xf, 0xf, true) & SEL);
#define SHIFT6(dst, src, CTRL, SEL) { SHIFTR(dst, src, h, 0, CTRL, SEL) SHIFTR(dst, src, h, 1, CTRL, SEL) SHIFTR(dst, src, h, 2, CTRL, SEL) SHIFTR(dst, src, h, 3, CTRL, SEL) \
    SHIFTR(dst, src, m, 0, CTRL, SEL) SHIFTR(dst, src, m, 1, CTRL, SEL) asm volatile("" : "+v"(dst.h[0]), "+v"(dst.h[1]), "+v"(dst.h[2]), "+v"(dst.h[3]), "+v"(dst.m[0]), "+v"(dst.m[1])); }
#define SHIFT6B(dst, src, CTRL, SEL) { SHIFTR(dst, src, m, 2, CTRL, SEL) SHIFTR(dst, src, m, 3, CTRL, SEL) SHIFTR(dst, src, l, 0, CTRL, SEL) SHIFTR(dst, src, l, 1, CTRL, SEL) \
    SHIFTR(dst, src, l, 2, CTRL, SEL) SHIFTR(dst, src, l, 3, CTRL, SEL) asm volatile("" : "+v"(dst.m[2]), "+v"(dst.m[3]), "+v"(dst.l[0]), "+v"(dst.l[1]), "+v"(dst.l[2]), "+v"(dst.l[3])); }
#define READB(dst, U) { dst[0] = l4[((U) * 3 + 0) * 64]; dst[1] = l4[((U) * 3 + 1) * 64]; dst[2] = l4[((U) * 3 + 2) * 64]; }
  if (LAYOUT != 0 && LAYOUT != 6 && NACC == 1) { SHIFT6(opm[0], pc[0], 0x138, selm) SHIFT6B(opm[0], pc[0], 0x138, selm) SHIFT6(opp[0], pc[0], 0x130, selp) SHIFT6B(opp[0], pc[0], 0x130, selp) }
  READB(pb[0], unit_of(0, 0));
  typedef AccT<SHAPE> Acc;
  Acc acc[3], accd[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { acc[a].zero(); accd[a].zero(); }
  float sink = 0.f;
  if (LAYOUT == 3 && wave >= 4) asm volatile("s_barrier" ::: "memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int n = 0; n < tiles_per_wave; ++n, tile += NW) {
    const unsigned nxt = tile_off(tile + NW);
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      if (LAYOUT != 8) { if (i + HRING < NH) issue_hg(cur, i + HRING); else issue_hg(nxt, i + HRING - NH); }
      if (LAYOUT == 7) { if (i + HRING < NH) issue_hg2(cur, i + HRING); else issue_hg2(nxt, i + HRING - NH); }
      const Bf16Planes& pc_ = pc[i & 1];
      Bf16Planes& nc = pc[(i + 1) & 1];
      const floatx4* rc = rawc[(i + 1) % HRING];
      Acc& a0 = acc[0];
      Acc& a1 = acc[NACC == 3 ? 1 : 0];
      Acc& a2 = acc[NACC == 3 ? 2 : 0];
      if constexpr (LAYOUT == 0) {
        Bf16Planes& om = opm[0]; Bf16Planes& op = opp[0];
        if constexpr (NACC == 1) {
          READB(pb[(3 * i + 1) & 1], unit_of(i, -1));
          SIX(pc_, pb[(3 * i) & 1], a0, SHIFT6(om, pc_, 0x138, selm), SHIFT6B(om, pc_, 0x138, selm), split_pair<0>(rc[0], nc), split_pair<1>(rc[0], nc))
          READB(pb[(3 * i + 2) & 1], unit_of(i, 1));
          SIX(om, pb[(3 * i + 1) & 1], a0, SHIFT6(op, pc_, 0x130, selp), SHIFT6B(op, pc_, 0x130, selp), split_pair<2>(rc[1], nc), split_pair<3>(rc[1], nc))
          READB(pb[(3 * i + 3) & 1], unit_of((i + 1) % NH, 0));
          SIX(op, pb[(3 * i + 2) & 1], a0, (void)0, (void)0, (void)0, (void)0)
        } else {
          READB(pb[(3 * i + 1) & 1], unit_of(i, -1));
          SIX(pc_, pb[(3 * i) & 1], a0, (void)0, split_pair<0>(rc[0], nc), (void)0, split_pair<1>(rc[0], nc))
          READB(pb[(3 * i + 2) & 1], unit_of(i, 1));
          SIX(pc_, pb[(3 * i + 1) & 1], a1, (void)0, split_pair<2>(rc[1], nc), (void)0, split_pair<3>(rc[1], nc))
          READB(pb[(3 * i + 3) & 1], unit_of((i + 1) % NH, 0));
          SIX(pc_, pb[(3 * i + 2) & 1], a2, (void)0, (void)0, (void)0, (void)0)
        }
      } else if constexpr (LAYOUT == 8) {
        // half-group i = (row R = i / 2, k half i & 1); rows are counted through the tiles: 3 per tile, ring slots by row number
        const int R = i >> 1, kb = i & 1;
        if (kb == 0) {
          // the row after next into the registers of this row; the next row (loaded one row ago) into its LDS slot
          const int r2 = R + 2;
          load_row(r2 < 3 ? row_base(tile, r2 - 1) : row_base(tile + NW, r2 - 4), R & 1);
          write_row((R + 1) & 1, (R + 1) % 3);
        }
        // the operand of half-group i + 1 from LDS, split between this half-group's MFMAs
        { const int i1 = (i + 1) % NH; read_op(((i + 1) >> 1) % 3, i1 & 1, rawc[(i + 1) % HRING]); }
        READB(pb[(3 * i + 1) & 1], unit_of(i, -1));
        SIX(pc_, pb[(3 * i) & 1], a0, (void)0, split_pair<0>(rc[0], nc), (void)0, split_pair<1>(rc[0], nc))
        READB(pb[(3 * i + 2) & 1], unit_of(i, 1));
        SIX(pc_, pb[(3 * i + 1) & 1], a1, (void)0, split_pair<2>(rc[1], nc), (void)0, split_pair<3>(rc[1], nc))
        READB(pb[(3 * i + 3) & 1], unit_of((i + 1) % NH, 0));
        SIX(pc_, pb[(3 * i + 2) & 1], a2, (void)0, (void)0, (void)0, (void)0)
      } else if constexpr (LAYOUT == 7) {
        const Bf16Planes& pd_ = pd[i & 1];
        Bf16Planes& nd = pd[(i + 1) & 1];
        const floatx4* rd = rawd[(i + 1) % HRING];
        READB(pb[(3 * i + 1) & 1], unit_of(i, -1));
        SIX(pc_, pb[(3 * i) & 1], acc[0], (void)0, split_pair<0>(rc[0], nc), (void)0, split_pair<1>(rc[0], nc))
        SIX(pd_, pb[(3 * i) & 1], accd[0], (void)0, split_pair<0>(rd[0], nd), (void)0, split_pair<1>(rd[0], nd))
        READB(pb[(3 * i + 2) & 1], unit_of(i, 1));
        SIX(pc_, pb[(3 * i + 1) & 1], acc[1], (void)0, split_pair<2>(rc[1], nc), (void)0, split_pair<3>(rc[1], nc))
        SIX(pd_, pb[(3 * i + 1) & 1], accd[1], (void)0, split_pair<2>(rd[1], nd), (void)0, split_pair<3>(rd[1], nd))
        READB(pb[(3 * i + 3) & 1], unit_of((i + 1) % NH, 0));
        SIX(pc_, pb[(3 * i + 2) & 1], acc[2], (void)0, (void)0, (void)0, (void)0)
        SIX(pd_, pb[(3 * i + 2) & 1], accd[2], (void)0, (void)0, (void)0, (void)0)
      } else if constexpr (LAYOUT == 6) {
        // one split stage / four shifts behind every MFMA (the split of a pair is three dependent stages; written as whole pairs every third slot)
        Bf16Planes& om = opm[0]; Bf16Planes& op = opp[0];
        READB(pb[(3 * i + 1) & 1], unit_of(i, -1));
        SB; MF(pc_.l, pb[(3 * i) & 1][0], a0); SB; if (NACC == 1) { SHIFTR(om, pc_, h, 0, 0x138, selm) SHIFTR(om, pc_, h, 1, 0x138, selm) SHIFTR(om, pc_, h, 2, 0x138, selm) SHIFTR(om, pc_, h, 3, 0x138, selm) asm volatile("" : "+v"(om.h[0]), "+v"(om.h[1]), "+v"(om.h[2]), "+v"(om.h[3])); }
        SB; MF(pc_.h, pb[(3 * i) & 1][2], a0); SB; if (NACC == 1) { SHIFTR(om, pc_, m, 0, 0x138, selm) SHIFTR(om, pc_, m, 1, 0x138, selm) SHIFTR(om, pc_, m, 2, 0x138, selm) SHIFTR(om, pc_, m, 3, 0x138, selm) asm volatile("" : "+v"(om.m[0]), "+v"(om.m[1]), "+v"(om.m[2]), "+v"(om.m[3])); }
        SB; MF(pc_.m, pb[(3 * i) & 1][1], a0); SB; if (NACC == 1) { SHIFTR(om, pc_, l, 0, 0x138, selm) SHIFTR(om, pc_, l, 1, 0x138, selm) SHIFTR(om, pc_, l, 2, 0x138, selm) SHIFTR(om, pc_, l, 3, 0x138, selm) asm volatile("" : "+v"(om.l[0]), "+v"(om.l[1]), "+v"(om.l[2]), "+v"(om.l[3])); }
        SB; MF(pc_.m, pb[(3 * i) & 1][0], a0); SB; if (NACC == 1) { SHIFTR(op, pc_, h, 0, 0x130, selp) SHIFTR(op, pc_, h, 1, 0x130, selp) SHIFTR(op, pc_, h, 2, 0x130, selp) SHIFTR(op, pc_, h, 3, 0x130, selp) asm volatile("" : "+v"(op.h[0]), "+v"(op.h[1]), "+v"(op.h[2]), "+v"(op.h[3])); }
        SB; MF(pc_.h, pb[(3 * i) & 1][1], a0); SB; if (NACC == 1) { SHIFTR(op, pc_, m, 0, 0x130, selp) SHIFTR(op, pc_, m, 1, 0x130, selp) SHIFTR(op, pc_, m, 2, 0x130, selp) SHIFTR(op, pc_, m, 3, 0x130, selp) asm volatile("" : "+v"(op.m[0]), "+v"(op.m[1]), "+v"(op.m[2]), "+v"(op.m[3])); }
        SB; MF(pc_.h, pb[(3 * i) & 1][0], a0); SB; if (NACC == 1) { SHIFTR(op, pc_, l, 0, 0x130, selp) SHIFTR(op, pc_, l, 1, 0x130, selp) SHIFTR(op, pc_, l, 2, 0x130, selp) SHIFTR(op, pc_, l, 3, 0x130, selp) asm volatile("" : "+v"(op.l[0]), "+v"(op.l[1]), "+v"(op.l[2]), "+v"(op.l[3])); }
        READB(pb[(3 * i + 2) & 1], unit_of(i, 1));
        const Bf16Planes& u1 = NACC == 1 ? om : pc_;
        SB; MF(u1.l, pb[(3 * i + 1) & 1][0], a1); SB;
        SB; MF(u1.h, pb[(3 * i + 1) & 1][2], a1); SB; split_pair<0>(rc[0], nc);
        SB; MF(u1.m, pb[(3 * i + 1) & 1][1], a1); SB;
        SB; MF(u1.m, pb[(3 * i + 1) & 1][0], a1); SB;
        SB; MF(u1.h, pb[(3 * i + 1) & 1][1], a1); SB; split_pair<1>(rc[0], nc);
        SB; MF(u1.h, pb[(3 * i + 1) & 1][0], a1); SB;
        READB(pb[(3 * i + 3) & 1], unit_of((i + 1) % NH, 0));
        const Bf16Planes& u2 = NACC == 1 ? op : pc_;
        SB; MF(u2.l, pb[(3 * i + 2) & 1][0], a2); SB;
        SB; MF(u2.h, pb[(3 * i + 2) & 1][2], a2); SB; split_pair<2>(rc[1], nc);
        SB; MF(u2.m, pb[(3 * i + 2) & 1][1], a2); SB;
        SB; MF(u2.m, pb[(3 * i + 2) & 1][0], a2); SB;
        SB; MF(u2.h, pb[(3 * i + 2) & 1][1], a2); SB; split_pair<3>(rc[1], nc);
        SB; MF(u2.h, pb[(3 * i + 2) & 1][0], a2); SB;
      } else {
        // ---- vector phase: the split of half-group i + 1 and (one accumulator) its two shifted copies
        Bf16Planes& om_n = opm[(i + 1) & 1]; Bf16Planes& op_n = opp[(i + 1) & 1];
        SB;
        if constexpr (LAYOUT != 4) {
          split_pair<0>(rc[0], nc); split_pair<1>(rc[0], nc); split_pair<2>(rc[1], nc); split_pair<3>(rc[1], nc);
          if constexpr (NACC == 1) { SHIFT6(om_n, nc, 0x138, selm) SHIFT6B(om_n, nc, 0x138, selm) SHIFT6(op_n, nc, 0x130, selp) SHIFT6B(op_n, nc, 0x130, selp) }
        } else {
          asm volatile("" : "+v"(nc.h[0]), "+v"(nc.m[0]), "+v"(nc.l[0]) : "v"(rc[0]), "v"(rc[1]));      // (the loads stay live)
        }
        SB;
        if constexpr (LAYOUT == 3) asm volatile("s_barrier" ::: "memory");
        if constexpr (LAYOUT == 2) asm volatile("s_setprio 1");
        // ---- matrix phase: 18 MFMAs, the weight planes of the next unit read from LDS one unit ahead
        const Bf16Planes& u1 = NACC == 1 ? opm[i & 1] : pc_;
        const Bf16Planes& u2 = NACC == 1 ? opp[i & 1] : pc_;
        if constexpr (LAYOUT != 5) {
          READB(pb[(3 * i + 1) & 1], unit_of(i, -1));
          SIX(pc_, pb[(3 * i) & 1], a0, (void)0, (void)0, (void)0, (void)0)
          READB(pb[(3 * i + 2) & 1], unit_of(i, 1));
          SIX(u1, pb[(3 * i + 1) & 1], a1, (void)0, (void)0, (void)0, (void)0)
          READB(pb[(3 * i + 3) & 1], unit_of((i + 1) % NH, 0));
          SIX(u2, pb[(3 * i + 2) & 1], a2, (void)0, (void)0, (void)0, (void)0)
        } else {
          asm volatile("" :: "v"(pc_.h[0]), "v"(pc_.m[1]), "v"(pc_.l[2]), "v"(u1.h[3]), "v"(u1.m[0]), "v"(u1.l[1]), "v"(u2.h[2]), "v"(u2.m[3]), "v"(u2.l[0]));
          asm volatile("" :: "v"(pc_.h[1]), "v"(pc_.m[2]), "v"(pc_.l[3]), "v"(u1.h[0]), "v"(u1.m[1]), "v"(u1.l[2]), "v"(u2.h[3]), "v"(u2.m[0]), "v"(u2.l[1]));
        }
        if constexpr (LAYOUT == 2) asm volatile("s_setprio 0");
        if constexpr (LAYOUT == 3) asm volatile("s_barrier" ::: "memory");
      }
    }
    cur = nxt;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (LAYOUT == 3 && wave < 4) asm volatile("s_barrier" ::: "memory");
#pragma unroll
  for (int a = 0; a < 3; ++a) sink += acc[a].sum() + accd[a].sum();
  sink += __uint_as_float(pd[0].h[1] ^ pd[1].l[2]);
  sink += __uint_as_float(pc[0].h[0] ^ pc[1].l[3] ^ opm[0].m[1] ^ opp[0].l[2] ^ opm[1].h[2] ^ opp[1].m[3]);
  out[(size_t)blockIdx.x * 512 + tid] = sink;
  if (lane == 0) { stamps[(blockIdx.x * 8 + wave) * 2] = t1 - t0; stamps[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0; }
}

// ---- issue model: a dependent chain of MFMAs with NV independent v_fma_f32 behind each
template <int NV>
__global__ __launch_bounds__(512, 1) void fma_sweep(float* out, unsigned long long* stamps, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  mi_u32x4 a = {threadIdx.x * 0x3f803f80u, 0x3f813f80u, 0x3f803f82u, 0x3f833f80u};
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
  const float b = 1.0001f, c = 0.5f;
  if (threadIdx.x == 0) smem[0] = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 18; ++m) {
      SB; acc = MI_BF_MFMA(a, a, acc); SB;
#pragma unroll
      for (int u = 0; u < NV; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u % 8]) : "v"(b), "v"(c));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[r];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = t1 - t0; stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0; }
}

__global__ void fill(float* p, size_t n, unsigned seed, int relu) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    float v = ((int)x) * (1.0f / 2147483648.0f) * 1.7f;
    p[i] = relu ? (v > 0.f ? v : 0.f) : v;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float* g_stream; static unsigned g_bytes; static mi_u32x4* g_w; static float* g_out; static unsigned long long* g_st;

// Sustained regime: the clock the chip holds depends on the load (MI355X_MICROARCH.md, DVFS give-back), so every variant is launched back to back
// for ~0.1 s before the timed launches; reported: wall time per launch (events around NT launches), the median wave's shader cycles per MFMA and
// SIMD, and the in-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz.
template <class K> static void run(const char* name, K kern, int threads, int tpw, double mfma_per_wave, bool is_fma, int nwarm = 600, int nt = 200) {
  const int grid = 256;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
    if (is_fma) hipLaunchKernelGGL(((void (*)(float*, unsigned long long*, int))kern), dim3(grid), dim3(threads), 152 * 1024, 0, g_out, g_st, tpw);
    else hipLaunchKernelGGL(((void (*)(const float*, unsigned, const mi_u32x4*, float*, unsigned long long*, int))kern), dim3(grid), dim3(threads), 152 * 1024, 0, g_stream, g_bytes, g_w, g_out, g_st, tpw);
  };
  for (int i = 0; i < nwarm; ++i) launch();
  CK(hipEventRecord(e0));
  for (int i = 0; i < nt; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= nt;
  std::vector<unsigned long long> st(grid * 8 * 2);
  CK(hipMemcpy(st.data(), g_st, st.size() * 8, hipMemcpyDeviceToHost));
  const int nw = threads / 64;
  std::vector<double> cyc, clk;
  for (int b = 0; b < grid; ++b) for (int w = 0; w < nw; ++w) { const double c = (double)st[(b * 8 + w) * 2], r = (double)st[(b * 8 + w) * 2 + 1]; cyc.push_back(c); if (r > 0) clk.push_back(c / r * 0.1); }
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  const int wps = threads / 256;
  const double per_simd = mfma_per_wave * wps;
  printf("%-40s %7.4f ms/launch  %6.2f ns/MFMA/SIMD  %6.1f cyc/MFMA/SIMD (median wave)  clock %.3f GHz\n", name, ms, ms * 1e6 / per_simd,
         cyc[cyc.size() / 2] / per_simd, clk.empty() ? 0.0 : clk[clk.size() / 2]);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int relu = argc > 1 ? atoi(argv[1]) : 1;
  const size_t n = (size_t)800 * 1764 * 32 + (1 << 20);
  g_bytes = (unsigned)((size_t)800 * 1764 * 32 * 4);
  CK(hipMalloc(&g_stream, n * 4)); CK(hipMalloc(&g_w, 36 * 3 * 64 * 16)); CK(hipMalloc(&g_out, 256 * 512 * 4)); CK(hipMalloc(&g_st, 256 * 8 * 2 * 8));
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, g_stream, n, 12345u, relu);
  hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, (float*)g_w, (size_t)36 * 3 * 64 * 4, 777u, 0);
  CK(hipDeviceSynchronize());
  // cfg2 block 2: 47,040 tiles of 30 pixels over 256 workgroups x 8 waves = 23 tiles per wave; one term = 6 half-groups = 108 MFMAs per tile
  const int tpw = 23; const double mpw = tpw * 108.0;
  printf("# operands: %s; 256 workgroups x 8 waves (2 per SIMD), %d tiles per wave, 108 MFMAs per tile\n", relu ? "ReLU-like (half zeros)" : "dense", tpw);
#define RUN(L, N, F) run("layout " #L " acc " #N " flags " #F, probe<L, N, F>, 512, tpw, mpw, false)
#define RUN16(L, N, F) run("layout " #L " acc " #N " flags " #F " 16x16x32", probe<L, N, F, 1>, 512, tpw, mpw, false)
  RUN(4, 1, 0); RUN(5, 1, 0); RUN(0, 1, 0); RUN(1, 1, 0); RUN(3, 1, 0); RUN(6, 1, 0);
  RUN(5, 3, 0); RUN(0, 3, 0); RUN(1, 3, 0); RUN(3, 3, 0); RUN(6, 3, 0);
  RUN16(4, 1, 0); RUN16(0, 1, 0); RUN16(0, 3, 0); RUN16(6, 3, 0); RUN16(3, 3, 0);
  RUN(4, 1, 2); RUN(0, 1, 2); RUN(0, 3, 2); RUN(5, 1, 2);
  // two tiles per wave sharing every weight read, one wave per SIMD (4-wave workgroups): 216 MFMAs per loop trip and wave
  run("layout 7 (2 tiles/wave) 32x32x16, 1 wave/SIMD", probe<7, 3, 0, 0>, 256, tpw, mpw * 2, false);
  run("layout 7 (2 tiles/wave) 16x16x32, 1 wave/SIMD", probe<7, 3, 0, 1>, 256, tpw, mpw * 2, false);
  run("layout 7 (2 tiles/wave) 16x16x32, 2 waves/SIMD", probe<7, 3, 0, 1>, 512, tpw, mpw * 2, false);
  run("layout 0 acc 3 16x16x32, 1 wave/SIMD", probe<0, 3, 0, 1>, 256, tpw, mpw, false);
  run("layout 8 (rows through LDS) 32x32x16", probe<8, 3, 0, 0>, 512, tpw, mpw, false);
  run("layout 8 (rows through LDS) 16x16x32", probe<8, 3, 0, 1>, 512, tpw, mpw, false);
  run("layout 0 acc 3 16x16x32 (again)", probe<0, 3, 0, 1>, 512, tpw, mpw, false);
  RUN(0, 1, 0);                                                          // (repeat of the shipped order: drift check)
  const int it = 140; const double fm = it * 18.0;
#define RUNF(NV) run("fma sweep NV=" #NV " 2 waves", fma_sweep<NV>, 512, it, fm, true); run("fma sweep NV=" #NV " 1 wave", fma_sweep<NV>, 256, it, fm, true)
  RUNF(0); RUNF(4); RUNF(6); RUNF(8);
  return 0;
}

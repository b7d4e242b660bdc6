// Stride-1 hidden -> hidden 3x3 convolution (forward / dgrad, one or two terms) on v_mfma_f32_16x16x32_bf16 -- round 5.
// Included by conv_mfma.hip (uses its helpers); replaces, for the split-bf16 operand form, the 32x32x16 kernel above it
// (conv3x3_s1_mfma_kernel<.., BF = true>), which stays selectable (MI_CONV_B16=0 / mi_conv_set_b16) and remains the fp16 form's kernel.
// Same arithmetic as that kernel -- every fp32 operand as three exact bf16 planes, six products per multiply-add, fp32 accumulation
// (bf16_split.h) -- and the same tiles of 30 output pixels (32 pixel rows, the first and the last one halo).  Replaces the conv2d forward /
// backward / double-backward behind ConvBlock.conv (reference core_functions/vision_models.py:177-185,189).
//
// Why another shape (tools/conv_sched_probe.hip, profiles/r5/conv_sched_probe_*.txt; MI355X_MICROARCH.md "DVFS give-back" 7): these
// kernels run at the socket power cap, where time = energy.  In the sustained regime the K loop of the 32x32x16 kernel issues at 36 cycles
// per MFMA and SIMD -- its vector work IS hidden -- but the chip holds 1.77 GHz on it; the same FLOPs as 16x16x32 MFMAs hold 2.1-2.2 GHz
// (-10 % wall for the loop with the lane shifts gone, -14 % with the two waves of a SIMD in ping-pong).  The lane shifts go because the
// horizontal taps are three ACCUMULATORS fed from the unshifted planes:  out[x] = P0[x] + P-[x - 1] + P+[x + 1], the displacement applied
// once per tile to the sums (pixel index = accumulator register and lane group: three of four rows are register renaming, the fourth a
// ds_bpermute by 16 lanes), the image-column padding as bit masks from a ballot -- instead of 24 v_and_b32_dpp per 18 MFMAs in the loop.
//
// Operand layout (16x16x32: lane l = (n = l & 15, g = l >> 4) holds A[row n][k = 8g + 0..7] and B[k = 8g + 0..7][col n]; D[row 4g + r][col n]):
//   A  block mb (0 / 1): pixel row pm = 16 mb + n of the tile, channels cc*32 + 8g + 0..7 -- one lane's 32 contiguous bytes (2 x 16-B
//      loads; the four lanes of a pixel cover its 128-byte line), three planes of 4 registers;
//   B  block nb (0 / 1): output channel cbase + 2n + nb (even / odd channels, so that a lane's two results of a pixel are adjacent:
//      8-byte stores, a pixel's 32 channels one 128-byte line per instruction);
//   D  block (mb, nb): register r = pixel row 16 mb + 4g + r, channel cbase + 2n + nb.
// Weights in LDS, operand order: 16-B item ((unit*3 + plane)*2 + nb)*64 + lane, unit = (term*9 + tap)*NCC + cc: 54 KB per 32-channel term.
// Pipeline per ROW-STEP (term, row displacement, 32-channel chunk): 4 row loads two row-steps ahead; the split of the NEXT row-step's 16
// values in 40 stages of 1-3 instructions, one behind every other MFMA; weight planes through a ring of three 8-register chunks read two
// chunks ahead (l-plane: 4 MFMAs, m-plane: 8, h-plane: 12 per tap); 72 MFMAs.
#pragma once

#ifndef MI_CONV_B16_DEFAULT
#define MI_CONV_B16_DEFAULT 1
#endif

__device__ __forceinline__ void buf_st2_untracked(mi_u32x4 rsrc, unsigned off, float v0, float v1) {
  floatx2 v = {v0, v1};
  asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen" : : "v"(v), "v"(off), "s"(rsrc) : "memory");
}
__device__ __forceinline__ floatx2 buf_ld8(mi_rsrc r, unsigned off) {
  return __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
}
// -1 where bit `bit` (a compile-time constant after unrolling) of v is set; assembly, so that the merge that uses it stays a bit operation
__device__ __forceinline__ int lane_mask_bit_rt(unsigned v, int bit) {
  int r;
  asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r) : "v"(v), "i"(bit));
  return r;
}
__device__ __forceinline__ float bf16_sub_v(float a, float b) {
  float r;
  asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#define MI_B16_MFMA(x, y, acc) __builtin_amdgcn_mfma_f32_16x16x32_bf16(MI_BF8(x), MI_BF8(y), acc, 0, 0, 0)

// lanes (n, g) hold channels 2n, 2n + 1: sums over the lane groups, then the waves (LDS), one partial per workgroup, last-arriver fold
__device__ __forceinline__ void stats_block_reduce_pairs(double s0, double q0, double s1, double q1, double* ldsd, int lane, int wave, int nwaves,
                                                         double* partial_task, int co_total, int cbase, const FinArgs& fin, int task, int bx) {
  double* partial_blk = partial_task + (size_t)bx * 2 * co_total;
  s0 += __shfl_xor(s0, 16, 64); q0 += __shfl_xor(q0, 16, 64); s1 += __shfl_xor(s1, 16, 64); q1 += __shfl_xor(q1, 16, 64);
  s0 += __shfl_xor(s0, 32, 64); q0 += __shfl_xor(q0, 32, 64); s1 += __shfl_xor(s1, 32, 64); q1 += __shfl_xor(q1, 32, 64);
  __syncthreads();  // weights in LDS are dead from here on
  if (lane < 16) {
    ldsd[(wave * 2 + 0) * 32 + 2 * lane] = s0; ldsd[(wave * 2 + 0) * 32 + 2 * lane + 1] = s1;
    ldsd[(wave * 2 + 1) * 32 + 2 * lane] = q0; ldsd[(wave * 2 + 1) * 32 + 2 * lane + 1] = q1;
  }
  __syncthreads();
  if (wave == 0 && lane < 32) {
    double ts = 0.0, tq = 0.0;
    for (int w = 0; w < nwaves; ++w) {
      ts += ldsd[(w * 2 + 0) * 32 + lane];
      tq += ldsd[(w * 2 + 1) * 32 + lane];
    }
    mi_partial_store(partial_blk + cbase + lane, ts, fin);
    mi_partial_store(partial_blk + co_total + cbase + lane, tq, fin);
  }
  mi_finalize_last(fin, partial_task, gridDim.x, co_total, task, gridDim.x * gridDim.z, ldsd);
}

// Diagnostic build only (-DMI_B16_STAMPS, another file name, selected with MI_MAML_LIB; tools/conv_b16_stamps.py): waves 0 and 4 of workgroup
// (0, 0, 0) -- SIMD partners in an 8-wave workgroup -- write shader-clock totals to the buffer of mi_debug_conv_stamps:
// [8 w4 + 0] kernel start, [1] weights staged, [2] sum over tiles of the epilogue's first part (the combination of the three accumulators), [3] of the K loops (with the tile head), [4] of the epilogues,
// [5] last tile done, [6] kernel end, [7] tiles.  No stamp executes in the production build.
#ifndef MI_B16_EXP
#define MI_B16_EXP 0           // (diagnostic builds only: bit 0 drops the epilogue's stores, bit 1 its statistics -- what each costs; results are then wrong)
#endif
#ifdef MI_B16_STAMPS
#define B16_ST(...) __VA_ARGS__
#else
#define B16_ST(...)
#endif
template <int CI, int NTERMS, int EPI, int MODE>
__global__ __launch_bounds__((ConvWaves<CI, NTERMS, true, false>::value * 64), 2) void conv3x3_s1_b16_kernel(ConvArgs a) {
  constexpr int NW = ConvWaves<CI, NTERMS, true, false>::value, NT = NW * 64, CO = CI;
  constexpr int NCC = CI / 32, NS = NTERMS * 3 * NCC;             // row-steps per tile
  constexpr int NU = NTERMS * 9 * NCC;                            // weight units (term, tap, chunk)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  // XCD-aware placement, as conv3x3_s1_mfma_kernel: XCD k works through a contiguous run of (task, band, channel tile) triples
  int bx = blockIdx.x, task = blockIdx.y, ct = blockIdx.z;
  {
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    if ((total & 7u) == 0u) {
      const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      const unsigned nl = (lin & 7u) * (total >> 3) + (lin >> 3);
      ct = (int)(nl % gridDim.z);
      bx = (int)((nl / gridDim.z) % gridDim.x);
      task = (int)(nl / (gridDim.z * gridDim.x));
    }
  }
  const int H = a.g.h, W = a.g.w;
  const int cbase = ct * 32;
  B16_ST(unsigned long long* stp = (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (wave & 3) == 0 && lane == 0 && g_conv_stamps) ? g_conv_stamps + (wave >> 2) * 8 : nullptr;
         unsigned long long st_k = 0, st_epi = 0, st_n = 0, st_a = 0, st_b = 0, st_comb = 0; if (stp) stp[0] = __builtin_amdgcn_s_memtime();)

  // ---- stage this task's weights as bf16 planes in operand order (all of a thread's loads first, then the splits)
  {
    mi_u32x4* lw = reinterpret_cast<mi_u32x4*>(lds);
    constexpr int NIT = NU * 2 * 64, IPT = (NIT + NT - 1) / NT;     // items: the 8 weights k = cc*32 + 8g + 0..7 of one output channel
    floatx4 w0[IPT], w1[IPT];
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      const int it = tid + q * NT;
      const int ln = it & 63, grp = it >> 6;
      const int nb = grp & 1, u = grp >> 1;
      const int cc = u % NCC, tt = u / NCC, term = tt / 9, tap = tt - term * 9;
      const int k0 = cc * 32 + (ln >> 4) * 8, co = cbase + 2 * (ln & 15) + nb;
      const bool live = it < NIT;
      const float* wsrc = a.wt[live ? term : 0] + (size_t)task * a.wstride;
      if (MODE == 0) {
        const float* src = wsrc + ((size_t)(live ? tap : 0) * CI + k0) * CO + co;
#pragma unroll
        for (int i = 0; i < 4; ++i) { w0[q][i] = src[(size_t)i * CO]; w1[q][i] = src[(size_t)(i + 4) * CO]; }
      } else {
        const float* src = wsrc + ((size_t)(live ? tap : 0) * CO + co) * CI + k0;
        w0[q] = *reinterpret_cast<const floatx4*>(src);
        w1[q] = *reinterpret_cast<const floatx4*>(src + 4);
      }
    }
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      const int it = tid + q * NT;
      if (it < NIT) {
        const int ln = it & 63, grp = it >> 6;
        const int nb = grp & 1, u = grp >> 1;
        Bf16Planes pw;
        bf16_split8(w0[q], w1[q], pw);
        lw[((u * 3 + 0) * 2 + nb) * 64 + ln] = mi_u32x4{pw.h[0], pw.h[1], pw.h[2], pw.h[3]};
        lw[((u * 3 + 1) * 2 + nb) * 64 + ln] = mi_u32x4{pw.m[0], pw.m[1], pw.m[2], pw.m[3]};
        lw[((u * 3 + 2) * 2 + nb) * 64 + ln] = mi_u32x4{pw.l[0], pw.l[1], pw.l[2], pw.l[3]};
      }
    }
  }
  // (no barrier yet: the first operand loads go out before the wait for the staged weights)

  const int mpix = a.mpix;
  const unsigned hw = (unsigned)(H * W);
  const float rhw = 1.0f / (float)hw, rw = 1.0f / (float)W;
  const size_t t_elems = (size_t)mpix * CI;
  const unsigned t_bytes = (unsigned)(t_elems * 4);
  mi_rsrc rin[NTERMS];
#pragma unroll
  for (int term = 0; term < NTERMS; ++term)
    rin[term] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in[term] + (size_t)task * t_elems), 0, t_bytes, 0x00020000);
  const mi_rsrc rout = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)task * t_elems), 0, t_bytes, 0x00020000);
  const unsigned long long out_addr = (unsigned long long)(a.out + (size_t)task * t_elems);
  const mi_u32x4 rout_raw = {(unsigned)out_addr, (unsigned)(out_addr >> 32) & 0xffffu, t_bytes, 0x00020000u};
  mi_rsrc rz = rout;
  float mu_c[2] = {0.f, 0.f}, r_c[2] = {0.f, 0.f};
  if (EPI == EPI_TSTATS) {
    rz = __builtin_amdgcn_make_buffer_rsrc((void*)(a.z + (size_t)task * t_elems), 0, t_bytes, 0x00020000);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      mu_c[nb] = a.mu[(size_t)task * CO + cbase + 2 * n + nb];
      r_c[nb] = a.rstd[(size_t)task * CO + cbase + 2 * n + nb];
    }
  }
  mi_rsrc rbp = rout, rbzh = rout, rbzhd = rout, rbdp = rout;
  const bool bred_arg = EPI == EPI_BRED && a.barg != nullptr;
  if (EPI == EPI_BRED) {
    if (bred_arg) rbp = __builtin_amdgcn_make_buffer_rsrc((void*)(a.barg + (size_t)task * t_elems), 0, (unsigned)t_elems, 0x00020000);
    else rbp = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bp + (size_t)task * t_elems), 0, t_bytes, 0x00020000);
    rbzh = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bzh + (size_t)task * t_elems), 0, t_bytes, 0x00020000);
    if (NTERMS == 2) {
      rbzhd = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bzhd + (size_t)task * t_elems), 0, t_bytes, 0x00020000);
      rbdp = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bdp + (size_t)task * t_elems), 0, t_bytes, 0x00020000);
    }
  }
  double s[2] = {0.0, 0.0}, q[2] = {0.0, 0.0};
  const int wci = W * CI * 4;
  const int tile_base = bx * NW * a.tiles_per_wave;
  const int tile_end = min(tile_base + NW * a.tiles_per_wave, a.ntiles);

  // A tile: 32 pixel rows pm = 0..31 <-> pixels tile*30 - 1 + pm of the task; rows 0 and 31 are halo (operands only).
  struct TileSt { unsigned offc[2][3]; unsigned bmf, bml; };
  auto decode = [&](int tl) {
    TileSt t;
    unsigned bf[2], bl[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const unsigned pix = (unsigned)(tl * 30 - 1 + 16 * mb + n);       // tile 0, row 0: "-1" wraps to an invalid pixel
      unsigned nimg, rem, oy, ox;
      divmod24(pix, hw, rhw, nimg, rem);
      divmod24(rem, (unsigned)W, rw, oy, ox);
      const bool valid = tl < tile_end && pix < (unsigned)mpix;          // past the wave's last tile: every load reads out of range (zeros)
      const bool rowok[3] = {valid && oy >= 1u, valid, valid && oy + 1u < (unsigned)H};
      const unsigned base = pix * (unsigned)(CI * 4) + (unsigned)(g * 32);
#pragma unroll
      for (int d = 0; d < 3; ++d) t.offc[mb][d] = rowok[d] ? base + (unsigned)((d - 1) * wci) : MI_OOB;
      // image-column padding: the sum P-[pm - 1] must not reach an output pixel in the first column, P+[pm + 1] not one in the last
      // (bit pm of a wave-uniform mask; the pixel just past the task's end decodes as a first column, which also keeps the sums of
      // rows past the end exact zeros)
      bf[mb] = (unsigned)__builtin_amdgcn_ballot_w64(ox == 0u) & 0xffffu;
      bl[mb] = (unsigned)__builtin_amdgcn_ballot_w64(ox + 1u == (unsigned)W) & 0xffffu;
    }
    t.bmf = bf[0] | (bf[1] << 16);
    t.bml = bl[0] | (bl[1] << 16);
    return t;
  };
  floatx4 raw[3][4];                                             // [ring slot][mb * 2 + 16-byte half]
  auto issue_step = [&](const TileSt& t, int st) {
    const int term = st / (3 * NCC), d = (st % (3 * NCC)) / NCC, cc = st % NCC;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const unsigned off = t.offc[mb][d] + (unsigned)(cc * 128);
      raw[st % 3][mb * 2 + 0] = buf_ld16(rin[term], off);
      raw[st % 3][mb * 2 + 1] = buf_ld16(rin[term], off + 16);
    }
  };
  // unit (weights in LDS) of row-step st and horizontal displacement ddx
  auto unit_of = [](int st, int ddx) {
    const int term = st / (3 * NCC), ddy = (st % (3 * NCC)) / NCC - 1, cc = st % NCC;
    const int tap = (MODE == 0) ? (ddy + 1) * 3 + (ddx + 1) : (1 - ddy) * 3 + (1 - ddx);
    return (term * 9 + tap) * NCC + cc;
  };
  const mi_u32x4* l4 = reinterpret_cast<const mi_u32x4*>(lds) + lane;
  mi_u32x4 bq[3][2];                                             // weight-plane ring: [slot][nb]
  // chunk c of a tile (9 per row-step): (row-step, tap index x = 0..2 <-> ddx = x - 1, plane order k = 0: l, 1: m, 2: h)
  auto read_chunk = [&](int c) {
    const int cw = c % (NS * 9);                                  // past the tile's last chunk: the next tile's first ones (same weights)
    const int st = cw / 9, x = (cw % 9) / 3, k = cw % 3;
    const int u = unit_of(st, x - 1), plane = 2 - k;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) bq[c % 3][nb] = l4[((u * 3 + plane) * 2 + nb) * 64];
  };

  int tile = tile_base + wave;
  TileSt cur = decode(tile);
  issue_step(cur, 0);
  issue_step(cur, 1);
  __syncthreads();                                               // weights staged (the first operand loads are already in flight)
  B16_ST(if (stp) stp[1] = __builtin_amdgcn_s_memtime();)
  Bf16Planes pl[2][2];                                           // [buffer][mb]
  bf16_split_pair<0>(raw[0][0], pl[0][0]); bf16_split_pair<1>(raw[0][0], pl[0][0]); bf16_split_pair<2>(raw[0][1], pl[0][0]); bf16_split_pair<3>(raw[0][1], pl[0][0]);
  bf16_split_pair<0>(raw[0][2], pl[0][1]); bf16_split_pair<1>(raw[0][2], pl[0][1]); bf16_split_pair<2>(raw[0][3], pl[0][1]); bf16_split_pair<3>(raw[0][3], pl[0][1]);
  read_chunk(0);
  read_chunk(1);
  // Stagger (MI355X_MICROARCH.md, "Two waves per SIMD" 9).  The two waves of a SIMD run the same program on equal work and start together, so
  // they stay in LOCKSTEP: both in the K loop (sharing the matrix pipe), then both in their epilogues -- lane rotations, 100+ vector
  // instructions, 64 fp64 operations, stores -- with the pipe idle (SQ counters of the two-term forward: matrix pipe busy 62 % of the launch
  // where the K loop alone sustains 86 %).  A phase offset between the partners is neutrally stable (whichever wave is in its epilogue, the
  // other has the pipe to itself for exactly that long), so ONE delay of half a tile's matrix time at the start keeps one wave's epilogue
  // under the other's K loop for the rest of the launch.  Partners: waves w and w + 4 of an 8-wave workgroup; with 4-wave workgroups (two
  // per CU) the workgroups of the second half of the launch order, which land on the CUs the first half already occupies.
  const unsigned total_wg = gridDim.x * gridDim.y * gridDim.z;
  const unsigned lin_wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  const bool late = NW == 8 ? wave >= 4 : lin_wg >= (total_wg >> 1);     // the second (younger) wave of its SIMD
  int prio_phase = late ? 1 : 0;
  if (a.stagger & 1) {
    // one tile's matrix time, NS x 72 MFMAs x 16 cycles = half the period of two waves that share the pipe; s_sleep counts units of 64 cycles
    if (late) __builtin_amdgcn_s_sleep(NS * 18 > 127 ? 127 : NS * 18);
  }
  const unsigned lane_out = (unsigned)((cbase + 2 * n) * 4);
  const unsigned addr_dn = (unsigned)(((lane - 16) & 63) * 4), addr_up = (unsigned)(((lane + 16) & 63) * 4);
  int g0 = g == 0 ? -1 : 0, g3 = g == 3 ? -1 : 0;                // lane-group masks (opaque: merges stay v_bfi / v_and, never compare + v_cndmask)
  asm volatile("" : "+v"(g0), "+v"(g3));

  // One tile.  PAR: which plane buffer holds the tile's first row-step (the buffers alternate per row-step; NS may be odd).
  auto tile_body = [&](auto par_c) {
    constexpr int PAR = decltype(par_c)::value;
    // (a.stagger & 2) issue priority alternates between the SIMD's two waves from tile to tile: at equal priority the older wave's vector
    // instructions go first every time, its K loop runs 10.3k cycles per two-term tile against the younger wave's 14.5k, it finishes a quarter
    // of the launch early and the SIMD ends on one wave (tools/conv_b16_stamps.py) -- taking turns, both finish together
    if (a.stagger & 2) {
      if (prio_phase & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
      ++prio_phase;
    }
    B16_ST(st_a = __builtin_amdgcn_s_memtime();)
    const TileSt nxt = decode(tile + NW);
    floatx4 acc[3][2][2];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[x][mb][nb] = floatx4{0.f, 0.f, 0.f, 0.f};
    float sr0 = 0.f, sr1 = 0.f;                                   // residuals of the pair being split
    // Element (mb, nb, r) of the tile's results: pixel row pm = 16 mb + 4g + r, channel cbase + 2n + nb.
    const unsigned tbase = (unsigned)(tile * 30 - 1) * (unsigned)(CO * 4) + lane_out;   // byte offset of pixel row 0's channel pair (tile 0: wraps; row 0 is dropped)
    auto row_off = [&](int mb, int r) { return tbase + (unsigned)((16 * mb + 4 * g + r) * CO * 4); };
    floatx2 zpre[2][4];
    B16_ST(st_b = st_a;)
    // (block-below reduction) block 1's tensors at this lane's output positions travel in groups of BGR pixel rows.  (Requesting the first
    // group one row-step early, like z, was measured and is not kept: one-term dgrad -0.1 % instead of -2.4 % against the 32x32x16 kernel,
    // two-term -6.2 % instead of -8.0 % -- 16 more live registers in the loop cost more than the latency they hide.)
    constexpr int BGR = NTERMS == 2 ? 2 : 4;
    struct Grp { floatx2 pp[BGR], zz[BGR], zd[BGR], dq[BGR]; unsigned pb[BGR]; };
    auto bred_fetch = [&](auto arg_c, int grp, Grp& gq) {
      constexpr bool ARG = decltype(arg_c)::value;
#pragma unroll
      for (int rr = 0; rr < BGR; ++rr) {
        const int e = grp * BGR + rr, mb = e >> 2, r = e & 3;
        const unsigned of = row_off(mb, r);
        if constexpr (ARG) gq.pb[rr] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rbp, of >> 2, 0, 0);    // two bytes: channels 2n, 2n + 1
        else gq.pp[rr] = buf_ld8(rbp, of);
        gq.zz[rr] = buf_ld8(rbzh, of);
        if (NTERMS == 2) { gq.zd[rr] = buf_ld8(rbzhd, of); gq.dq[rr] = buf_ld8(rbdp, of); }
      }
    };
    Grp ga;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      if (st + 2 < NS) issue_step(cur, st + 2); else issue_step(nxt, st + 2 - NS);
      if (EPI == EPI_TSTATS && st == NS - 1) {
        // z at this lane's 16 output positions, requested one row-step (72 MFMAs) before the epilogue reads it: the registers are the raw
        // row of the step that is running, dead since its split
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r) zpre[mb][r] = buf_ld8(rz, row_off(mb, r));
      }
      const Bf16Planes* pa = pl[(st + PAR) & 1];
      Bf16Planes* pn = pl[(st + 1 + PAR) & 1];
      const floatx4* rn = raw[(st + 1) % 3];
      // stage sq (0..39) of the next row-step's split: pair sq / 5 = (mb, P), values 2P, 2P + 1 of the block's eight
      auto stage = [&](int sq) {
        const int pr = sq / 5, sg = sq % 5, mb = pr >> 2, P = pr & 3;
        const floatx4& x = rn[mb * 2 + (P >> 1)];
        const float x0 = x[(P & 1) * 2], x1 = x[(P & 1) * 2 + 1];
        Bf16Planes& d = pn[mb];
        // (the subtractions are volatile assembly: they anchor their stage between the MFMAs -- instruction selection otherwise sinks the
        // arithmetic to its use -- without an empty pin statement behind them, which costs an s_nop each on gfx950)
        if (sg == 0) {
          d.h[P] = __builtin_bit_cast(unsigned, __builtin_convertvector(floatx2{x0, x1}, bf16x2));
          sr0 = bf16_sub_v(x0, __uint_as_float(d.h[P] << 16));
        } else if (sg == 1) {
          sr1 = bf16_sub_v(x1, __uint_as_float(d.h[P] & 0xffff0000u));
        } else if (sg == 2) {
          d.m[P] = __builtin_bit_cast(unsigned, __builtin_convertvector(floatx2{sr0, sr1}, bf16x2));
          sr0 = bf16_sub_v(sr0, __uint_as_float(d.m[P] << 16));
        } else if (sg == 3) {
          sr1 = bf16_sub_v(sr1, __uint_as_float(d.m[P] & 0xffff0000u));
        } else {
          // (sr1 was written at least one MFMA earlier: the wait state v_cvt_pk_bf16_f32 needs behind a VALU write of its source has passed)
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d.l[P]) : "v"(sr0), "v"(sr1));
        }
      };
      // MFMA j of the row-step (0..71), and behind it the split stage floor(j * 40 / 72) when that is a new one
      auto mm = [&](int j, const unsigned (&ap)[4], const mi_u32x4& b, floatx4& c) {
        __builtin_amdgcn_sched_barrier(0);
        c = MI_B16_MFMA(ap, b, c);
        __builtin_amdgcn_sched_barrier(0);
        const int s0 = j * 40 / 72, s1 = (j + 1) * 40 / 72;
        if (s1 > s0) stage(s0);
      };
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const int cb = (st * 3 + x) * 3, j0 = x * 24;
        // l-plane chunk: h x l
        read_chunk(cb + 2);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) mm(j0 + mb * 2 + nb, pa[mb].h, bq[cb % 3][nb], acc[x][mb][nb]);
        // m-plane chunk: m x m, h x m
        read_chunk(cb + 3);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) mm(j0 + 4 + mb * 2 + nb, pa[mb].m, bq[(cb + 1) % 3][nb], acc[x][mb][nb]);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) mm(j0 + 8 + mb * 2 + nb, pa[mb].h, bq[(cb + 1) % 3][nb], acc[x][mb][nb]);
        // h-plane chunk: l x h, m x h, h x h
        read_chunk(cb + 4);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) mm(j0 + 12 + mb * 2 + nb, pa[mb].l, bq[(cb + 2) % 3][nb], acc[x][mb][nb]);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) mm(j0 + 16 + mb * 2 + nb, pa[mb].m, bq[(cb + 2) % 3][nb], acc[x][mb][nb]);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) mm(j0 + 20 + mb * 2 + nb, pa[mb].h, bq[(cb + 2) % 3][nb], acc[x][mb][nb]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    B16_ST(st_a = __builtin_amdgcn_s_memtime(); st_k += st_a - st_b;)

    // ---- epilogue
    // the halo rows (pm = 0: mb 0, g 0, r 0; pm = 31: mb 1, g 3, r 3) are neither stored nor summed
    auto keep_mask = [&](int mb, int r) { return (mb == 0 && r == 0) ? ~g0 : ((mb == 1 && r == 3) ? ~g3 : -1); };
    // out[pm] = P0[pm] + P-[pm - 1] + P+[pm + 1]: rows r = 1..3 (r = 0..2) take the neighbouring register, row 0 (3) the last (first)
    // register of the lane group below (above): a rotation of the wave by 16 lanes, block 1's group 0 taking block 0's group 3
    float o[2][2][4];
    const bool masked = (cur.bmf | cur.bml) != 0u;              // wave-uniform: an image row begins / ends inside the tile
    int mf[2][4], ml[2][4];                                     // -1 where this lane's pixel row (mb, r) sits in the image's first / last column
    if (masked) {
      const unsigned vf = cur.bmf >> (4 * g), vl = cur.bml >> (4 * g);   // bit 16 mb + r: this lane's pixel row (mb, r)
#define MI_B16_ROWMASK(MB, R) mf[MB][R] = lane_mask_bit<16 * MB + R>(vf); ml[MB][R] = lane_mask_bit<16 * MB + R>(vl);
      MI_B16_ROWMASK(0, 0) MI_B16_ROWMASK(0, 1) MI_B16_ROWMASK(0, 2) MI_B16_ROWMASK(0, 3)
      MI_B16_ROWMASK(1, 0) MI_B16_ROWMASK(1, 1) MI_B16_ROWMASK(1, 2) MI_B16_ROWMASK(1, 3)
#undef MI_B16_ROWMASK
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float am[2][4], bp[2][4];
      const float s_dn1 = lane_select(g3, acc[0][0][nb][3], acc[0][1][nb][3]);      // block 1, group 0 <- block 0, group 3 (the SOURCE lanes of group 3 send block 0's row)
      const float s_up0 = lane_select(g0, acc[2][1][nb][0], acc[2][0][nb][0]);      // block 0, group 3 <- block 1, group 0
      // (the rotated values go through an opaque copy first: handed a matrix accumulator's element directly, hipcc 7.2 rotated element 0
      // of the accumulator instead of the one named -- found by tools/conv_b16_debug.py, checked in the ISA)
      auto rot = [](unsigned addr, float v) {
        int i = __builtin_bit_cast(int, v);
        asm volatile("" : "+v"(i));
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)addr, i));
      };
      am[0][0] = rot(addr_dn, acc[0][0][nb][3]);
      am[1][0] = rot(addr_dn, s_dn1);
      bp[0][3] = rot(addr_up, s_up0);
      bp[1][3] = rot(addr_up, acc[2][1][nb][0]);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int r = 1; r < 4; ++r) am[mb][r] = acc[0][mb][nb][r - 1];
#pragma unroll
        for (int r = 0; r < 3; ++r) bp[mb][r] = acc[2][mb][nb][r + 1];
      }
      if (masked) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            am[mb][r] = lane_zero_where(mf[mb][r], am[mb][r]);
            bp[mb][r] = lane_zero_where(ml[mb][r], bp[mb][r]);
          }
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[mb][nb][r] = acc[1][mb][nb][r] + (am[mb][r] + bp[mb][r]);
    }
    B16_ST(const unsigned long long st_c = __builtin_amdgcn_s_memtime(); st_comb += st_c - st_a;)
    // The statistics: this lane's 8 values per channel are summed in fp32 and the tile's two partial sums go into the fp64 accumulators -- 4
    // double-precision instructions per tile instead of 48 to 128.  While the partner wave of the SIMD is in its K loop an fp64 instruction
    // costs this wave about 16 cycles (tools/conv_b16_stamps.py with -DMI_B16_EXP: the statistics were 1400-1700 cycles of a tile's epilogue).
    // A partial sum of 8 terms carries a rounding of a few 2^-24 of itself, below that of the 288 products behind each term.
    float ts[2] = {0.f, 0.f}, tq[2] = {0.f, 0.f};
    if (EPI == EPI_BRED) {
      // block 1's pooled-resolution tensors at this lane's 16 output positions: "ReLU on" from the argmax byte (below 4) or from p > 0;
      //   1 term : sum [on] out zh, sum [on] out        2 terms: sum [on] (out zh + dp zhd), sum [on] out
      // One code path per source of "ReLU on" (a wave-uniform choice per launch, not per element); all of a group's loads are issued
      // before anything of it is used (the argmax bytes are unpacked where they are consumed).
      auto bred = [&](auto arg_c) {
        constexpr bool ARG = decltype(arg_c)::value;
        constexpr int GR = BGR;
        auto fetch = [&](int grp, Grp& gq) { bred_fetch(arg_c, grp, gq); };
        auto consume = [&](int grp, const Grp& gq) {
#pragma unroll
          for (int rr = 0; rr < GR; ++rr) {
            const int e = grp * GR + rr, mb = e >> 2, r = e & 3;
            const int keep = keep_mask(mb, r);
            buf_st2_untracked(rout_raw, keep ? row_off(mb, r) : MI_OOB, o[mb][0][r], o[mb][1][r]);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
              const float v = o[mb][nb][r];
              // -1 where the ReLU is on: the byte is below 4 (bits 2 and 10 of the pair clear) / p > 0 (0 - p carries a sign bit)
              int on = ARG ? ~lane_mask_bit_rt(gq.pb[rr], nb ? 10 : 2) : lane_mask_negative(0.f - gq.pp[rr][nb]);
              on &= keep;
              const float vv = lane_keep_where(on, v);
              if (NTERMS == 1) {
                ts[nb] = __builtin_fmaf(vv, gq.zz[rr][nb], ts[nb]);
              } else {
                const float dv = lane_keep_where(on, gq.dq[rr][nb]);
                ts[nb] = __builtin_fmaf(dv, gq.zd[rr][nb], __builtin_fmaf(vv, gq.zz[rr][nb], ts[nb]));
              }
              tq[nb] += vv;
            }
          }
        };
        constexpr int NG = 8 / GR;
        Grp gb;
        fetch(0, ga);
#pragma unroll
        for (int grp = 0; grp < NG; grp += 2) {
          if (grp + 1 < NG) fetch(grp + 1, gb);
          consume(grp, ga);
          if (grp + 2 < NG) fetch(grp + 2, ga);
          if (grp + 1 < NG) consume(grp + 1, gb);
        }
      };
      if (bred_arg) bred(std::true_type{}); else bred(std::false_type{});
    } else {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int keep = keep_mask(mb, r);
#if !(MI_B16_EXP & 1)
          buf_st2_untracked(rout_raw, keep ? row_off(mb, r) : MI_OOB, o[mb][0][r], o[mb][1][r]);
#endif
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const float v = lane_keep_where(keep, o[mb][nb][r]);
#if MI_B16_EXP & 2
            asm volatile("" :: "v"(v));
            if (EPI == EPI_TSTATS) asm volatile("" :: "v"(zpre[mb][r][nb]));
            continue;
#endif
            if (EPI == EPI_STATS) {
              ts[nb] += v;
              tq[nb] = __builtin_fmaf(v, v, tq[nb]);
            } else if (EPI == EPI_TSTATS) {
              const float zh = bn_zh(zpre[mb][r][nb], mu_c[nb], r_c[nb]);
              ts[nb] += v;
              tq[nb] = __builtin_fmaf(zh, v, tq[nb]);
            }
          }
        }
    }
    if (EPI != EPI_NONE) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) { s[nb] += (double)ts[nb]; q[nb] += (double)tq[nb]; }
    }
    cur = nxt;
    B16_ST(st_b = __builtin_amdgcn_s_memtime(); st_epi += st_b - st_a; ++st_n;)
  };

  // plane-buffer parity of a tile's first row-step: alternates from tile to tile when NS is odd
  if constexpr (NS % 2 == 0) {
    for (; tile < tile_end; tile += NW) tile_body(std::integral_constant<int, 0>{});
  } else {
    for (; tile < tile_end; tile += 2 * NW) {
      tile_body(std::integral_constant<int, 0>{});
      tile += NW;
      if (tile < tile_end) tile_body(std::integral_constant<int, 1>{});
      tile -= NW;
    }
  }
  B16_ST(if (stp) { stp[2] = st_comb; stp[3] = st_k; stp[4] = st_epi; stp[5] = __builtin_amdgcn_s_memtime(); stp[7] = st_n; })
  if (EPI != EPI_NONE) {
    double* pb = a.partial + (size_t)task * gridDim.x * 2 * CO;
    stats_block_reduce_pairs(s[0], q[0], s[1], q[1], reinterpret_cast<double*>(lds), lane, wave, NW, pb, CO, cbase, a.fin, task, bx);
  }
  B16_ST(if (stp) stp[6] = __builtin_amdgcn_s_memtime();)
}

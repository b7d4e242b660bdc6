// Shared device/host helpers for libmi_maml (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define MI_WAVE 64
#define MI_BN_EPS 1e-5

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// ---- launch geometry of one ConvBlock as the kernels see it
struct ConvGeom {
  int n;        // images per task
  int h, w;     // conv input spatial size
  int ho, wo;   // conv output spatial size
  int ci, co;   // reduction / output channels of THIS op
  int stride;   // 1 or 2 (pad is always 1, kernel 3x3)
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// u = gamma * ((z - mu) * rstd) + beta, written so that every kernel evaluates it with the same roundings
// (the pooling argmax / ReLU mask is RE-computed from z in backward and tangent kernels, never stored).
__device__ __forceinline__ float bn_zh(float z, float mu, float r) { return (z - mu) * r; }
__device__ __forceinline__ float bn_u(float zh, float g, float b) { return fmaf(g, zh, b); }

// Per-lane selects as bit-field operations.  v_cndmask_b32 issues at a QUARTER of the plain VALU rate on gfx950 (16 cycles, and a
// v_cmp in front of it: tools/valu_rate_probe.hip, profiles/r3/valu_rate_probe.txt), and on the fp32 matrix pipe VALU time adds
// to MFMA time; a lane mask kept in a VGPR (0 / -1) and v_bfi_b32 / v_and_b32 do the same select at full rate.
// Only the MASKS are inline assembly (so that instruction selection cannot fold mask-and-merge back into compare + v_cndmask); the
// merges are plain C on the bits.  That split matters: the hazard recogniser does not look inside inline assembly, so an assembly
// instruction must neither read a matrix result nor produce a matrix operand (MFMA -> VALU and VALU -> MFMA wait states are the
// compiler's to insert) -- the masks read ordinary VALU / load results and feed ordinary VALU instructions only.
// ONE documented exception: the epilogue stores of the split-bf16 stride-1 convolutions (conv_mfma.hip, buf_st_untracked) read the
// accumulators from assembly, behind an explicit 24-wait-state s_nop fence tied to them; tools/asm_hazard_scan.py checks that fence on
// every build (tests/test_asm_hazards.py).  The fp32-pipe variants of the same kernel use the compiler-visible store.
__device__ __forceinline__ int lane_mask_negative(float d) {            // -1 where d's sign bit is set
  int r;
  asm("v_ashrrev_i32 %0, 31, %1" : "=v"(r) : "v"(d));
  return r;
}
template <int BIT> __device__ __forceinline__ int lane_mask_bit(unsigned v) {   // -1 where bit BIT of v is set
  int r;
  asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r) : "v"(v), "n"(BIT));
  return r;
}
__device__ __forceinline__ unsigned lane_select(int mask, unsigned a, unsigned b) {   // mask ? a : b  (v_bfi_b32)
  return ((unsigned)mask & a) | (~(unsigned)mask & b);
}
__device__ __forceinline__ float lane_select(int mask, float a, float b) {
  return __builtin_bit_cast(float, lane_select(mask, __builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)));
}
// The same merge as ONE v_bfi_b32 in inline assembly (instruction selection does not always fuse the C form).  Per the rule above:
// a and b must be results of ordinary VALU instructions (never matrix accumulators) and the result must not be a matrix operand.
__device__ __forceinline__ float lane_select_valu(int mask, float a, float b) {
  float r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned lane_select_valu(int mask, unsigned a, unsigned b) {
  unsigned r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float lane_zero_where(int mask, float b) {          // mask ? 0 : b
  return __builtin_bit_cast(float, ~(unsigned)mask & __builtin_bit_cast(unsigned, b));
}
__device__ __forceinline__ float lane_keep_where(int mask, float b) {          // mask ? b : 0
  return __builtin_bit_cast(float, (unsigned)mask & __builtin_bit_cast(unsigned, b));
}

// Out-of-image operands are fetched from this zero word by selecting the ADDRESS (never the loaded value): a predicated load
// costs an exec-mask branch region, and a select on the loaded value makes the wave wait for the load before it can issue
// the MFMAs of the previous, already loaded, tile.
static __device__ __attribute__((aligned(64))) float mi_zero_word[16] = {0.f};   // not const: must live in the global address space

// Operands come through raw buffer loads: one descriptor per (tensor, task), 32-bit byte offsets, and the hardware range
// check returns 0 for any offset >= the task's tensor size -- so image padding costs no predicated loads and no selects:
// an invalid ROW poisons the lane's row offset with OOB, an invalid COLUMN poisons the (wave-uniform, scalar) column addend.
#define MI_OOB 0x40000000u   // >= any per-task tensor size; OOB + OOB does not wrap
typedef __amdgpu_buffer_rsrc_t mi_rsrc;
__device__ __forceinline__ float buf_ld(mi_rsrc r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

typedef unsigned int mi_u32x4 __attribute__((ext_vector_type(4)));
// NOTE: the loaded vector must be re-typed as a WHOLE (bit_cast to floatx4).  Extracting the four lanes of the integer vector
// one by one (bit_cast(float, v.x) ...) makes hipcc 7.2 narrow the instruction to buffer_load_dword and leave three of the four
// values undefined (reproduced in isolation; this is the "miscompiled b128" of round 1).
__device__ __forceinline__ floatx4 buf_ld16(mi_rsrc r, unsigned off) {
  return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ void buf_st(mi_rsrc r, unsigned off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
}

// Largest-magnitude cells (the fp16 operand form's per-(task, tensor) scale, bf16_split.h): a producer folds the bit patterns of |x| of
// the values it writes into `m` (mi_amax_acc), reduces over the wave and commits ONE atomic max per wave at the end of the kernel.
// Unsigned maxima of non-negative fp32 bit patterns: order-independent, so the cell -- and everything scaled by it -- is reproducible.
__device__ __forceinline__ void mi_amax_acc(unsigned& m, float v) {
  const unsigned b = __builtin_bit_cast(unsigned, v) & 0x7fffffffu;
  m = b > m ? b : m;
}
// Only the exponent is committed (all the scale needs).  Device-scope atomics on one line are served one at a time, ~12 ns each across
// the XCDs' L2s (one atomic per WAVE on 32 adjacent cells cost a 13 us kernel 100 us; one per workgroup on one line per task still cost
// 0.15 / 0.27 ms per cfg2 iteration at 1 / 4 tasks per call, where a task's launch has up to 2048 workgroups).  Hence: one atomic per
// WORKGROUP (waves meet in LDS; every thread of the workgroup must call mi_amax_commit), and a (task, tensor) cell is MI_CELL_SUB
// sub-cells, each alone in a 256-byte line, a workgroup committing to sub-cell (blockIdx.x + blockIdx.z) mod MI_CELL_SUB.  A reader takes
// the maximum of the sub-cells (mi_cell_read: one load per lane, four lane exchanges).
#define MI_CELL_STRIDE 64                         // words between sub-cells
#define MI_CELL_SUB 16
#define MI_CELL_WORDS (MI_CELL_SUB * MI_CELL_STRIDE)   // words per (task, tensor)
__device__ __forceinline__ void mi_amax_commit(unsigned m, unsigned* cells, int task) {
  __shared__ unsigned mi_amax_red[16];
  m &= 0x7f800000u;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o, 64); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0) mi_amax_red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (int)((blockDim.x + 63) >> 6);
    for (int w = 1; w < nw; ++w) m = mi_amax_red[w] > m ? mi_amax_red[w] : m;
#ifndef MI_AMAX_DBG_NOATOMIC      /* timing experiment (wrong results): the commit without its atomic */
    // (a workgroup whose exponent does not exceed what its sub-cell already holds skips the atomic: a launch of 2048 short workgroups
    // otherwise ends in 2048 atomics, ~3.5 us of them -- 0.27 ms per cfg2 iteration at 4 tasks per call; a stale read only costs an atomic)
    unsigned* c = cells + (size_t)task * MI_CELL_WORDS + ((blockIdx.x + blockIdx.z) & (MI_CELL_SUB - 1)) * MI_CELL_STRIDE;
    if (m > __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(c, m);
#endif
  }
}
// the cell of `task`, in two steps so that the load (a miss: the atomics ran at the memory side) is in flight while the kernel issues
// its other prologue loads: mi_cell_fetch at the top of the kernel (every lane of the wave), mi_cell_fold where the value is needed
// (the result is wave-uniform)
__device__ __forceinline__ unsigned mi_cell_fetch(const unsigned* cells, int task) {
  return cells[(size_t)task * MI_CELL_WORDS + (threadIdx.x & (MI_CELL_SUB - 1)) * MI_CELL_STRIDE];
}
__device__ __forceinline__ unsigned mi_cell_fold(unsigned m) {
#pragma unroll
  for (int o = MI_CELL_SUB / 2; o >= 1; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o, 64); m = t > m ? t : m; }
  return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// BatchNorm + ReLU + MaxPool window analysis shared by the streaming kernels of bn_pool.hip and the one-workgroup-per-task tail of
// tail.hip: one thread owns a 2x2 pooling window (or one pixel without pooling) x 4 channels; the ReLU mask and pooling argmax are
// re-derived from z with bit-identical arithmetic (bn_zh / bn_u) wherever they are needed.
#pragma once
#include "mi_common.h"
#include "kernels.h"

struct ChanConst {
  float mu[4], r[4], g[4], b[4];
};

__device__ __forceinline__ void load4(const float* p, float* o) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void store4(float* p, const float* o) {
  *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}

__device__ __forceinline__ ChanConst load_consts(const BnArgs& a, int task, int c0) {
  ChanConst k;
  load4(a.mu + (size_t)task * a.c + c0, k.mu);
  load4(a.rstd + (size_t)task * a.c + c0, k.r);
  load4(a.gamma + (size_t)task * a.pstride + c0, k.g);
  load4(a.beta + (size_t)task * a.pstride + c0, k.b);
  return k;
}

// Window geometry shared by all kernels.  POOL: windows tile ceil(ho/2) x ceil(wo/2) (so the dropped odd row/column of
// floor pooling is still visited for dz); !POOL: one position per "window".
template <int POOL>
struct WinIter {
  int hw2, ww2, nwin, hp, wp;
  __device__ WinIter(const BnArgs& a) {
    if (POOL) {
      hw2 = (a.ho + 1) >> 1; ww2 = (a.wo + 1) >> 1; hp = a.ho >> 1; wp = a.wo >> 1;
    } else {
      hw2 = a.ho; ww2 = a.wo; hp = a.ho; wp = a.wo;
    }
    nwin = a.n * hw2 * ww2;
  }
};

// Analyse one window for 4 channels: zh and u at its (up to 4) positions, first-max argmax.
template <int POOL>
struct Window {
  static constexpr int NP = POOL ? 4 : 1;
  size_t off[NP];     // element offset of each position's channel quad inside the task's z
  bool exists[NP];
  bool pooled;        // window produces a pooled output (inside the floor-pooled grid)
  size_t poff;        // offset of the pooled output quad
  float zh[NP][4], u[NP][4];
  int arg[4];
  float umax[4];
  bool better[NP][4];

  __device__ __forceinline__ void locate(const BnArgs& a, const WinIter<POOL>& it, int win, int c0) {
    const int n = win / (it.hw2 * it.ww2);
    const int rem = win - n * it.hw2 * it.ww2;
    const int wy = rem / it.ww2, wx = rem - wy * it.ww2;
    if (POOL) {
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const int y = 2 * wy + (k >> 1), x = 2 * wx + (k & 1);
        exists[k] = (y < a.ho) && (x < a.wo);
        off[k] = ((size_t)(n * a.ho + y) * a.wo + x) * a.c + c0;
      }
      pooled = (wy < it.hp) && (wx < it.wp);
      poff = ((size_t)(n * it.hp + wy) * it.wp + wx) * a.c + c0;
    } else {
      exists[0] = true;
      off[0] = ((size_t)(n * a.ho + wy) * a.wo + wx) * a.c + c0;
      pooled = true;
      poff = off[0];
    }
  }
  __device__ __forceinline__ void analyse(const float* __restrict__ z_t, const ChanConst& k) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      float zv[4] = {0.f, 0.f, 0.f, 0.f};
      if (exists[p]) load4(z_t + off[p], zv);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        zh[p][c] = bn_zh(zv[c], k.mu[c], k.r[c]);
        u[p][c] = bn_u(zh[p][c], k.g[c], k.b[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int best = 0;
      float bu = u[0][c];
#pragma unroll
      for (int p = 1; p < NP; ++p) {
        better[p][c] = u[p][c] > bu;                      // strict '>' keeps the first maximum (torch max_pool2d order)
        if (better[p][c]) { bu = u[p][c]; best = p; }
      }
      arg[c] = best;
      umax[c] = bu;
    }
  }
  // value of a per-position quantity at the argmax position of channel c.  A chain of selects on the comparison flags
  // (the last position that beat the running maximum wins); selecting on `arg[c] == p` instead lets LLVM re-roll the
  // chain into a dynamically indexed scratch array.
  __device__ __forceinline__ float at_arg(const float (&v)[NP][4], int c) const {
    float r = v[0][c];
#pragma unroll
    for (int p = 1; p < NP; ++p) r = better[p][c] ? v[p][c] : r;
    return r;
  }
};

// Streaming scan of one window: running maximum of u and the values of zh (and zd) AT the running maximum, carried as
// plain SSA values (no per-position arrays: selecting among stored positions by index makes LLVM spill them to scratch).
// scan_values: the scan on values already in registers (a caller that issues the loads of several windows before it waits for any);
// scan_window: loads + the same scan.
template <int POOL, bool WITH_ZD>
__device__ __forceinline__ void scan_values(const floatx4* zv, const floatx4* zdv, const ChanConst& k, floatx4& umax, floatx4& zh_at,
                                            floatx4& zd_at) {
#pragma unroll
  for (int p = 0; p < Window<POOL>::NP; ++p) {
    const floatx4 z = zv[p];
    floatx4 zd = {0.f, 0.f, 0.f, 0.f};
    if (WITH_ZD) zd = zdv[p];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float zh = bn_zh(z[c], k.mu[c], k.r[c]);
      const float u = bn_u(zh, k.g[c], k.b[c]);
      if (p == 0) {
        umax[c] = u; zh_at[c] = zh; zd_at[c] = zd[c];
      } else {
        const bool gt = u > umax[c];                      // strict '>' keeps the first maximum (torch max_pool2d order)
        umax[c] = gt ? u : umax[c];
        zh_at[c] = gt ? zh : zh_at[c];
        zd_at[c] = gt ? zd[c] : zd_at[c];
      }
    }
  }
}
template <int POOL, bool WITH_ZD>
__device__ __forceinline__ void scan_window(const Window<POOL>& w, const float* __restrict__ z_t, const float* __restrict__ zd_t,
                                            const ChanConst& k, floatx4& umax, floatx4& zh_at, floatx4& zd_at) {
  floatx4 zv[Window<POOL>::NP], zdv[Window<POOL>::NP];
#pragma unroll
  for (int p = 0; p < Window<POOL>::NP; ++p) {
    zv[p] = *reinterpret_cast<const floatx4*>(z_t + w.off[p]);
    if (WITH_ZD) zdv[p] = *reinterpret_cast<const floatx4*>(zd_t + w.off[p]);
  }
  scan_values<POOL, WITH_ZD>(zv, zdv, k, umax, zh_at, zd_at);
}

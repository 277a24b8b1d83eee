// mnk_pair_scan.h -- the win scan of the two-lanes-per-env rollout (gfx950 / MI355X only): lane 0 of a pair scans two
// of the four directions, lane 1 the other two, the shift amount being a per-lane VGPR.
#pragma once
#include "mnk_device.h"

// value of the partner lane (lane ^ 1): a DPP quad_perm [1,0,3,2] move, no LDS round trip
__device__ __forceinline__ uint32_t pair_swap(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
}

// x >>= (role ? S1 : S0): the two lanes of a pair shift by different compile-time amounts.  Where the
// word parts of the two amounts agree the word move is uniform and only the bit part (one v_alignbit_b32 per
// word, shift amount in a VGPR) differs per lane; where they differ a per-word select picks the source word.
template <int NW, int S0, int S1>
__device__ __forceinline__ void bs_shr_pair(uint32_t (&x)[NW], uint32_t role) {
  constexpr int Q0 = S0 >> 5, Q1 = S1 >> 5;
  uint32_t y[NW + 1];
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const uint32_t a = (w + Q0 < NW) ? x[w + Q0] : 0u;
    if (Q0 == Q1) y[w] = a;
    else y[w] = role ? ((w + Q1 < NW) ? x[w + Q1] : 0u) : a;
  }
  y[NW] = 0u;
  const uint32_t r = role ? (uint32_t)(S1 & 31) : (uint32_t)(S0 & 31);
#pragma unroll
  for (int w = 0; w < NW; ++w) x[w] = __builtin_amdgcn_alignbit(y[w + 1], y[w], r);
}

// run-doubling scan (see bs_has_run) with the direction stride D0 on role 0 and D1 on role 1; b = the plane
template <int NW, int CK, int D0, int D1, int LEN = 1>
__device__ __forceinline__ void bs_run_pair_steps(uint32_t (&x)[NW], const uint32_t (&b)[NW], uint32_t role) {
  if constexpr (2 * LEN <= CK) {
    uint32_t t[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) t[w] = x[w];
    bs_shr_pair<NW, LEN * D0, LEN * D1>(t, role);
#pragma unroll
    for (int w = 0; w < NW; ++w) x[w] &= t[w];
    bs_run_pair_steps<NW, CK, D0, D1, 2 * LEN>(x, b, role);
  } else if constexpr (LEN + 1 == CK) {  // one stone short: AND with the plane itself (see bs_has_run)
    uint32_t t[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) t[w] = b[w];
    bs_shr_pair<NW, LEN * D0, LEN * D1>(t, role);
#pragma unroll
    for (int w = 0; w < NW; ++w) x[w] &= t[w];
  } else if constexpr (LEN < CK) {
    uint32_t t[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) t[w] = x[w];
    bs_shr_pair<NW, (CK - LEN) * D0, (CK - LEN) * D1>(t, role);
#pragma unroll
    for (int w = 0; w < NW; ++w) x[w] &= t[w];
  }
}

template <int NW, int CK, int D0, int D1>
__device__ __forceinline__ uint32_t bs_run_bits_pair(const uint32_t (&b)[NW], uint32_t role) {
  uint32_t x[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) x[w] = b[w];
  bs_run_pair_steps<NW, CK, D0, D1>(x, b, role);
  uint32_t any = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) any |= x[w];
  return any;
}


// mnk_draw.h -- the masked categorical head + inverse-CDF draw as device building blocks (gfx950 only).
//
// alg/architectures/cnn.py:69-79 (= resnet.py:84-95, transformer.py:80-91: logits where(mask) -inf, all-masked rows ->
// zeros, Categorical) followed by dist.sample() / argmax(dist.logits) (selfplay/policy.py:46-52) and
// dist.log_prob(action) (alg/ppo.py:96-97).  Used by k_sample_logits (mnk_sample.hip: the draw as a launch of its own)
// and by the self-play step kernels that take logits instead of actions (mnk_selfplay_draw.hip: the draw folded into
// the step) -- the same code, so both give the same action for the same row, uniform and
// (LPR, K) shape.
//
//   1. slab_to_lds: a workgroup's contiguous slab of rows -> LDS with full-width loads (16 B of logits + the matching
//      mask bytes per lane), the mask applied on the way (illegal cell -> -inf): LDS holds one f32 per cell;
//   2. draw_row: LPR lanes per row (an aligned group inside one wave), lane s owning the interleaved cells s, s + LPR,
//      ...: row max by DPP butterfly, weights 2^((logit - max) * log2 e) (one FMA + one v_exp per cell; -inf gives an
//      exact 0), an LPR-lane inclusive scan, the uniform picks the point u * total on the cumulative axis (cells
//      ordered lane-major), a ballot finds the lane that holds it and a count of that lane's cells below the point
//      finds the cell.  If rounding leaves the point beyond the last cell's cumulative weight the last legal cell of
//      the last weighted lane is taken.
#pragma once
#include "mnk_device.h"

namespace mnk_draw {

constexpr float LOG2E = 1.4426950408889634f;

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

enum { DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_ROW_SHR = 0x110 };

// all-lanes-equal max / sum over aligned groups of LPR lanes (butterfly: every lane adds the same pairs)
template <int LPR>
__device__ __forceinline__ float group_max(float v) {
  if (LPR >= 2) v = fmaxf(v, dpp<DPP_XOR1>(v));
  if (LPR >= 4) v = fmaxf(v, dpp<DPP_XOR2>(v));
  if (LPR >= 8) v = fmaxf(v, dpp<DPP_HALF_MIRROR>(v));
  if (LPR >= 16) v = fmaxf(v, dpp<DPP_MIRROR>(v));
  if (LPR >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
  return v;
}

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
  if (LPR >= 2) v += dpp<DPP_XOR1>(v);
  if (LPR >= 4) v += dpp<DPP_XOR2>(v);
  if (LPR >= 8) v += dpp<DPP_HALF_MIRROR>(v);
  if (LPR >= 16) v += dpp<DPP_MIRROR>(v);
  if (LPR >= 32) v += __shfl_xor(v, 16, 64);
  return v;
}

// inclusive scan over the LPR lanes of a group; row_shr moves data up by `off` lanes inside a 16-lane DPP row
// (zero shifted in at the row's start), lanes whose source lies in the neighbouring group add nothing; a 32-lane
// group is two DPP rows, the upper one adds the lower one's total (its lane 15)
template <int LPR>
__device__ __forceinline__ float group_scan(float v, int sub) {
  if (LPR >= 2) { const float up = dpp<DPP_ROW_SHR + 1>(v); v += (sub >= 1) ? up : 0.0f; }
  if (LPR >= 4) { const float up = dpp<DPP_ROW_SHR + 2>(v); v += (sub >= 2) ? up : 0.0f; }
  if (LPR >= 8) { const float up = dpp<DPP_ROW_SHR + 4>(v); v += (sub >= 4) ? up : 0.0f; }
  if (LPR >= 16) { const float up = dpp<DPP_ROW_SHR + 8>(v); v += (sub >= 8) ? up : 0.0f; }
  if (LPR >= 32) { const float low = __shfl(v, 15, 32); v += (sub >= 16) ? low : 0.0f; }
  return v;
}

// elements per 16-byte load of the logits: f32 4, bf16 8; the uniform form walks 16 mask bytes at a time
template <typename LT> struct Slab { static constexpr int BYTES = (int)sizeof(LT), VE = 16 / BYTES; };
template <> struct Slab<void> { static constexpr int BYTES = 0, VE = 16; };

// floats of LDS a slab of `rows` rows of C cells needs (the slab itself + the cells the first, unaligned load drags in)
template <typename LT>
__host__ __device__ constexpr size_t slab_floats(int rows, int C) { return (size_t)rows * C + 2 * Slab<LT>::VE; }

// masked logits of the global cells [e0, e1) -> lds[e - a0] as f32, a0 = e0 rounded down to a multiple of VE.
// A lane takes VE consecutive cells: one 16-byte load of logits and one 4/8/16-byte load of their mask bytes.
// Cells in front of e0 that the first load drags in belong to the previous workgroup's rows and are not used.
template <typename LT>
__device__ __forceinline__ void slab_to_lds(const LT* logits, const uint8_t* mask, int64_t e0, int64_t e1,
                                            int64_t total, float* lds, bool vec, int tid, int nthreads) {
  constexpr int VE = Slab<LT>::VE;
  const float NEG = -__builtin_huge_valf();
  const int64_t a0 = e0 & ~(int64_t)(VE - 1);
  if (!vec) {  // unaligned base pointers: one cell per lane and trip
    for (int64_t c = e0 + tid; c < e1; c += nthreads) {
      float x = 0.0f;
      if constexpr (Slab<LT>::BYTES == 4) x = logits[c];
      if constexpr (Slab<LT>::BYTES == 2) x = __uint_as_float((uint32_t)logits[c] << 16);
      lds[c - a0] = mask[c] ? x : NEG;
    }
    return;
  }
  for (int64_t c = a0 + (int64_t)tid * VE; c < e1; c += (int64_t)nthreads * VE) {
    float* dst = lds + (c - a0);
    if (c + VE <= total) {
      float x[VE];
      uint32_t m[VE / 4];
      if constexpr (VE == 4) {
        const uint4 v = *reinterpret_cast<const uint4*>(logits + c);
        m[0] = *reinterpret_cast<const uint32_t*>(mask + c);
        x[0] = __uint_as_float(v.x); x[1] = __uint_as_float(v.y); x[2] = __uint_as_float(v.z); x[3] = __uint_as_float(v.w);
      } else if constexpr (VE == 8) {
        const uint4 v = *reinterpret_cast<const uint4*>(logits + c);
        const uint2 mm = *reinterpret_cast<const uint2*>(mask + c);
        m[0] = mm.x; m[1] = mm.y;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          x[2 * j] = __uint_as_float(w[j] << 16);
          x[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
        }
      } else {
        const uint4 mm = *reinterpret_cast<const uint4*>(mask + c);
        m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w;
#pragma unroll
        for (int j = 0; j < VE; ++j) x[j] = 0.0f;
      }
#pragma unroll
      for (int q = 0; q < VE / 4; ++q) {
        float4 o;
        o.x = (m[q] & 0x000000FFu) ? x[4 * q + 0] : NEG;
        o.y = (m[q] & 0x0000FF00u) ? x[4 * q + 1] : NEG;
        o.z = (m[q] & 0x00FF0000u) ? x[4 * q + 2] : NEG;
        o.w = (m[q] & 0xFF000000u) ? x[4 * q + 3] : NEG;
        *reinterpret_cast<float4*>(dst + 4 * q) = o;
      }
    } else {  // the last few cells of the whole array
      for (int j = 0; j < VE && c + j < total; ++j) {
        float x = 0.0f;
        if constexpr (Slab<LT>::BYTES == 4) x = logits[c + j];
        if constexpr (Slab<LT>::BYTES == 2) x = __uint_as_float((uint32_t)logits[c + j] << 16);
        dst[j] = mask[c + j] ? x : NEG;
      }
    }
  }
}

// the uniform in (0, 1) of a row: one Philox block per row (stream MNK_STREAM_SAMPLE)
__device__ __forceinline__ float row_uniform(uint64_t seed, uint64_t row_id, uint64_t step) {
  const uint32_t x = mnk_rand_u32(seed, row_id, step, MNK_STREAM_SAMPLE);
  return ((float)(x >> 8) + 0.5f) * 5.9604644775390625e-08f;
}

struct Drawn {
  int chosen;        // the cell, the same in all LPR lanes of the group
  float rowmax, total_w;
  bool none_legal;
  // log-probability of `chosen` under the masked softmax (f32 arithmetic); lrow = the row's masked logits in LDS
  __device__ __forceinline__ float logp(const float* lrow) const {
    return (none_legal ? 0.0f : lrow[chosen]) - rowmax - logf(total_w);
  }
};

// One row, by the LPR lanes of an aligned group (ALL lanes of the wave must call this: ballots inside); lane `tid` is
// lane tid % LPR of its group.  lrow: the row's masked logits in LDS (C floats); u: the row's uniform (not read when
// deterministic).  LPR lanes per row, K cells per lane; EXACT: LPR * (K - 1) < C, so only a lane's last cell can lie
// outside the row.
template <int LPR, int K, bool EXACT>
__device__ __forceinline__ Drawn draw_row(const float* lrow, int C, float u, int deterministic, int tid) {
  const int sub = tid % LPR;
  const float NEG = -__builtin_huge_valf();
  const int lane = tid & 63;
  const int gbase = lane & ~(LPR - 1);
  const unsigned long long gmask = (1ull << LPR) - 1ull;
  Drawn out;

  float l[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const int c = sub + LPR * j;
    if (EXACT && j < K - 1) {
      l[j] = lrow[c];
    } else {
      const bool in = c < C;
      const float x = lrow[in ? c : 0];
      l[j] = in ? x : NEG;
    }
  }
  float mx = l[0];
#pragma unroll
  for (int j = 1; j < K; ++j) mx = fmaxf(mx, l[j]);
  float rowmax = group_max<LPR>(mx);
  // a row without a legal cell (cnn.py:76-77: all-masked -> zeros -> uniform over all cells): practically never
  bool none_legal = false;
  if (__ballot(rowmax == NEG) != 0ull) {
    none_legal = rowmax == NEG;
#pragma unroll
    for (int j = 0; j < K; ++j) l[j] = none_legal ? ((sub + LPR * j < C) ? 0.0f : NEG) : l[j];
    rowmax = none_legal ? 0.0f : rowmax;
  }
  const float bias = -rowmax * LOG2E;
  float w[K];
  float mine = 0.0f;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    w[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(l[j], LOG2E, bias));  // 2^(-inf) = 0 for masked cells
    mine += w[j];
  }
  const float total_w = group_sum<LPR>(mine);
  int chosen;
  if (deterministic) {
    // argmax, ties -> lowest cell like torch.argmax (policy.py:48-49)
    float best = NEG;
    int arg = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < K; ++j)
      if (l[j] > best) { best = l[j]; arg = sub + LPR * j; }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) {
      const float ov = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(arg, off, 64);
      if (ov > best || (ov == best && oi < arg)) { best = ov; arg = oi; }
    }
    chosen = arg;
  } else {
    const float incl = group_scan<LPR>(mine, sub);
    const float target = u * total_w;
    // cells of this lane whose running sum stays at or below the target = index of the first one above it
    const float base = incl - mine;  // the lanes before this one
    float run = base;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      run += w[j];
      cnt += (run <= target) ? 1 : 0;
    }
    // The lane that holds the target: the lanes before it end at or below it and ITS OWN running sum -- the very values
    // the count above compared, not the scan's `incl`, which is the same sum in another association -- ends above it.
    // Then the count stops at a cell that moved the sum, i.e. a cell with weight: never a masked cell, never one of the
    // cells past the end of the row that the lane's last slot stands for.  (Round 3 tested `incl > target`: when the two
    // associations differed in the last bit and the target fell between them, the count ran through and the pick was
    // the lane's last slot -- cell C..C+LPR-2, an out-of-range action, about once in 10^7 draws on 3x3.)  If rounding
    // leaves no lane with both properties, the fallback below takes the last cell with weight.
    const unsigned long long pass = __ballot(run > target && base <= target && mine > 0.0f);
    const uint32_t pass_g = (uint32_t)((pass >> gbase) & gmask);
    int owner = __ffs(pass_g) - 1;
    if (__ballot(pass_g == 0u) != 0ull) {
      // rounding left the target at or beyond the total: the last cell with weight of the last lane with weight
      const unsigned long long heavy = __ballot(mine > 0.0f);
      const uint32_t heavy_g = (uint32_t)((heavy >> gbase) & gmask);
      int last = 0;
#pragma unroll
      for (int j = 0; j < K; ++j) last = (w[j] > 0.0f) ? j : last;
      if (pass_g == 0u) {
        owner = 31 - __clz(heavy_g | 1u);
        cnt = last;
      }
    }
    const int pick = sub + LPR * (cnt < K ? cnt : K - 1);
    chosen = __shfl(pick, owner, LPR);
  }
  out.chosen = chosen;
  out.rowmax = rowmax;
  out.total_w = total_w;
  out.none_legal = none_legal;
  return out;
}

// lanes per row x cells per lane of the boards with a compile-time shape, by measurement (9x9: 4 lanes per row 7.1 us,
// 8 lanes per row 9.9 us; profiles/r02_api_kernels.md).  The cell order of the inverse-CDF walk is lane-major, so two
// kernels give the same action for the same uniform only when they use the same shape: this table is the one place
// that names it.
// Every other row width takes its shape by size.  EXACT (only a lane's last cell can lie outside the row) holds for the
// measured shapes.  As constexpr functions of the row width, so that the host can size the LDS of a run-time compiled
// kernel whose Draw<LT, C> it never instantiates.
__host__ __device__ constexpr bool shape_measured(int C) { return C == 81 || C == 9 || C == 169 || C == 225 || C == 361; }
__host__ __device__ constexpr int shape_lpr(int C) {
  return (C == 81 || C == 9) ? 4 : (C == 169 ? 8 : ((C == 225 || C == 361) ? 16 : (C <= 32 ? 4 : (C <= 96 ? 8 : (C <= 256 ? 16 : 32)))));
}
__host__ __device__ constexpr int shape_k(int C) {
  return C == 81 ? 21 : (C == 9 ? 3 : (C == 169 ? 22 : (C == 225 ? 15 : (C == 361 ? 23 : (C <= 32 ? 8 : (C <= 96 ? 12 : (C <= 512 ? 16 : 32)))))));
}
template <int C> struct Shape {
  static constexpr int LPR = shape_lpr(C), K = shape_k(C);
  static constexpr bool EXACT = shape_measured(C);
};

}  // namespace mnk_draw

// mnk_sample.hip -- masked categorical head fused with the draw (gfx950 / MI355X only).
//
// Replaces alg/architectures/cnn.py:69-79 (= resnet.py:84-95, transformer.py:80-91: logits where(mask) -inf,
// all-masked rows -> zeros, Categorical) followed by dist.sample() / argmax(dist.logits)
// (selfplay/policy.py:46-52) and dist.log_prob(action) (alg/ppo.py:96-97).
//
// HBM traffic is 5C + 12 bytes per row (f32 logits; 3C + 12 with bf16 logits, C + 12 for the uniform form) and
// nothing is reused.  A row of logits is C*4 bytes with C odd on every board people play (81, 169, 225, 361), so
// rows are not 16-byte aligned and lanes that own "their" cells directly would issue strided scalar loads (the
// round-1 kernel: 1.6 TB/s).  Here a 256-thread workgroup owns ROWS consecutive rows = one contiguous slab:
//   1. slab -> LDS with full-width loads (16 B of logits + the matching mask bytes per lane), the mask applied on
//      the way (illegal cell -> -inf), so LDS holds one f32 per cell and the mask is never looked at again;
//      the first ROWS threads meanwhile draw the rows' uniforms (one Philox block per row, not per lane);
//   2. LPR lanes per row, lane s owning the interleaved cells s, s + LPR, s + 2 LPR, ...: row max by DPP butterfly,
//      weights 2^((logit - max) * log2 e) (one FMA + one v_exp per cell; -inf gives an exact 0), an LPR-lane
//      inclusive scan, the uniform picks the point u * total on the cumulative axis (cells ordered lane-major --
//      any fixed order gives a draw from the same distribution), a ballot finds the lane that holds it and a
//      count of that lane's cells below the point finds the cell.
// After round 1's memory fix the kernel was bound by its own instruction count (the mask-only form took 7.7 of
// the f32 form's 12 us); per cell it now issues ~8 VALU instructions instead of ~18.
// Distribution = softmax over the legal cells (chi-square test); if rounding leaves the point beyond the last
// cell's cumulative weight the last legal cell of the last weighted lane is taken.
#include "mnk_host.h"

namespace {

constexpr int SAMPLE_THREADS = 256;
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

// masked logits of the global cells [e0, e1) -> lds[e - a0] as f32, a0 = e0 rounded down to a multiple of VE.
// A lane takes VE consecutive cells: one 16-byte load of logits and one 4/8/16-byte load of their mask bytes.
// Cells in front of e0 that the first load drags in belong to the previous workgroup's rows and are not used.
template <typename LT>
__device__ __forceinline__ void slab_to_lds(const LT* logits, const uint8_t* mask, int64_t e0, int64_t e1,
                                            int64_t total, float* lds, bool vec, int tid) {
  constexpr int VE = Slab<LT>::VE;
  const float NEG = -__builtin_huge_valf();
  const int64_t a0 = e0 & ~(int64_t)(VE - 1);
  if (!vec) {  // unaligned base pointers: one cell per lane and trip
    for (int64_t c = e0 + tid; c < e1; c += SAMPLE_THREADS) {
      float x = 0.0f;
      if constexpr (Slab<LT>::BYTES == 4) x = logits[c];
      if constexpr (Slab<LT>::BYTES == 2) x = __uint_as_float((uint32_t)logits[c] << 16);
      lds[c - a0] = mask[c] ? x : NEG;
    }
    return;
  }
  for (int64_t c = a0 + (int64_t)tid * VE; c < e1; c += (int64_t)SAMPLE_THREADS * VE) {
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

// LPR lanes per row (an aligned group inside one wave), K cells per lane, LT = float / uint16_t (bf16 bits) /
// void (no logits: all zero).  EXACT: LPR * (K - 1) < C, so only a lane's last cell can lie outside the row.
template <int LPR, int K, bool EXACT, typename LT>
__global__ void __launch_bounds__(SAMPLE_THREADS)
k_sample_logits(const LT* logits, const uint8_t* mask, int64_t N, int C, uint64_t seed, uint64_t step,
                const uint64_t* step_dev, int64_t env_id0, int deterministic, int64_t* actions, float* logp, int vec) {
  constexpr int ROWS = SAMPLE_THREADS / LPR;
  constexpr int VE = Slab<LT>::VE;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x;
  if (step_dev) step += *step_dev;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int rows_here = (int)((N - row0 < ROWS) ? (N - row0) : ROWS);
  const int64_t e0 = row0 * C, e1 = (row0 + rows_here) * C, total = N * C;
  float* lds_l = reinterpret_cast<float*>(lds_raw);          // [ROWS*C + 2*VE] masked logits
  float* lds_u = lds_l + ((size_t)ROWS * C + 2 * VE);        // [ROWS] the rows' uniforms
  slab_to_lds<LT>(logits, mask, e0, e1, total, lds_l, vec != 0, tid);
  if (tid < ROWS && !deterministic) {
    const uint32_t x = mnk_rand_u32(seed, (uint64_t)(env_id0 + row0 + tid), step, MNK_STREAM_SAMPLE);
    lds_u[tid] = ((float)(x >> 8) + 0.5f) * 5.9604644775390625e-08f;  // (0,1)
  }
  __syncthreads();

  const int r = tid / LPR, sub = tid % LPR;
  const int64_t row = row0 + r;
  const bool live = r < rows_here;
  const float* lrow = lds_l + (e0 & (VE - 1)) + (size_t)(live ? r : 0) * C;  // idle groups of the last workgroup redo row 0
  const float NEG = -__builtin_huge_valf();
  const int lane = tid & 63;
  const int gbase = lane & ~(LPR - 1);
  const unsigned long long gmask = (1ull << LPR) - 1ull;

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
    const float target = lds_u[live ? r : 0] * total_w;
    // first lane whose inclusive sum passes the target
    const unsigned long long pass = __ballot(incl > target && mine > 0.0f);
    const uint32_t pass_g = (uint32_t)((pass >> gbase) & gmask);
    int owner = __ffs(pass_g) - 1;
    // cells of this lane whose running sum stays at or below the target = index of the first one above it
    float run = incl - mine;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      run += w[j];
      cnt += (run <= target) ? 1 : 0;
    }
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
  if (sub == 0 && live) {
    actions[row] = chosen;
    if (logp) logp[row] = (none_legal ? 0.0f : lrow[chosen]) - rowmax - logf(total_w);
  }
}

template <int LPR, int K, bool EXACT, typename LT>
void launch_sample(const void* logits, const uint8_t* mask, int64_t N, int C, uint64_t seed, uint64_t step,
                   const uint64_t* step_dev, int64_t env_id0, int deterministic, int64_t* actions, float* logp,
                   hipStream_t s) {
  constexpr int ROWS = SAMPLE_THREADS / LPR;
  constexpr int VE = Slab<LT>::VE;
  // vector path: logits on a 16-byte boundary, mask on a 16-byte boundary (its 4/8/16-byte loads then are aligned too)
  const int vec = (aligned16(logits) && aligned16(mask)) ? 1 : 0;
  const size_t lds = ((size_t)ROWS * C + 2 * VE + ROWS) * sizeof(float);
  const dim3 grid((unsigned)((N + ROWS - 1) / ROWS));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sample_logits<LPR, K, EXACT, LT>), grid, dim3(SAMPLE_THREADS), lds, s,
                     (const LT*)logits, mask, N, C, seed, step, step_dev, env_id0, deterministic, actions, logp, vec);
}

template <typename LT>
void dispatch_sample(const void* logits, const uint8_t* mask, int64_t N, int C, uint64_t seed, uint64_t step,
                     const uint64_t* step_dev, int64_t env_id0, int deterministic, int64_t* actions, float* logp,
                     hipStream_t s) {
#define MNK_SAMPLE(LPRv, Kv, EXv) \
  launch_sample<LPRv, Kv, EXv, LT>(logits, mask, N, C, seed, step, step_dev, env_id0, deterministic, actions, logp, s)
  // lanes per row x cells per lane, by measurement (9x9: 4 lanes per row 7.1 us, 8 lanes per row 9.9 us; profiles/r02_api_kernels.md)
  if (C == 81) MNK_SAMPLE(4, 21, true);          // 9x9
  else if (C == 9) MNK_SAMPLE(4, 3, true);       // 3x3
  else if (C == 169) MNK_SAMPLE(8, 22, true);    // 13x13
  else if (C == 225) MNK_SAMPLE(16, 15, true);   // 15x15
  else if (C == 361) MNK_SAMPLE(16, 23, true);   // 19x19
  else if (C <= 32) MNK_SAMPLE(4, 8, false);
  else if (C <= 96) MNK_SAMPLE(8, 12, false);
  else if (C <= 256) MNK_SAMPLE(16, 16, false);
  else MNK_SAMPLE(32, 16, false);                // up to 512 cells (22x22)
#undef MNK_SAMPLE
}

}  // namespace

extern "C" int mnk_sample_logits(const void* logits, int logits_dtype, const uint8_t* mask, int64_t N, int C,
                                 uint64_t seed, uint64_t step, const uint64_t* step_dev, int64_t env_id0,
                                 int deterministic, int64_t* actions, float* logp, void* stream) {
  if (!mask || !actions || N < 0 || C < 1 || C > 512) return MNK_EINVAL;
  if (logits_dtype != MNK_LOGITS_F32 && logits_dtype != MNK_LOGITS_BF16) return MNK_EINVAL;
  if (N == 0) return MNK_OK;
  if (N > 0x7fffffffLL) return MNK_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (!logits) dispatch_sample<void>(nullptr, mask, N, C, seed, step, step_dev, env_id0, deterministic, actions, logp, s);
  else if (logits_dtype == MNK_LOGITS_BF16)
    dispatch_sample<uint16_t>(logits, mask, N, C, seed, step, step_dev, env_id0, deterministic, actions, logp, s);
  else dispatch_sample<float>(logits, mask, N, C, seed, step, step_dev, env_id0, deterministic, actions, logp, s);
  return mnk_launch_status("sample_logits");
}

// mnk_sample.hip -- masked categorical head fused with the draw (gfx950 / MI355X only).
//
// Replaces alg/architectures/cnn.py:69-79 (= resnet.py:84-95, transformer.py:80-91: logits where(mask) -inf,
// all-masked rows -> zeros, Categorical) followed by dist.sample() / argmax(dist.logits)
// (selfplay/policy.py:46-52) and dist.log_prob(action) (alg/ppo.py:96-97).
//
// HBM-bound: 5C + 12 bytes per row (f32 logits; 3C + 12 with bf16 logits) and nothing is reused.  A row of
// logits is C*4 bytes with C odd on every board people play (81, 169, 225, 361), so rows are not 16-byte
// aligned and lanes that own "their" cells directly would issue strided scalar loads (the round-1 kernel:
// 1.6 TB/s).  Here a 256-thread workgroup owns ROWS consecutive rows = one contiguous slab of logits and one of
// mask bytes, copies both to LDS with full-width 16-byte loads, and only then splits into LPR lanes per row:
//   lane s of a row owns the K consecutive cells [s*K, (s+1)*K)
//   row max / argmax by xor-shuffles inside the LPR-lane group (ties -> lowest cell, like torch.argmax)
//   weights e^(logit - max), an LPR-lane inclusive scan, ONE Philox uniform per row picks the point u * total on
//   the cumulative axis, a ballot finds the lane that holds it and that lane's walk over its K cells the cell.
// One exp per cell, no per-cell random numbers.  Distribution = softmax over the legal cells (chi-square test);
// rounding at the very end of the axis falls on the last legal cell.
#include "mnk_host.h"

namespace {

constexpr int SAMPLE_THREADS = 256;

// bf16 bit pattern -> f32
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// global elements [e0, e1) of `g` (an array of `total` elements) -> lds[e - a0], a0 = e0 rounded down to a
// 16-byte boundary.  Full 16-byte loads wherever the 16 bytes lie inside the array; the bytes in front of e0
// that such a load drags in belong to the previous workgroup's rows and are simply not used.
template <typename LT>
__device__ __forceinline__ void slab_to_lds_f32(const LT* g, int64_t e0, int64_t e1, int64_t total, float* lds,
                                                bool vec, int tid) {
  constexpr int VE = 16 / (int)sizeof(LT);
  const int64_t a0 = e0 & ~(int64_t)(VE - 1);
  if (vec) {
    for (int64_t c = a0 + (int64_t)tid * VE; c < e1; c += (int64_t)SAMPLE_THREADS * VE) {
      float* dst = lds + (c - a0);
      if (c + VE <= total) {
        const uint4 v = *reinterpret_cast<const uint4*>(g + c);
        if constexpr (sizeof(LT) == 4) {
          *reinterpret_cast<uint4*>(dst) = v;
        } else {
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            dst[2 * j] = __uint_as_float(w[j] << 16);
            dst[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
          }
        }
      } else {
        for (int j = 0; j < VE && c + j < total; ++j) {
          if constexpr (sizeof(LT) == 4) dst[j] = g[c + j];
          else dst[j] = bf16_to_f32(g[c + j]);
        }
      }
    }
  } else {
    for (int64_t c = e0 + tid; c < e1; c += SAMPLE_THREADS) {
      if constexpr (sizeof(LT) == 4) lds[c - a0] = g[c];
      else lds[c - a0] = bf16_to_f32(g[c]);
    }
  }
}

__device__ __forceinline__ void slab_to_lds_u8(const uint8_t* g, int64_t e0, int64_t e1, int64_t total, uint8_t* lds,
                                               bool vec, int tid) {
  const int64_t a0 = e0 & ~(int64_t)15;
  if (vec) {
    for (int64_t c = a0 + (int64_t)tid * 16; c < e1; c += (int64_t)SAMPLE_THREADS * 16) {
      if (c + 16 <= total) {
        *reinterpret_cast<uint4*>(lds + (c - a0)) = *reinterpret_cast<const uint4*>(g + c);
      } else {
        for (int j = 0; j < 16 && c + j < total; ++j) lds[c - a0 + j] = g[c + j];
      }
    }
  } else {
    for (int64_t c = e0 + tid; c < e1; c += SAMPLE_THREADS) lds[c - a0] = g[c];
  }
}

// LPR lanes per row (8 / 16 / 32, an aligned group inside one wave), K cells per lane, LT = float or bf16 bits.
// logits == nullptr: all logits are zero (RandomPolicy: uniform over the legal cells) and only the mask is read.
template <int LPR, int K, typename LT>
__global__ void __launch_bounds__(SAMPLE_THREADS)
k_sample_logits(const LT* logits, const uint8_t* mask, int64_t N, int C, uint64_t seed, uint64_t step,
                const uint64_t* step_dev, int64_t env_id0, int deterministic, int64_t* actions, float* logp,
                int vec_ok) {
  constexpr int ROWS = SAMPLE_THREADS / LPR;
  constexpr int VE = 16 / (int)sizeof(LT);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x;
  if (step_dev) step += *step_dev;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int64_t rows_here = (N - row0 < ROWS) ? (N - row0) : ROWS;
  const int64_t e0 = row0 * C, e1 = (row0 + rows_here) * C, total = N * C;
  // LDS image: f32 logits [ROWS*C + 2*VE] | mask bytes [ROWS*C + 32]
  float* lds_l = reinterpret_cast<float*>(lds_raw);
  uint8_t* lds_m = lds_raw + (((size_t)ROWS * C + 2 * VE) * 4 + 15) / 16 * 16;
  if (logits) slab_to_lds_f32<LT>(logits, e0, e1, total, lds_l, (vec_ok & 1) != 0, tid);
  slab_to_lds_u8(mask, e0, e1, total, lds_m, (vec_ok & 2) != 0, tid);
  __syncthreads();

  const int r = tid / LPR, sub = tid % LPR;
  const int64_t row = row0 + r;
  const bool live = r < rows_here;
  const int rr = live ? r : 0;  // idle groups of the last workgroup recompute row 0 of the slab; nothing is written
  const float* lrow = lds_l + (e0 & (VE - 1)) + (size_t)rr * C;
  const uint8_t* mrow = lds_m + (e0 & 15) + (size_t)rr * C;
  const float NEG = -__builtin_huge_valf();
  const int c_lo = sub * K;
  const int lane = tid & 63;
  const int gbase = lane & ~(LPR - 1);
  const unsigned long long gmask = (LPR == 64) ? ~0ull : ((1ull << LPR) - 1ull);

  float l[K];
  int any = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const int c = c_lo + j;
    const int cc = c < C ? c : C - 1;
    const bool legal = (c < C) && mrow[cc] != 0;
    any |= legal ? 1 : 0;
    l[j] = legal ? (logits ? lrow[cc] : 0.0f) : NEG;
  }
  // does the row have a legal cell at all?  (cnn.py:76-77: all-masked -> zeros -> uniform over all cells)
  const unsigned long long votes = __ballot(any != 0);
  const bool none_legal = ((votes >> gbase) & gmask) == 0ull;
#pragma unroll
  for (int j = 0; j < K; ++j) l[j] = none_legal ? ((c_lo + j < C) ? 0.0f : NEG) : l[j];
  float rowmax = NEG;
  int rowarg = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < K; ++j)
    if (l[j] > rowmax) { rowmax = l[j]; rowarg = c_lo + j; }  // first maximum of the lane, cells ascend
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) {  // ties -> lowest cell, like torch.argmax (policy.py:48-49)
    const float ov = __shfl_xor(rowmax, off, 64);
    const int oi = __shfl_xor(rowarg, off, 64);
    if (ov > rowmax || (ov == rowmax && oi < rowarg)) { rowmax = ov; rowarg = oi; }
  }
  float w[K];
  float mine = 0.0f;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    w[j] = (l[j] == NEG) ? 0.0f : __expf(l[j] - rowmax);
    mine += w[j];
  }
  float incl = mine;  // inclusive scan over the LPR lanes of the row
#pragma unroll
  for (int off = 1; off < LPR; off <<= 1) {
    const float up = __shfl_up(incl, off, LPR);
    if (sub >= off) incl += up;
  }
  const float total_w = __shfl(incl, LPR - 1, LPR);
  int chosen = rowarg;
  if (!deterministic) {
    const uint32_t x = mnk_rand_u32(seed, (uint64_t)(env_id0 + (live ? row : row0)), step, MNK_STREAM_SAMPLE);
    const float u = ((float)(x >> 8) + 0.5f) * 5.9604644775390625e-08f;  // (0,1)
    const float target = u * total_w;
    // first lane whose inclusive sum passes the target (the last lane with weight, if rounding overshoots)
    const unsigned long long pass = __ballot(incl > target && mine > 0.0f);
    const unsigned long long heavy = __ballot(mine > 0.0f);
    const uint32_t pass_g = (uint32_t)((pass >> gbase) & gmask), heavy_g = (uint32_t)((heavy >> gbase) & gmask);
    const int owner = pass_g ? __ffs(pass_g) - 1 : 31 - __clz(heavy_g);
    // every lane walks its own cells (no divergence); the owner's answer is broadcast
    float run = incl - mine;
    int pick = 0x7fffffff, last = c_lo;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      run += w[j];
      const bool has = w[j] > 0.0f;
      last = has ? c_lo + j : last;
      pick = (has && pick == 0x7fffffff && run > target) ? c_lo + j : pick;
    }
    pick = pick == 0x7fffffff ? last : pick;
    chosen = __shfl(pick, owner, LPR);
  }
  if (sub == 0 && live) {
    actions[row] = chosen;
    if (logp) logp[row] = ((none_legal || !logits) ? 0.0f : lrow[chosen]) - rowmax - logf(total_w);
  }
}

template <int LPR, int K, typename LT>
void launch_sample(const void* logits, const uint8_t* mask, int64_t N, int C, uint64_t seed, uint64_t step,
                   const uint64_t* step_dev, int64_t env_id0, int deterministic, int64_t* actions, float* logp,
                   int vec_ok, hipStream_t s) {
  constexpr int ROWS = SAMPLE_THREADS / LPR;
  constexpr int VE = 16 / (int)sizeof(LT);
  const size_t lds = (((size_t)ROWS * C + 2 * VE) * 4 + 15) / 16 * 16 + (size_t)ROWS * C + 32;
  const dim3 grid((unsigned)((N + ROWS - 1) / ROWS));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sample_logits<LPR, K, LT>), grid, dim3(SAMPLE_THREADS), lds, s,
                     (const LT*)logits, mask, N, C, seed, step, step_dev, env_id0, deterministic, actions, logp,
                     vec_ok);
}

template <typename LT>
void dispatch_sample(const void* logits, const uint8_t* mask, int64_t N, int C, uint64_t seed, uint64_t step,
                     const uint64_t* step_dev, int64_t env_id0, int deterministic, int64_t* actions, float* logp,
                     int vec_ok, hipStream_t s) {
#define MNK_SAMPLE(LPRv, Kv)                                                                                     \
  launch_sample<LPRv, Kv, LT>(logits, mask, N, C, seed, step, step_dev, env_id0, deterministic, actions, logp, \
                              vec_ok, s)
  if (C <= 16) MNK_SAMPLE(8, 2);          // 3x3, 4x4
  else if (C <= 32) MNK_SAMPLE(8, 4);     // 4x6, 5x5
  else if (C <= 64) MNK_SAMPLE(8, 8);     // 7x9, 8x8
  else if (C <= 88) MNK_SAMPLE(8, 11);    // 9x9
  else if (C <= 128) MNK_SAMPLE(16, 8);   // 10x10, 11x11
  else if (C <= 192) MNK_SAMPLE(16, 12);  // 12x12, 13x13
  else if (C <= 256) MNK_SAMPLE(16, 16);  // 15x15
  else if (C <= 384) MNK_SAMPLE(32, 12);  // 19x19
  else MNK_SAMPLE(32, 16);                // up to 512 cells (22x22)
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
  const int vec_ok = (aligned16(logits) ? 1 : 0) | (aligned16(mask) ? 2 : 0);
  if (logits_dtype == MNK_LOGITS_BF16)
    dispatch_sample<uint16_t>(logits, mask, N, C, seed, step, step_dev, env_id0, deterministic, actions, logp, vec_ok,
                              (hipStream_t)stream);
  else
    dispatch_sample<float>(logits, mask, N, C, seed, step, step_dev, env_id0, deterministic, actions, logp, vec_ok,
                           (hipStream_t)stream);
  return mnk_launch_status("sample_logits");
}

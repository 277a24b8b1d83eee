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
#include "mnk_draw.h"

namespace {

using namespace mnk_draw;
constexpr int SAMPLE_THREADS = 256;

// LPR lanes per row (an aligned group inside one wave), K cells per lane, LT = float / uint16_t (bf16 bits) /
// void (no logits: all zero).  EXACT: LPR * (K - 1) < C, so only a lane's last cell can lie outside the row.
template <int LPR, int K, bool EXACT, typename LT>
__global__ void __launch_bounds__(SAMPLE_THREADS)
k_sample_logits(const LT* logits, const uint8_t* mask, int64_t N, int C, uint64_t seed, const uint64_t* seed_dev,
                uint64_t step, const uint64_t* step_dev, int64_t env_id0, int deterministic, int64_t* actions, float* logp,
                int vec) {
  constexpr int ROWS = SAMPLE_THREADS / LPR;
  constexpr int VE = Slab<LT>::VE;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x;
  if (step_dev) step += *step_dev;
  if (seed_dev) seed = *seed_dev;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int rows_here = (int)((N - row0 < ROWS) ? (N - row0) : ROWS);
  const int64_t e0 = row0 * C, e1 = (row0 + rows_here) * C, total = N * C;
  float* lds_l = reinterpret_cast<float*>(lds_raw);          // [ROWS*C + 2*VE] masked logits
  float* lds_u = lds_l + slab_floats<LT>(ROWS, C);           // [ROWS] the rows' uniforms
  slab_to_lds<LT>(logits, mask, e0, e1, total, lds_l, vec != 0, tid, SAMPLE_THREADS);
  if (tid < ROWS && !deterministic) lds_u[tid] = row_uniform(seed, (uint64_t)(env_id0 + row0 + tid), step);
  __syncthreads();

  const int r = tid / LPR, sub = tid % LPR;
  const int64_t row = row0 + r;
  const bool live = r < rows_here;
  const float* lrow = lds_l + (e0 & (VE - 1)) + (size_t)(live ? r : 0) * C;  // idle groups of the last workgroup redo row 0
  const Drawn d = draw_row<LPR, K, EXACT>(lrow, C, lds_u[live ? r : 0], deterministic, tid);
  if (sub == 0 && live) {
    actions[row] = d.chosen;
    if (logp) logp[row] = d.logp(lrow);
  }
}

template <int LPR, int K, bool EXACT, typename LT>
void launch_sample(const void* logits, const uint8_t* mask, int64_t N, int C, uint64_t seed, const uint64_t* seed_dev,
                   uint64_t step, const uint64_t* step_dev, int64_t env_id0, int deterministic, int64_t* actions, float* logp,
                   hipStream_t s) {
  constexpr int ROWS = SAMPLE_THREADS / LPR;
  constexpr int VE = Slab<LT>::VE;
  // vector path: logits on a 16-byte boundary, mask on a 16-byte boundary (its 4/8/16-byte loads then are aligned too)
  const int vec = (aligned16(logits) && aligned16(mask)) ? 1 : 0;
  const size_t lds = (slab_floats<LT>(ROWS, C) + ROWS) * sizeof(float);
  const dim3 grid((unsigned)((N + ROWS - 1) / ROWS));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sample_logits<LPR, K, EXACT, LT>), grid, dim3(SAMPLE_THREADS), lds, s,
                     (const LT*)logits, mask, N, C, seed, seed_dev, step, step_dev, env_id0, deterministic, actions, logp, vec);
}

template <typename LT>
void dispatch_sample(const void* logits, const uint8_t* mask, int64_t N, int C, uint64_t seed, const uint64_t* seed_dev,
                     uint64_t step, const uint64_t* step_dev, int64_t env_id0, int deterministic, int64_t* actions,
                     float* logp, hipStream_t s) {
#define MNK_SAMPLE(LPRv, Kv, EXv) \
  launch_sample<LPRv, Kv, EXv, LT>(logits, mask, N, C, seed, seed_dev, step, step_dev, env_id0, deterministic, actions, logp, s)
#define MNK_SAMPLE_SHAPE(Cv) MNK_SAMPLE(Shape<Cv>::LPR, Shape<Cv>::K, Shape<Cv>::EXACT)
  // lanes per row x cells per lane, by measurement (9x9: 4 lanes per row 7.1 us, 8 lanes per row 9.9 us; profiles/r02_api_kernels.md)
  // (the shapes of these five live in mnk_draw.h: the step kernels with a folded-in draw must use the same ones)
  if (C == 81) MNK_SAMPLE_SHAPE(81);             // 9x9
  else if (C == 9) MNK_SAMPLE_SHAPE(9);          // 3x3
  else if (C == 169) MNK_SAMPLE_SHAPE(169);      // 13x13
  else if (C == 225) MNK_SAMPLE_SHAPE(225);      // 15x15
  else if (C == 361) MNK_SAMPLE_SHAPE(361);      // 19x19
  // (the generic shapes: mnk_draw::Shape's primary template names the same ones, for the run-time specialised step
  // kernels that fold this draw in)
  else if (C <= 32) MNK_SAMPLE_SHAPE(32);
  else if (C <= 96) MNK_SAMPLE_SHAPE(96);
  else if (C <= 256) MNK_SAMPLE_SHAPE(256);
  else if (C <= 512) MNK_SAMPLE_SHAPE(512);      // up to 512 cells (22x22)
  else MNK_SAMPLE_SHAPE(1024);                   // up to 1 024 cells (25x25, 31x31)
#undef MNK_SAMPLE_SHAPE
#undef MNK_SAMPLE
}

}  // namespace

// the draw as a launch of its own, for the translation units that fold it into a step kernel where they can and fall
// back to two launches where they cannot (mnk_selfplay_draw.hip)
int mnk_launch_sample(const MnkSample& sa, int64_t N, int C, hipStream_t s) {
  if (!sa.logits) dispatch_sample<void>(nullptr, sa.mask, N, C, sa.seed, sa.seed_dev, sa.step, sa.step_dev, sa.env_id0, sa.deterministic, sa.actions, sa.logp, s);
  else if (sa.logits_dtype == MNK_LOGITS_BF16)
    dispatch_sample<uint16_t>(sa.logits, sa.mask, N, C, sa.seed, sa.seed_dev, sa.step, sa.step_dev, sa.env_id0, sa.deterministic, sa.actions, sa.logp, s);
  else dispatch_sample<float>(sa.logits, sa.mask, N, C, sa.seed, sa.seed_dev, sa.step, sa.step_dev, sa.env_id0, sa.deterministic, sa.actions, sa.logp, s);
  return mnk_launch_status("sample_logits");
}

extern "C" int mnk_sample_logits(const void* logits, int logits_dtype, const uint8_t* mask, int64_t N, int C,
                                 uint64_t seed, const uint64_t* seed_dev, uint64_t step, const uint64_t* step_dev,
                                 int64_t env_id0, int deterministic, int64_t* actions, float* logp, void* stream) {
  if (!mask || !actions || N < 0 || C < 1 || C > 1024) return MNK_EINVAL;
  if (logits_dtype != MNK_LOGITS_F32 && logits_dtype != MNK_LOGITS_BF16) return MNK_EINVAL;
  if (N == 0) return MNK_OK;
  if (N > 0x7fffffffLL) return MNK_EINVAL;
  const MnkSample sa = {logits, logits_dtype, mask, seed, seed_dev, step, step_dev, env_id0, deterministic, actions, logp};
  return mnk_launch_sample(sa, N, C, (hipStream_t)stream);
}

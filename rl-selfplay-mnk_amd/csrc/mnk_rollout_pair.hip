// mnk_rollout_pair.hip -- the two-lanes-per-env form of the fused random rollout (gfx950 / MI355X only).
// Its own translation unit so the many variants compile in parallel with the one-lane kernels.
#include "mnk_host.h"
#include "mnk_rollout_lane.h"

// ------------------------------------------------------------------ two lanes per env
// For SMALL batches.  With at most 32 envs per SIMD of the chip (<= 32 768 envs) the one-lane kernel leaves
// SIMDs empty.  This variant gives every env to a PAIR of adjacent lanes -- 32 envs per wave, twice the
// waves -- and splits what splits cleanly: lane 0 scans rows + columns, lane 1 diagonals + anti-diagonals (the
// shift amount is a per-lane VGPR; one DPP swap ORs the verdicts); lane r writes half r of every record row;
// each lane computes every other Philox block and hands its four words to its partner by DPP.  Move selection
// and the state update are done redundantly by both lanes (cheaper than exchanging them).  Results are
// bit-identical to the one-lane kernel (all compile-time board geometries).  The pair form executes 1.6x the
// instructions per env, so it wins exactly while both lanes of every env fit one wave per SIMD (2N <= 65 536
// lanes) and the launcher uses it only up to 32 768 envs.

// the T plies of one lane pair (FAST: see mnk_rollout_lane.h -- every game of the wave consistent)
template <bool FAST, typename Lane>
__device__ __forceinline__ void pair_plies(Lane& L, int T, uint64_t seed, uint64_t step0, uint64_t env, uint32_t role) {
  int t = 0;
  uint64_t step = step0;
  // unshared Philox until the step counter sits on a multiple of 8 (two blocks)
  for (; t < T && (step & 7); ++t, ++step)
    L.template ply<FAST>(mnk_rand_u32(seed, env, step, MNK_STREAM_MOVE), (int)(step & 3));
  for (; t + 8 <= T; t += 8, step += 8) {
    // lane `role` computes block (step/4 + role); the partner's four words arrive by DPP
    const Philox4 mine = mnk_rng_block(seed, env, (step >> 2) + role, MNK_STREAM_MOVE);
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t other = pair_swap(mine.v[j]);
      lo[j] = role ? other : mine.v[j];  // block step/4
      hi[j] = role ? mine.v[j] : other;  // block step/4 + 1
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) L.template ply<FAST>(lo[j], j);
#pragma unroll
    for (int j = 0; j < 4; ++j) L.template ply<FAST>(hi[j], j);
  }
  for (; t < T; ++t, ++step) L.template ply<FAST>(mnk_rand_u32(seed, env, step, MNK_STREAM_MOVE), (int)(step & 3));
}

template <int NW, int CN, int CK, bool RECORD, int ACT>
__global__ void __launch_bounds__(64)
k_rollout_random_pair(MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed, uint64_t step0,
                      int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, unsigned long long* stats,
                      void* act_log) {
  __shared__ unsigned int lds_stats[MNK_STATS_COUNTERS];
  if (threadIdx.x < MNK_STATS_COUNTERS) lds_stats[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t role = threadIdx.x & 1u;
  const int64_t i = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 1);  // env of this lane pair
  if (i < N) {
    RolloutLane<NW, CN, CK, RECORD, ACT, true> L(g, N, i, rec_planes, rec_meta, act_log, role);
    L.load(planes, meta, i);
    const uint64_t env = (uint64_t)(env_id0 + i);
    // both lanes of a pair hold the whole board: the consistency test of the one-lane form applies as it is.  Small
    // boards only (what the launcher uses this form for: 3x3 and 9x9); the larger ones keep the one general loop.
    constexpr bool TRY_FAST = NW <= 3;
    if (TRY_FAST && __builtin_amdgcn_ballot_w64(!L.consistent()) == 0) {
      pair_plies<true>(L, T, seed, step0, env, role);
      L.finish_fast();
    } else {
      pair_plies<false>(L, T, seed, step0, env, role);
    }
    if (ACT && (T & 3)) L.log_flush();
    L.store(planes, meta, i);  // lane `role` stores plane `role`; the meta word is written by both
    if (stats && role == 0) {
      const uint32_t len_sum = L.moves_in + (uint32_t)T - L.moves;
      if (L.acc_done) atomicAdd(&lds_stats[0], L.acc_done);
      if (L.acc_win - L.acc_white) atomicAdd(&lds_stats[1], L.acc_win - L.acc_white);
      if (L.acc_white) atomicAdd(&lds_stats[2], L.acc_white);
      if (L.acc_done - L.acc_win) atomicAdd(&lds_stats[3], L.acc_done - L.acc_win);
      if (len_sum) atomicAdd(&lds_stats[4], len_sum);
    }
  }
  __syncthreads();
  if (stats && threadIdx.x < MNK_STATS_COUNTERS && lds_stats[threadIdx.x])
    atomicAdd(&stats[(size_t)(blockIdx.x % MNK_STATS_REPLICAS) * MNK_STATS_STRIDE + threadIdx.x],
              (unsigned long long)lds_stats[threadIdx.x]);
}

void mnk_launch_rollout_pair(const MnkGeom& g, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed,
                             uint64_t step0, int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, int64_t* stats,
                             void* act_log, int act_bytes, void* stream) {
  const bool rec = rec_planes && rec_meta;
    const dim3 pgrid((unsigned)((N + 31) / 32));
#define MNK_PAIR(NWv, CNv, CKv, REC, ACTB)                                                                     \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rollout_random_pair<NWv, CNv, CKv, REC, ACTB>), pgrid, dim3(64), 0,     \
                     (hipStream_t)stream, g, planes, meta, N, T, seed, step0, env_id0, rec_planes, rec_meta,   \
                     (unsigned long long*)stats, act_log)
#define MNK_PAIR_GEOM(REC, ACTB)                               \
  do {                                                         \
    if (g.n == 9) MNK_PAIR(3, 9, 5, REC, ACTB);                \
    else if (g.n == 3) MNK_PAIR(1, 3, 3, REC, ACTB);           \
    else if (g.n == 13) MNK_PAIR(6, 13, 5, REC, ACTB);         \
    else if (g.n == 15) MNK_PAIR(8, 15, 5, REC, ACTB);         \
    else MNK_PAIR(12, 19, 5, REC, ACTB);                       \
  } while (0)
    if (rec && act_bytes == 1) MNK_PAIR_GEOM(true, 1);
    else if (rec && act_bytes == 2) MNK_PAIR_GEOM(true, 2);
    else if (rec) MNK_PAIR_GEOM(true, 0);
    else if (act_bytes == 1) MNK_PAIR_GEOM(false, 1);
    else if (act_bytes == 2) MNK_PAIR_GEOM(false, 2);
    else MNK_PAIR_GEOM(false, 0);
#undef MNK_PAIR_GEOM
#undef MNK_PAIR
}

// mnk_rollout_pair.hip -- the two-lanes-per-env form of the fused random rollout (gfx950 / MI355X only).
// Its own translation unit so the many variants compile in parallel with the one-lane kernels.
#include "mnk_host.h"

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

template <int NW, int CN, int CK, bool RECORD, int ACT>
struct PairLane {
  const MnkGeom& g;
  MnkEnv<NW> e;
  int64_t N;
  uint32_t role;      // 0 / 1 within the pair
  uint32_t* rp = nullptr;  // half `role` of rec_planes[t][0][env]: lane 0 writes the mover's words, lane 1 the other side's
  uint32_t* rm = nullptr;  // rec_meta[t][env]
  uint8_t* ra = nullptr;   // act_log[t / 4][env]
  uint64_t quad = 0;       // four actions, 8 or 16 bits each (ACT = 1 / 2)
  uint32_t acc_done_draw = 0, acc_black_white = 0, len_sum = 0;

  __device__ __forceinline__ PairLane(const MnkGeom& g_, int64_t N_, int64_t env, uint32_t role_,
                                      uint64_t* rec_planes, uint32_t* rec_meta, void* act_log)
      : g(g_), N(N_), role(role_) {
    if (RECORD) { rp = (uint32_t*)(rec_planes + env) + role; rm = rec_meta + env; }
    if (ACT) ra = (uint8_t*)act_log + env * 4 * ACT;
  }

  __device__ __forceinline__ void ply(uint32_t x, int field) {
    const int a = env_pick_legal<NW, CN>(g, e, x);
    if (ACT) {
      quad |= (uint64_t)(uint32_t)a << (8 * ACT * field);
      if (field == 3) flush_log();
    }
    const uint32_t side = e.meta & 1u;
    if (RECORD) {  // lane 0 writes the mover's word of every row, lane 1 the other side's: one 256-byte store per wave
      const bool white_half = (role ^ side) != 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) rp[(int64_t)w * 2 * N] = white_half ? e.p[1][w] : e.p[0][w];
      rp += (int64_t)NW * 2 * N;
    }
    const uint32_t bit = (uint32_t)a + (uint32_t)a / (uint32_t)CN;
    const int wsel = (int)(bit >> 5);
    const uint32_t one = 1u << (bit & 31u);
    uint32_t mover[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const uint32_t add = (w == wsel) ? one : 0u;
      e.p[0][w] |= side ? 0u : add;
      e.p[1][w] |= side ? add : 0u;
      mover[w] = side ? e.p[1][w] : e.p[0][w];
    }
    // role 0 scans columns and rows, role 1 diagonals and anti-diagonals; paired so that the word parts of
    // the shift amounts agree wherever the board allows (n+1 with n+2, 1 with n)
    uint32_t hit = bs_run_bits_pair<NW, CK, CN + 1, CN + 2>(mover, role) | bs_run_bits_pair<NW, CK, 1, CN>(mover, role);
    hit |= pair_swap(hit);  // the partner's two directions
    const uint32_t win = hit ? 1u : 0u;
    const uint32_t moves = (e.meta >> 1) + 1u;
    const uint32_t done = (win | (moves >= (uint32_t)g.C ? 1u : 0u));
    e.meta = (moves << 1) | (side ^ 1u);
    if (RECORD) {  // both lanes write the same word to the same address
      *rm = (uint32_t)a | (win << MNK_REC_REWARD_SHIFT) | (done << MNK_REC_DONE_BIT) | (side << MNK_REC_SIDE_BIT);
      rm += N;
    }
    acc_done_draw += done + ((done & ~win) << 16);
    acc_black_white += (win & ~side) + ((win & side) << 16);
    len_sum += done ? moves : 0u;
    if (done) env_clear<NW>(e);
  }

  __device__ __forceinline__ void flush_log() {  // both lanes write the same word
    if (ACT == 1) *(uint32_t*)ra = (uint32_t)quad;
    if (ACT == 2) *(uint64_t*)ra = quad;
    ra += N * 4 * ACT;
    quad = 0;
  }
};

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
    PairLane<NW, CN, CK, RECORD, ACT> L(g, N, i, role, rec_planes, rec_meta, act_log);
    env_load<NW, true>(L.e, planes, meta, N, g.W, i);
    const uint64_t env = (uint64_t)(env_id0 + i);
    int t = 0;
    uint64_t step = step0;
    // unshared Philox until the step counter sits on a multiple of 8 (two blocks)
    for (; t < T && (step & 7); ++t, ++step) L.ply(mnk_rand_u32(seed, env, step, MNK_STREAM_MOVE), (int)(step & 3));
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
      for (int j = 0; j < 4; ++j) L.ply(lo[j], j);
#pragma unroll
      for (int j = 0; j < 4; ++j) L.ply(hi[j], j);
    }
    for (; t < T; ++t, ++step) L.ply(mnk_rand_u32(seed, env, step, MNK_STREAM_MOVE), (int)(step & 3));
    if (ACT && (T & 3)) L.flush_log();
    // lane `role` stores plane `role`; the meta word is written by both
    {
      uint32_t mine_plane[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) mine_plane[w] = role ? L.e.p[1][w] : L.e.p[0][w];
      plane_store<NW, true>(mine_plane, planes + (int64_t)role * g.W * N, N, g.W, i);
      meta[i] = L.e.meta;
    }
    if (stats && role == 0) {
      if (L.acc_done_draw & 0xFFFFu) atomicAdd(&lds_stats[0], L.acc_done_draw & 0xFFFFu);
      if (L.acc_black_white & 0xFFFFu) atomicAdd(&lds_stats[1], L.acc_black_white & 0xFFFFu);
      if (L.acc_black_white >> 16) atomicAdd(&lds_stats[2], L.acc_black_white >> 16);
      if (L.acc_done_draw >> 16) atomicAdd(&lds_stats[3], L.acc_done_draw >> 16);
      if (L.len_sum) atomicAdd(&lds_stats[4], L.len_sum);
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

// mnk_rollout_ws.hip -- the waves-per-env-group form of the fused random rollout (gfx950 / MI355X only).
// Its own translation unit so its variants compile in parallel with the other rollout kernels.
#include "mnk_host.h"
#include "mnk_rollout_lane.h"

// ------------------------------------------------------------------ WS waves per group of 64 envs
// The one-lane kernel gives a SIMD one wave at 65 536 envs and leaves half the SIMDs empty at 32 768 -- and a wave
// that is alone on its SIMD issues one instruction per ~4.5 cycles whatever it is.  On 19x19
// the four-direction scan is ~45 % of a ply's instructions.  Here a workgroup of WS waves (2 or 4) carries the
// same 64 envs in every wave: each wave picks the move and updates the state redundantly (uniform random play is a
// pure function of (seed, env, step), so no exchange is needed for that), scans only its 4 / WS directions with
// compile-time shift amounts (the role is wave-uniform: a scalar branch, unlike the per-lane roles of the pair
// form), and writes only its share of the record rows.  The verdicts meet in LDS: one ds_write_b32, one
// s_barrier, one ds_read per ply -- north_star's "win scan staged in LDS with a wave-level any-reduce", in the
// form that needs the fewest LDS operations.  Results are bit-identical to the one-lane kernel.
// MEASURED AND NOT USED BY DEFAULT (tools/exp_forms.py, us per 256 plies, lane / pair / ws2 / ws4): 9x9x5 x 65 536
// envs 95 / 128 / 145 / 216; 19x19x5 x 32 768 envs 248 / 216 / 356 / 285; 19x19x5 x 16 384 envs 237 / 213 / 254 / 209.
// The per-ply s_barrier + LDS round trip costs ~700 cycles per ply at two waves per SIMD -- more than the scan it
// splits.  Kept as the measured alternative (MNK_ROLLOUT_FORM=ws2|ws4, no action log) and covered by the parity tests.
template <int NW, int CN, int CK, bool RECORD, int ACT, int WS>
__global__ void __launch_bounds__(64 * WS)
k_rollout_random_ws(MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed, uint64_t step0,
                    int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, unsigned long long* stats,
                    void* act_log) {
  __shared__ unsigned int lds_stats[MNK_STATS_COUNTERS];
  __shared__ uint32_t lds_verdict[2 * 64 * WS];
  if (threadIdx.x < MNK_STATS_COUNTERS) lds_stats[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t wrole = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  if (i < N) {
    RolloutLane<NW, CN, CK, RECORD, ACT, false, WS> L(g, N, i, rec_planes, rec_meta, act_log, 0, wrole, lds_verdict);
    L.load(planes, meta, i);
    const uint64_t env = (uint64_t)(env_id0 + i);
    int t = 0;
    uint64_t step = step0;
    if (step & 3) {
      const Philox4 blk = mnk_rng_block(seed, env, step >> 2, MNK_STREAM_MOVE);
      for (; t < T && (step & 3); ++t, ++step) L.ply(philox_word(blk, (uint32_t)(step & 3)), (int)(step & 3));
    }
    for (; t + 4 <= T; t += 4, step += 4) {
      const Philox4 blk = mnk_rng_block(seed, env, step >> 2, MNK_STREAM_MOVE);
      L.ply(blk.v[0], 0);
      L.ply(blk.v[1], 1);
      L.ply(blk.v[2], 2);
      L.ply(blk.v[3], 3);
    }
    if (t < T) {
      const Philox4 blk = mnk_rng_block(seed, env, step >> 2, MNK_STREAM_MOVE);
      for (uint32_t j = 0; t < T; ++t, ++j) L.ply(philox_word(blk, j), (int)j);
    }
    if (ACT && (T & 3)) L.log_flush();
    if (wrole == 0) {  // every wave of the group ends in the same state
      L.store(planes, meta, i);
      if (stats) {
        const uint32_t len_sum = L.moves_in + (uint32_t)T - L.moves;
        if (L.acc_done) atomicAdd(&lds_stats[0], L.acc_done);
        if (L.acc_win - L.acc_white) atomicAdd(&lds_stats[1], L.acc_win - L.acc_white);
        if (L.acc_white) atomicAdd(&lds_stats[2], L.acc_white);
        if (L.acc_done - L.acc_win) atomicAdd(&lds_stats[3], L.acc_done - L.acc_win);
        if (len_sum) atomicAdd(&lds_stats[4], len_sum);
      }
    }
  }
  __syncthreads();
  if (stats && threadIdx.x < MNK_STATS_COUNTERS && lds_stats[threadIdx.x])
    atomicAdd(&stats[(size_t)(blockIdx.x % MNK_STATS_REPLICAS) * MNK_STATS_STRIDE + threadIdx.x],
              (unsigned long long)lds_stats[threadIdx.x]);
}

// ws = 2 or 4; geometry must be one of the boards below (mnk_rollout_ws_supported)
bool mnk_rollout_ws_supported(const MnkGeom& g, int act_bytes) {
  if (act_bytes) return false;
  return (g.n == 9 && g.k == 5 && g.NW == 3) || (g.n == 19 && g.k == 5 && g.NW == 12);
}

void mnk_launch_rollout_ws(const MnkGeom& g, int ws, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed,
                           uint64_t step0, int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, int64_t* stats,
                           void* act_log, int act_bytes, void* stream) {
  const bool rec = rec_planes && rec_meta;
  const dim3 grid((unsigned)((N + 63) / 64));
#define MNK_WS(NWv, CNv, CKv, REC, ACTB, WSv)                                                                      \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rollout_random_ws<NWv, CNv, CKv, REC, ACTB, WSv>), grid, dim3(64 * WSv), 0, \
                     (hipStream_t)stream, g, planes, meta, N, T, seed, step0, env_id0, rec_planes, rec_meta,       \
                     (unsigned long long*)stats, act_log)
#define MNK_WS_BOARD(NWv, CNv, CKv)                         \
  do {                                                      \
    if (ws == 4) {                                          \
      if (rec) MNK_WS(NWv, CNv, CKv, true, 0, 4);           \
      else MNK_WS(NWv, CNv, CKv, false, 0, 4);              \
    } else {                                                \
      if (rec) MNK_WS(NWv, CNv, CKv, true, 0, 2);           \
      else MNK_WS(NWv, CNv, CKv, false, 0, 2);              \
    }                                                       \
  } while (0)
  (void)act_log; (void)act_bytes;
  if (g.n == 9) MNK_WS_BOARD(3, 9, 5);
  else MNK_WS_BOARD(12, 19, 5);
#undef MNK_WS_BOARD
#undef MNK_WS
}

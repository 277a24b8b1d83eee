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

// (pair_plies and the kernel's body live in mnk_rollout_lane.h: boards without an ahead-of-time variant get this form
// compiled at run time too, mnk_jit.hip)
template <int NW, int CN, int CK, bool RECORD, int ACT>
__global__ void __launch_bounds__(64)
k_rollout_random_pair(MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed, uint64_t step0,
                      int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, unsigned long long* stats,
                      void* act_log) {
  rollout_random_pair_body<NW, CN, CK, RECORD, ACT>(g, planes, meta, N, T, seed, step0, env_id0, rec_planes, rec_meta, stats,
                                                    act_log);
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

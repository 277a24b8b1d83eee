// mnk_rollout_log.hip -- the fused random rollout with the action log switched on (gfx950 / MI355X only):
// the variants the multi-GPU exchange uses.  Its own translation unit so it compiles beside mnk_rollout.hip.
#include "mnk_host.h"
#include "mnk_rollout_lane.h"

void mnk_launch_rollout_log(const MnkGeom& g, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed,
                            uint64_t step0, int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, int64_t* stats,
                            void* act_log, int act_bytes, void* stream) {
  const int B = 64;
  const dim3 grid((unsigned)((N + B - 1) / B));
  const bool rec = rec_planes && rec_meta;
#define MNK_ROLLOUT(REC, ACTB)                                                                                   \
  MNK_DISPATCH(g, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rollout_random<NW, CN, CK, REC, ACTB>), grid, dim3(B), 0, \
                                     (hipStream_t)stream, g, planes, meta, N, T, seed, step0, env_id0,           \
                                     rec_planes, rec_meta, (unsigned long long*)stats, act_log))
  if (rec && act_bytes == 1) MNK_ROLLOUT(true, 1);
  else if (rec) MNK_ROLLOUT(true, 2);
  else if (act_bytes == 1) MNK_ROLLOUT(false, 1);
  else MNK_ROLLOUT(false, 2);
#undef MNK_ROLLOUT
}

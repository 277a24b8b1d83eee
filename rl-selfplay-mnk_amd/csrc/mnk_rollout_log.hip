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
  // compile-time boards, records on, one wave per SIMD: 32-bit lane offsets for the record stores (see mnk_rollout.hip)
  const bool fixed = (g.n == 9 && g.k == 5 && g.NW == 3) || (g.n == 3 && g.k == 3 && g.NW == 1) ||
                     (g.n == 13 && g.k == 5 && g.NW == 6) || (g.n == 15 && g.k == 5 && g.NW == 8) ||
                     (g.n == 19 && g.k == 5 && g.NW == 12);
  if (rec && fixed && mnk_rollout_saddr_ok(g, N, T)) {
#define MNK_SADDR(NWv, CNv, CKv, ACTB)                                                                                \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rollout_random<NWv, CNv, CKv, true, ACTB, true>), grid, dim3(B), 0,            \
                     (hipStream_t)stream, g, planes, meta, N, T, seed, step0, env_id0, rec_planes, rec_meta,          \
                     (unsigned long long*)stats, act_log)
    if (g.n == 19 && act_bytes == MNK_ACT_U8P1) MNK_SADDR(12, 19, 5, 4);  // 361 cells: a byte and a bit per action ...
    else if (g.n == 19) MNK_SADDR(12, 19, 5, 2);                 // ... or two bytes
    else if (act_bytes == MNK_ACT_BITS7) {                        // 7-bit stream: boards of at most 128 cells
      if (g.n == 9) MNK_SADDR(3, 9, 5, 3);
      else MNK_SADDR(1, 3, 3, 3);
    } else if (act_bytes == 1) {
      if (g.n == 9) MNK_SADDR(3, 9, 5, 1);
      else if (g.n == 3) MNK_SADDR(1, 3, 3, 1);
      else if (g.n == 13) MNK_SADDR(6, 13, 5, 1);
      else MNK_SADDR(8, 15, 5, 1);
    } else {
      if (g.n == 9) MNK_SADDR(3, 9, 5, 2);
      else if (g.n == 3) MNK_SADDR(1, 3, 3, 2);
      else if (g.n == 13) MNK_SADDR(6, 13, 5, 2);
      else MNK_SADDR(8, 15, 5, 2);
    }
#undef MNK_SADDR
    return;
  }
#define MNK_ROLLOUT(REC, ACTB)                                                                                   \
  MNK_DISPATCH16(g, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rollout_random<NW, CN, CK, REC, ACTB>), grid, dim3(B), 0, \
                                     (hipStream_t)stream, g, planes, meta, N, T, seed, step0, env_id0,           \
                                     rec_planes, rec_meta, (unsigned long long*)stats, act_log))
  if (act_bytes == MNK_ACT_U8P1) {  // boards of more than 256 cells: 19x19 and the generic 16-word form
#define MNK_ROLLOUT9(REC)                                                                                          \
  MNK_DISPATCH16_LARGE(g, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rollout_random<NW, CN, CK, REC, 4>), grid, dim3(B), 0, \
                                           (hipStream_t)stream, g, planes, meta, N, T, seed, step0, env_id0,       \
                                           rec_planes, rec_meta, (unsigned long long*)stats, act_log))
    if (rec) MNK_ROLLOUT9(true);
    else MNK_ROLLOUT9(false);
#undef MNK_ROLLOUT9
  } else if (act_bytes == MNK_ACT_BITS7) {
    // boards of at most 128 cells: 9x9, 3x3 and generic boards of up to 8 register words (e.g. 11x11 = 121 cells, NW 5)
#define MNK_ROLLOUT7(REC)                                                                                          \
  MNK_DISPATCH_SMALL(g, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rollout_random<NW, CN, CK, REC, 3>), grid, dim3(B), 0, \
                                           (hipStream_t)stream, g, planes, meta, N, T, seed, step0, env_id0,       \
                                           rec_planes, rec_meta, (unsigned long long*)stats, act_log))
    if (rec) MNK_ROLLOUT7(true);
    else MNK_ROLLOUT7(false);
#undef MNK_ROLLOUT7
  } else if (rec && act_bytes == 1) MNK_ROLLOUT(true, 1);
  else if (rec) MNK_ROLLOUT(true, 2);
  else if (act_bytes == 1) MNK_ROLLOUT(false, 1);
  else MNK_ROLLOUT(false, 2);
#undef MNK_ROLLOUT
}

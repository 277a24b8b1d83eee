// mnk_rollout.hip -- the fused random-policy rollout and the replay of its action log
// (gfx950 / MI355X only).  Separate translation unit: these kernels come in many variants
// (board specialisation x record / action-log forms) and compile in parallel with the rest.
#include "mnk_host.h"
#include "mnk_rollout_lane.h"

// ------------------------------------------------------------------ replay of an action log
// The receiving side of the multi-GPU exchange: a shard's rollout is fully determined by its
// chunk-start state and its action log (1-2 bytes per ply), so that is what crosses xGMI; this
// kernel re-plays the log and rebuilds the full packed records, bit-identical to the sender's.
// ACTB = bytes per logged action (1 or 2), a template parameter like everything else that shapes the ply loop: four
// plies per log word, unrolled with compile-time field positions (round 1 looped ply by ply with a run-time field
// and a branch per ply: 1.2e11 env-steps/s, slower than producing the log); the next word is fetched while the
// current four plies are played.
// ACTB = 3 (MNK_ACT_BITS7): the log is a stream of 7-bit actions; the reader mirrors the writer of mnk_rollout_lane.h --
// a 64-bit accumulator whose fill level is wave-uniform, one u32 word fetched (ahead) whenever fewer than 28 bits remain.
// (the body lives in mnk_rollout_lane.h: boards of more than 16 register words get it compiled at run time, mnk_jit.hip)
template <int NW, int CN, int CK, bool RECORD, int ACTB>
__global__ void __launch_bounds__(64)
k_replay_actions(MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, int T, const void* act_log,
                 uint64_t* rec_planes, uint32_t* rec_meta, int32_t* err) {
  replay_actions_body<NW, CN, CK, RECORD, ACTB>(g, planes, meta, N, T, act_log, rec_planes, rec_meta, err);
}

// ================================================================== C ABI
extern "C" {

int mnk_rollout_random(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k, int T, uint64_t seed,
                       uint64_t step0, int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, int64_t* stats,
                       void* act_log, int act_bytes, void* stream) {
  MnkGeom g;
  int rc = mnk_check_geom(m, n, k, &g);
  if (rc != MNK_OK) return rc;
  if (!planes || !meta || N < 0 || T < 0 || T > 65535 || (!rec_planes != !rec_meta)) return MNK_EINVAL;
  if (act_log && (act_bytes == 0 || !mnk_act_format_ok(act_bytes, g.C))) return MNK_EINVAL;
  if (act_log && (step0 & 3)) return MNK_EINVAL;  // log words hold plies 4q..4q+3 of the Philox step counter
  if (!act_log) act_bytes = 0;
  if (N == 0 || T == 0) return MNK_OK;
  const int B = 64;
  const dim3 grid((unsigned)((N + B - 1) / B));
  // two lanes per env only while both lanes of every env still fit one wave per SIMD (2N <= 65 536 lanes:
  // 9x9x5 106 vs 135 us per 256 plies at 32 768 envs, 164 vs 135 at 36 864; tools/exp_pair_threshold.py)
  // (MNK_ROLLOUT_PAIR=0/1 overrides, for A/B timing)
  const MnkConfig& cfg = mnk_config();  // environment knobs, read once (mnk_reload_config() re-reads them)
  const int pair_override = cfg.pair_override;
  const bool pair_geom = (g.n == 9 && g.k == 5 && g.NW == 3) || (g.n == 3 && g.k == 3 && g.NW == 1) ||
                         (g.n == 13 && g.k == 5 && g.NW == 6) || (g.n == 15 && g.k == 5 && g.NW == 8) ||
                         (g.n == 19 && g.k == 5 && g.NW == 12);
  const bool w_fits = ((int64_t)T * g.NW + 1) * N * 8 < (1ll << 32);  // the two-lane forms' record stores use 32-bit byte offsets
  // (the 7-bit action stream exists in the one-lane form only: a launch that writes one never takes a two-lane form)
  const bool use_pair = pair_geom && w_fits && act_bytes != MNK_ACT_BITS7 &&
                        (pair_override >= 0 ? pair_override != 0 : N <= 32768);
  const bool rec = rec_planes && rec_meta;
  // MNK_ROLLOUT_FORM=lane|pair|pairw|ws2|ws4 forces a kernel form (A/B timing, parity tests of every form)
  const int ws = cfg.form == MNK_FORM_WS2 ? 2 : (cfg.form == MNK_FORM_WS4 ? 4 : 0);
  if (ws && act_bytes != MNK_ACT_BITS7 && mnk_rollout_ws_supported(g, act_bytes)) {
    mnk_launch_rollout_ws(g, ws, planes, meta, N, T, seed, step0, env_id0, rec ? rec_planes : nullptr,
                          rec ? rec_meta : nullptr, stats, act_log, act_bytes, stream);
    return mnk_launch_status("rollout_random_ws");
  }
  // A board without an ahead-of-time specialisation gets one at run time (mnk_jit.hip) once a launch is large
  // enough to pay for the ~1 s of compilation: MNK_JIT=1 always, MNK_JIT=0 never, unset = from 2^20 env-steps per
  // launch (4 096 envs x 256 plies).  If the compile fails the generic kernel below still runs.
  if (!pair_geom) {
    const bool must = g.NW > 16;  // planes of more than 512 bits: no ahead-of-time kernel (mnk_host.h, MNK_DISPATCH16)
    const bool want = must || (cfg.jit >= 0 ? cfg.jit != 0 : (N * (int64_t)T >= (1ll << 20)));
    if (want) {
      // two lanes per env for small batches, like the compile-time boards (same threshold, same results): round 4
      const bool pair_ok = w_fits && act_bytes != MNK_ACT_BITS7 && act_bytes != MNK_ACT_U8P1 && cfg.form != MNK_FORM_LANE &&
                           (pair_override >= 0 ? pair_override != 0 : N <= 32768);
      if (pair_ok)
        if (hipFunction_t fn = mnk_jit_rollout_pair_function(g, rec, act_bytes))
          return mnk_jit_launch_rollout_lanes(fn, g, planes, meta, N, T, seed, step0, env_id0, rec ? rec_planes : nullptr,
                                              rec ? rec_meta : nullptr, stats, act_log, stream, 2);
      if (hipFunction_t fn = mnk_jit_rollout_function(g, rec, act_bytes, rec && mnk_rollout_saddr_ok(g, N, T)))
        return mnk_jit_launch_rollout(fn, g, planes, meta, N, T, seed, step0, env_id0, rec ? rec_planes : nullptr,
                                      rec ? rec_meta : nullptr, stats, act_log, stream);
      if (must) {
        snprintf(g_launch_err, sizeof(g_launch_err), "rollout_random: no kernel for this board: %.200s", mnk_jit_last_error());
        return MNK_ELAUNCH;
      }
    }
  }
  // two lanes per env: split by WORDS on the boards where that measured faster (us per 256 plies at 32 768 envs,
  // by directions / by words: 19x19 216 / 164, 15x15 155 / 135, 13x13 127 / 117; 9x9 86 / 98 stays split by
  // directions); MNK_ROLLOUT_FORM=pair|pairw forces one (pairw at any batch size)
  const bool force_w = cfg.form == MNK_FORM_PAIRW;
  // (the direction-split pair form writes byte / 16-bit logs only: a launch with a bit-packed log takes the word split)
  const bool force_d = cfg.form == MNK_FORM_PAIR && act_bytes != MNK_ACT_U8P1;
  if (mnk_rollout_pairw_supported(g) && w_fits && act_bytes != MNK_ACT_BITS7 &&
      (force_w || (use_pair && !force_d && g.n >= 13))) {
    mnk_launch_rollout_pairw(g, planes, meta, N, T, seed, step0, env_id0, rec ? rec_planes : nullptr,
                             rec ? rec_meta : nullptr, stats, act_log, act_bytes, stream);
    return mnk_launch_status("rollout_random_pairw");
  }
  if (use_pair && act_bytes != MNK_ACT_U8P1) {
    mnk_launch_rollout_pair(g, planes, meta, N, T, seed, step0, env_id0, rec ? rec_planes : nullptr,
                            rec ? rec_meta : nullptr, stats, act_log, act_bytes, stream);
    return mnk_launch_status("rollout_random_pair");
  }
  if (act_bytes) {
    mnk_launch_rollout_log(g, planes, meta, N, T, seed, step0, env_id0, rec ? rec_planes : nullptr,
                           rec ? rec_meta : nullptr, stats, act_log, act_bytes, stream);
    return mnk_launch_status("rollout_random");
  }
  // Record stores as `uniform base + 32-bit lane offset` (SADDR, mnk_rollout_lane.h) while a wave is alone on its SIMD
  // (N <= 65 536: the kernel is bound by its instruction count and this saves ~4 of ~150 per ply: 92.0 -> 88.5 us at
  // the headline size) and one launch's record rows fit 32-bit offsets; from 131 072 envs up the kernel is bound by
  // the HBM write rate and the 64-bit form measured faster (157 vs 166-184 us), so it stays there.
  if (rec && pair_geom && mnk_rollout_saddr_ok(g, N, T)) {
#define MNK_SADDR(NWv, CNv, CKv)                                                                                      \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rollout_random<NWv, CNv, CKv, true, 0, true>), grid, dim3(B), 0,               \
                     (hipStream_t)stream, g, planes, meta, N, T, seed, step0, env_id0, rec_planes, rec_meta,          \
                     (unsigned long long*)stats, act_log)
    if (g.n == 9) MNK_SADDR(3, 9, 5);
    else if (g.n == 3) MNK_SADDR(1, 3, 3);
    else if (g.n == 13) MNK_SADDR(6, 13, 5);
    else if (g.n == 15) MNK_SADDR(8, 15, 5);
    else MNK_SADDR(12, 19, 5);
#undef MNK_SADDR
    return mnk_launch_status("rollout_random");
  }
#define MNK_ROLLOUT(REC)                                                                                       \
  MNK_DISPATCH16(g, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rollout_random<NW, CN, CK, REC, 0>), grid, dim3(B), 0, \
                                     (hipStream_t)stream, g, planes, meta, N, T, seed, step0, env_id0,         \
                                     rec_planes, rec_meta, (unsigned long long*)stats, act_log))
  if (rec) MNK_ROLLOUT(true);
  else MNK_ROLLOUT(false);
#undef MNK_ROLLOUT
  return mnk_launch_status("rollout_random");
}

int mnk_action_log_words(int act_bytes, int T) {
  if (T < 0) return 0;
  const int q = (T + 3) >> 2;
  if (act_bytes == MNK_ACT_U8) return q;
  if (act_bytes == MNK_ACT_U16) return 2 * q;
  if (act_bytes == MNK_ACT_BITS7) return (7 * q + 7) >> 3;
  if (act_bytes == MNK_ACT_U8P1) return q + ((T + 31) >> 5);
  return 0;
}

int mnk_replay_actions(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k, int T, const void* act_log,
                       int act_bytes, uint64_t* rec_planes, uint32_t* rec_meta, int32_t* err, void* stream) {
  MnkGeom g;
  int rc = mnk_check_geom(m, n, k, &g);
  if (rc != MNK_OK) return rc;
  if (!planes || !meta || !act_log || N < 0 || T < 0 || (!rec_planes != !rec_meta)) return MNK_EINVAL;
  if (act_bytes == 0 || !mnk_act_format_ok(act_bytes, g.C)) return MNK_EINVAL;
  if (N == 0 || T == 0) return MNK_OK;
  const int B = 64;
  const dim3 grid((unsigned)((N + B - 1) / B));
  if (g.NW > 16) {  // planes of more than 512 bits: the run-time specialised kernel is the only one (mnk_host.h)
    const bool with_rec = rec_planes && rec_meta;
    if (hipFunction_t fn = mnk_jit_replay_function(g, with_rec, act_bytes))
      return mnk_jit_launch_replay(fn, g, planes, meta, N, T, act_log, with_rec ? rec_planes : nullptr,
                                   with_rec ? rec_meta : nullptr, err, stream);
    snprintf(g_launch_err, sizeof(g_launch_err), "replay_actions: no kernel for this board: %.200s", mnk_jit_last_error());
    return MNK_ELAUNCH;
  }
#define MNK_REPLAY(REC, ACTB)                                                                                       \
  MNK_DISPATCH16(g, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_replay_actions<NW, CN, CK, REC, ACTB>), grid, dim3(B), 0,    \
                                     (hipStream_t)stream, g, planes, meta, N, T, act_log, REC ? rec_planes : nullptr, \
                                     REC ? rec_meta : nullptr, err))
  const bool rec = rec_planes && rec_meta;
  if (act_bytes == MNK_ACT_U8P1) {
#define MNK_REPLAY9(REC)                                                                                               \
  MNK_DISPATCH16_LARGE(g, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_replay_actions<NW, CN, CK, REC, 4>), grid, dim3(B), 0,    \
                                           (hipStream_t)stream, g, planes, meta, N, T, act_log,                        \
                                           REC ? rec_planes : nullptr, REC ? rec_meta : nullptr, err))
    if (rec) MNK_REPLAY9(true);
    else MNK_REPLAY9(false);
#undef MNK_REPLAY9
  } else if (act_bytes == MNK_ACT_BITS7) {
#define MNK_REPLAY7(REC)                                                                                               \
  MNK_DISPATCH_SMALL(g, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_replay_actions<NW, CN, CK, REC, 3>), grid, dim3(B), 0,    \
                                           (hipStream_t)stream, g, planes, meta, N, T, act_log,                        \
                                           REC ? rec_planes : nullptr, REC ? rec_meta : nullptr, err))
    if (rec) MNK_REPLAY7(true);
    else MNK_REPLAY7(false);
#undef MNK_REPLAY7
  } else if (rec && act_bytes == 1) MNK_REPLAY(true, 1);
  else if (rec) MNK_REPLAY(true, 2);
  else if (act_bytes == 1) MNK_REPLAY(false, 1);
  else MNK_REPLAY(false, 2);
#undef MNK_REPLAY
  return mnk_launch_status("replay_actions");
}

}  // extern "C"

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
template <int NW, int CN, int CK, bool RECORD, int ACTB>
__global__ void __launch_bounds__(64)
k_replay_actions(MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, int T, const void* act_log,
                 uint64_t* rec_planes, uint32_t* rec_meta, int32_t* err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  RolloutLane<NW, CN, CK, RECORD> L(g, N, i, rec_planes, rec_meta, nullptr);
  L.load(planes, meta, i);
  if constexpr (ACTB == 3) {
    bool bad7 = false;
    auto play7 = [&](uint32_t a) {
      if (a >= (uint32_t)g.C) { bad7 = true; a = 0; }
      L.ply_action((int)a);
    };
    const int quads = (T + 3) >> 2, nwords = (7 * quads + 7) >> 3;
    const uint32_t* src = (const uint32_t*)act_log + i;
    uint32_t ahead = nwords ? src[0] : 0u;
    int w = 1;
    uint32_t cur = 0, have = 0;  // bits left over from the last word (low-aligned) and their number: 0, 4, ..., 28 (uniform)
    auto take_word = [&]() -> uint32_t {
      const uint32_t word = ahead;
      ahead = src[(int64_t)(w < nwords ? w : nwords - 1) * N];
      ++w;
      return word;
    };
    auto next_quad = [&]() -> uint32_t {  // 32-bit arithmetic only; seven words per eight quads
      uint32_t q;
      if (have == 28u) {
        q = cur;
        cur = 0u;
        have = 0u;
      } else {
        const uint32_t word = take_word();
        q = (cur | (word << have)) & 0x0FFFFFFFu;  // have == 0: cur == 0
        cur = word >> (28u - have);
        have += 4u;
      }
      return q;
    };
    int t = 0;
    for (; t + 4 <= T; t += 4) {
      const uint32_t q = next_quad();
      play7(q & 0x7Fu);
      play7((q >> 7) & 0x7Fu);
      play7((q >> 14) & 0x7Fu);
      play7((q >> 21) & 0x7Fu);
    }
    if (t < T) {
      uint32_t q = next_quad();
      for (; t < T; ++t, q >>= 7) play7(q & 0x7Fu);
    }
    if (bad7) mnk_report(err, MNK_ERR_ACTION_RANGE, i);
    L.store(planes, meta, i);
    return;
  }
  if constexpr (ACTB == 4) {  // MNK_ACT_U8P1: a word of four low bytes per group, a word of 32 high bits per 32 plies
    bool bad9 = false;
    auto play9 = [&](uint32_t a) {
      if (a >= (uint32_t)g.C) { bad9 = true; a = 0; }
      L.ply_action((int)a);
    };
    const int quads = (T + 3) >> 2, hwords = (T + 31) >> 5;
    const uint32_t* lo = (const uint32_t*)act_log + i;
    const uint32_t* hi = lo + (int64_t)quads * N;
    uint32_t ahead = quads ? lo[0] : 0u;
    uint32_t hbits = hwords ? hi[0] : 0u;
    int t = 0;
    for (int q = 0; t < T; ++q) {
      const uint32_t word = ahead;
      ahead = lo[(int64_t)(q + 1 < quads ? q + 1 : q) * N];
      if (q && (q & 7) == 0) hbits = hi[(int64_t)(q >> 3) * N];  // plies 4q .. 4q+3 are bits (4q .. 4q+3) % 32 of word q / 8
      const uint32_t h4 = hbits >> (4 * (q & 7));
      if (t + 4 <= T) {
        play9((word & 0xFFu) | ((h4 & 1u) << 8));
        play9(((word >> 8) & 0xFFu) | ((h4 & 2u) << 7));
        play9(((word >> 16) & 0xFFu) | ((h4 & 4u) << 6));
        play9((word >> 24) | ((h4 & 8u) << 5));
        t += 4;
      } else {
        for (int j = 0; t < T; ++t, ++j) play9(((word >> (8 * j)) & 0xFFu) | (((h4 >> j) & 1u) << 8));
      }
    }
    if (bad9) mnk_report(err, MNK_ERR_ACTION_RANGE, i);
    L.store(planes, meta, i);
    return;
  }
  constexpr uint32_t FIELD = ACTB == 1 ? 0xFFu : 0xFFFFu;
  auto fetch = [&](int q) -> uint64_t {
    if (ACTB == 1) return (uint64_t)((const uint32_t*)act_log)[(int64_t)q * N + i];
    return ((const uint64_t*)act_log)[(int64_t)q * N + i];
  };
  bool bad = false;
  auto play = [&](uint32_t a) {
    if (a >= (uint32_t)g.C) { bad = true; a = 0; }  // a log we did not write: flag it, keep the wave in step
    L.ply_action((int)a);
  };
  const int words = (T + 3) >> 2;
  uint64_t ahead = words ? fetch(0) : 0;
  int t = 0;
  for (int q = 0; t + 4 <= T; ++q, t += 4) {
    const uint64_t quad = ahead;
    ahead = fetch(q + 1 < words ? q + 1 : q);
    play((uint32_t)(quad >> (0 * 8 * ACTB)) & FIELD);
    play((uint32_t)(quad >> (1 * 8 * ACTB)) & FIELD);
    play((uint32_t)(quad >> (2 * 8 * ACTB)) & FIELD);
    play((uint32_t)(quad >> (3 * 8 * ACTB)) & FIELD);
  }
  for (uint64_t quad = ahead; t < T; ++t, quad >>= 8 * ACTB) play((uint32_t)quad & FIELD);  // a partly filled last word
  if (bad) mnk_report(err, MNK_ERR_ACTION_RANGE, i);
  L.store(planes, meta, i);
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
    const bool want = cfg.jit >= 0 ? cfg.jit != 0 : (N * (int64_t)T >= (1ll << 20));
    if (want) {
      if (hipFunction_t fn = mnk_jit_rollout_function(g, rec, act_bytes, rec && mnk_rollout_saddr_ok(g, N, T)))
        return mnk_jit_launch_rollout(fn, g, planes, meta, N, T, seed, step0, env_id0, rec ? rec_planes : nullptr,
                                      rec ? rec_meta : nullptr, stats, act_log, stream);
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
  MNK_DISPATCH(g, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rollout_random<NW, CN, CK, REC, 0>), grid, dim3(B), 0, \
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
#define MNK_REPLAY(REC, ACTB)                                                                                       \
  MNK_DISPATCH(g, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_replay_actions<NW, CN, CK, REC, ACTB>), grid, dim3(B), 0,    \
                                     (hipStream_t)stream, g, planes, meta, N, T, act_log, REC ? rec_planes : nullptr, \
                                     REC ? rec_meta : nullptr, err))
  const bool rec = rec_planes && rec_meta;
  if (act_bytes == MNK_ACT_U8P1) {
#define MNK_REPLAY9(REC)                                                                                               \
  MNK_DISPATCH_LARGE(g, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_replay_actions<NW, CN, CK, REC, 4>), grid, dim3(B), 0,    \
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

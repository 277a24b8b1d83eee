// mnk_selfplay_pre_logits.hip -- mnk_selfplay_pre with the AGENT's masked draw folded in (gfx950 / MI355X only).
// selfplay/policy.py:46-52 + alg/architectures/cnn.py:69-79 + alg/ppo.py:96-97 (mask, softmax, draw, log-probability)
// and selfplay/torch_self_play_wrapper.py:39-59 (reset-or-agent-ply, the opponent's view) in one launch.
#include "mnk_selfplay_draw.h"

extern "C" int mnk_selfplay_pre_logits(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k, const void* logits,
                                       int logits_dtype, const uint8_t* mask, uint64_t sample_seed,
                                       const uint64_t* sample_seed_dev, uint64_t sample_step, const uint64_t* sample_step_dev,
                                       int64_t sample_env_id0, int deterministic, int64_t* actions, float* logp,
                                       const uint8_t* pending, int64_t* agent_side, const int64_t* forced_side, uint64_t seed,
                                       uint64_t step, const uint64_t* step_dev, int64_t env_id0, float* rewards,
                                       uint8_t* terminated, uint8_t* sp_flags, void* opp_obs, int obs_dtype, uint8_t* opp_mask,
                                       int32_t* err, uint32_t flags, void* stream) {
  MnkSpArgs a;
  int rc = mnk_sp_args_pre(&a, planes, meta, N, m, n, k, pending, agent_side, forced_side, seed, step, step_dev, env_id0,
                           rewards, terminated, sp_flags, opp_obs, obs_dtype, opp_mask, err, flags);
  if (rc != MNK_OK) return rc;
  const MnkSample sa = {logits, logits_dtype, mask, sample_seed, sample_seed_dev, sample_step, sample_step_dev, sample_env_id0,
                        deterministic, actions, logp};
  if ((rc = mnk_sample_args_ok(sa, N, a.g.C)) != MNK_OK) return rc;
  if (N == 0) return MNK_OK;
  if (mnk_launch_sp_fused<MNK_SP_PRE>(a, sa, (hipStream_t)stream)) return mnk_launch_status("selfplay_pre_logits");
  // a board without a compile-time draw shape: the draw as a launch of its own, then the actions form
  if ((rc = mnk_launch_sample(sa, N, a.g.C, (hipStream_t)stream)) != MNK_OK) return rc;
  return mnk_selfplay_pre(planes, meta, N, m, n, k, actions, pending, agent_side, forced_side, seed, step, step_dev, env_id0,
                          rewards, terminated, sp_flags, opp_obs, obs_dtype, opp_mask, err, flags, stream);
}

// mnk_selfplay_step_logits.hip -- mnk_selfplay_step_random with the AGENT's masked draw folded in (gfx950 / MI355X only):
// a network agent against the uniformly random opponent (selfplay/policy.py:13-29) is ONE launch per agent-step after the
// forward -- draw + log-probability (policy.py:46-52, cnn.py:69-79, ppo.py:96-97), the whole of
// selfplay/torch_self_play_wrapper.py:32-67 and the next canonical observation (:99-112).
#include "mnk_selfplay_draw.h"

extern "C" int mnk_selfplay_step_random_logits(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k,
                                               const void* logits, int logits_dtype, const uint8_t* mask, uint64_t sample_seed,
                                               const uint64_t* sample_seed_dev, uint64_t sample_step,
                                               const uint64_t* sample_step_dev, int64_t sample_env_id0, int deterministic,
                                               int64_t* actions, float* logp, uint8_t* pending, int64_t* agent_side,
                                               const int64_t* forced_side, uint64_t seed, uint64_t step,
                                               const uint64_t* step_dev, int64_t env_id0, float* rewards, uint8_t* terminated,
                                               void* obs, int obs_dtype, uint8_t* legal_mask, uint64_t* packed_obs, int32_t* err,
                                               float* ep_return, int32_t* ep_length, int64_t* ep_stats, uint32_t flags,
                                               void* stream) {
  MnkSpArgs a;
  int rc = mnk_sp_args_step_random(&a, planes, meta, N, m, n, k, pending, agent_side, forced_side, seed, step, step_dev,
                                   env_id0, rewards, terminated, obs, obs_dtype, legal_mask, packed_obs, err, ep_return,
                                   ep_length, ep_stats, flags);
  if (rc != MNK_OK) return rc;
  const MnkSample sa = {logits, logits_dtype, mask, sample_seed, sample_seed_dev, sample_step, sample_step_dev, sample_env_id0,
                        deterministic, actions, logp};
  if ((rc = mnk_sample_args_ok(sa, N, a.g.C)) != MNK_OK) return rc;
  if (N == 0) return MNK_OK;
  if (mnk_launch_sp_fused<MNK_SP_STEP_RANDOM>(a, sa, (hipStream_t)stream)) return mnk_launch_status("selfplay_step_random_logits");
  if ((rc = mnk_launch_sample(sa, N, a.g.C, (hipStream_t)stream)) != MNK_OK) return rc;
  return mnk_selfplay_step_random(planes, meta, N, m, n, k, actions, pending, agent_side, forced_side, seed, step, step_dev,
                                  env_id0, rewards, terminated, obs, obs_dtype, legal_mask, packed_obs, err, ep_return,
                                  ep_length, ep_stats, flags, stream);
}

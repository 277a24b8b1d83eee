// mnk_selfplay_post_logits.hip -- mnk_selfplay_post with the OPPONENT's masked draw folded in (gfx950 / MI355X only):
// selfplay/torch_self_play_wrapper.py:83-96 (the opponent's policy.act on its view: policy.py:46-52 over the masked head of
// alg/architectures/cnn.py:69-79) + :59-65 (reply, zero-sum merge) + :99-112 (the agent's canonical view) in one launch --
// SURVEY.md section 7 step 5's "[masked sample + opp ply + zero-sum merge + canonical obs]".
#include "mnk_selfplay_draw.h"

extern "C" int mnk_selfplay_post_logits(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k, const void* opp_logits,
                                        int logits_dtype, const uint8_t* opp_mask, uint64_t sample_seed,
                                        const uint64_t* sample_seed_dev, uint64_t sample_step, const uint64_t* sample_step_dev,
                                        int64_t sample_env_id0, int deterministic, int64_t* opp_actions, float* opp_logp,
                                        const uint8_t* sp_flags, const int64_t* agent_side, float* rewards, uint8_t* terminated,
                                        uint8_t* pending, void* obs, int obs_dtype, uint8_t* legal_mask, uint64_t* packed_obs,
                                        int32_t* err, float* ep_return, int32_t* ep_length, int64_t* ep_stats, uint32_t flags,
                                        void* stream) {
  MnkSpArgs a;
  int rc = mnk_sp_args_post(&a, planes, meta, N, m, n, k, sp_flags, agent_side, rewards, terminated, pending, obs, obs_dtype,
                            legal_mask, packed_obs, err, ep_return, ep_length, ep_stats, flags);
  if (rc != MNK_OK) return rc;
  const MnkSample sa = {opp_logits, logits_dtype, opp_mask, sample_seed, sample_seed_dev, sample_step, sample_step_dev,
                        sample_env_id0, deterministic, opp_actions, opp_logp};
  if ((rc = mnk_sample_args_ok(sa, N, a.g.C)) != MNK_OK) return rc;
  if (N == 0) return MNK_OK;
  if (mnk_launch_sp_fused<MNK_SP_POST>(a, sa, (hipStream_t)stream)) return mnk_launch_status("selfplay_post_logits");
  if ((rc = mnk_launch_sample(sa, N, a.g.C, (hipStream_t)stream)) != MNK_OK) return rc;
  return mnk_selfplay_post(planes, meta, N, m, n, k, opp_actions, sp_flags, agent_side, rewards, terminated, pending, obs,
                           obs_dtype, legal_mask, packed_obs, err, ep_return, ep_length, ep_stats, flags, stream);
}

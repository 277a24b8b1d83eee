// mnk_selfplay_kernels.h -- the fused self-play step kernels (gfx950 only), with and without the masked draw folded in.
//
// selfplay/torch_self_play_wrapper.py:32-67 as fixed-shape masked kernels.  The reference builds nonzero() index lists
// (two host syncs each) for "envs to reset", "envs to play" and "envs where the opponent replies"; here every env carries
// those three facts as bits.
//
// Every kernel exists in two forms, selected by its DRAW policy type:
//   NoDraw           the moves come from an int64 array (wrapper.step(actions); an opponent policy that returned actions);
//   Draw<LT, C>      the moves are DRAWN in the kernel from a policy head's raw logits -- selfplay/policy.py:46-52 +
//                    alg/architectures/cnn.py:69-79 (mask -> softmax -> inverse-CDF draw, argmax when deterministic) +
//                    alg/ppo.py:96-97 (log-probability of the drawn action): SURVEY.md section 7 step 5, "[masked sample
//                    + ply + zero-sum merge + canonical obs] in one launch".  All threads of the workgroup first run the
//                    draw of mnk_draw.h on the workgroup's own rows (rows = envs: the logits slab of the workgroup's B
//                    envs is contiguous), leave the chosen cells in LDS (and in `actions` / `logp` in HBM: the rollout
//                    buffer wants them), then the first B lanes play them.  Same (LPR, K) shape as k_sample_logits, same
//                    Philox stream: the action of a row is bit-identical to mnk_sample_logits followed by the NoDraw form.
// Ahead-of-time instantiations of the Draw forms live in mnk_selfplay_{pre,post,step}_logits.hip (boards 3x3, 9x9, 13x13,
// 15x15, 19x19 x f32 / bf16 / no logits); other boards take two launches behind the same C-ABI entry points until
// mnk_jit.hip has compiled this header for them (any row width: mnk_draw::Shape).
// Device code only -- no host includes: hiprtc compiles this text at run time (launchers: mnk_selfplay_host.h).
#pragma once
#include "mnk_api_kernels.h"
#include "mnk_draw.h"

// ------------------------------------------------------------------ the draw folded into a step kernel
struct NoDraw {
  static constexpr bool ON = false;
};
// LT: float / uint16_t (bf16 bits) / void (no logits: uniform over the mask); CC = m * n of a board with a Shape
template <typename LT_, int CC>
struct Draw {
  static constexpr bool ON = true;
  using LT = LT_;
  static constexpr int C = CC, LPR = mnk_draw::Shape<CC>::LPR, K = mnk_draw::Shape<CC>::K;
  static constexpr bool EXACT = mnk_draw::Shape<CC>::EXACT;
};

// LDS the draw needs behind the write-out stage: int act[B] | float slab[rows * C + 2 VE] | float u[rows], rows = NT / LPR
template <typename D>
__host__ __device__ inline size_t mnk_draw_lds_bytes(int B, int nthreads) {
  const int rows = nthreads / D::LPR;
  return (size_t)B * 4 + (mnk_draw::slab_floats<typename D::LT>(rows, D::C) + rows) * sizeof(float);
}

// Draws the moves of the workgroup's envs [env0, env0 + nb) from sa.logits / sa.mask into lds_act[0 .. nb) (and
// sa.actions / sa.logp).  Call from ALL threads under a workgroup-uniform condition; ends with a barrier.
template <typename D>
__device__ __forceinline__ int* mnk_draw_block(const MnkSample& sa, unsigned char* lds_draw, int64_t env0, int nb, int64_t N,
                                               int B, int tid, int nthreads) {
  using namespace mnk_draw;
  using LT = typename D::LT;
  constexpr int C = D::C, LPR = D::LPR, VE = Slab<LT>::VE;
  const int rows = nthreads / LPR;
  int* lds_act = reinterpret_cast<int*>(lds_draw);
  float* lds_l = reinterpret_cast<float*>(lds_draw + (size_t)B * 4);
  float* lds_u = lds_l + slab_floats<LT>(rows, C);
  const uint64_t seed = sa.seed_dev ? *sa.seed_dev : sa.seed;
  const uint64_t step = sa.step + (sa.step_dev ? *sa.step_dev : 0ull);
  const bool vec = aligned16(sa.logits) && aligned16(sa.mask);
  const int r = tid / LPR, sub = tid % LPR;
  for (int p0 = 0; p0 < nb; p0 += rows) {
    const int rows_here = nb - p0 < rows ? nb - p0 : rows;
    const int64_t row0 = env0 + p0;
    const int64_t e0 = row0 * C, e1 = (row0 + rows_here) * C;
    if (p0) __syncthreads();  // the previous pass has been read
    slab_to_lds<LT>(reinterpret_cast<const LT*>(sa.logits), sa.mask, e0, e1, N * C, lds_l, vec, tid, nthreads);
    if (tid < rows_here && !sa.deterministic) lds_u[tid] = row_uniform(seed, (uint64_t)(sa.env_id0 + row0 + tid), step);
    __syncthreads();
    const bool live = r < rows_here;
    const float* lrow = lds_l + (e0 & (VE - 1)) + (size_t)(live ? r : 0) * C;  // idle groups redo row 0
    const Drawn d = draw_row<LPR, D::K, D::EXACT>(lrow, C, lds_u[live ? r : 0], sa.deterministic, tid);
    if (sub == 0 && live) {
      lds_act[p0 + r] = d.chosen;
      sa.actions[row0 + r] = d.chosen;
      if (sa.logp) sa.logp[row0 + r] = d.logp(lrow);
    }
  }
  __syncthreads();
  return lds_act;
}

// the dynamic LDS of a step kernel: [write-out stage, rounded up to 16 B][draw]
__host__ __device__ inline size_t mnk_stage_span(size_t stage_bytes) { return (stage_bytes + 15) & ~(size_t)15; }

// ------------------------------------------------------------------ device-side episode statistics
// Optional device-side episode accounting: what alg/ppo.py:110-120 does
// on the host with dones.any() + nonzero + tolist (two synchronisations per step).  Per env the running
// return and length (in agent-steps, the autoreset step included, as ppo.py:110-111 counts them); on
// termination the episode is classified by its return and folded into replicated counters.
struct MnkEpisodes {
  float* ep_return;            // [N]
  int32_t* ep_length;          // [N]
  unsigned long long* stats;   // [MNK_STATS_REPLICAS][MNK_STATS_STRIDE]: episodes, wins, losses, draws, sum of lengths
};

__device__ __forceinline__ void mnk_ep_account(const MnkEpisodes& ep, int64_t i, float rew, bool term,
                                               unsigned int* lds5) {
  float ret = ep.ep_return[i] + rew;
  int len = ep.ep_length[i] + 1;
  if (term) {
    atomicAdd(&lds5[0], 1u);
    atomicAdd(&lds5[ret > 0.0f ? 1 : (ret < 0.0f ? 2 : 3)], 1u);
    atomicAdd(&lds5[4], (unsigned int)len);
    ret = 0.0f;
    len = 0;
  }
  ep.ep_return[i] = ret;
  ep.ep_length[i] = len;
}

__device__ __forceinline__ void mnk_ep_flush(const MnkEpisodes& ep, const unsigned int* lds5) {
  if (threadIdx.x < MNK_STATS_COUNTERS && lds5[threadIdx.x])
    atomicAdd(&ep.stats[(size_t)(blockIdx.x % MNK_STATS_REPLICAS) * MNK_STATS_STRIDE + threadIdx.x],
              (unsigned long long)lds5[threadIdx.x]);
}

struct SpAgent {
  float reward;
  bool term, was_reset, need_opp;
};

// wrapper:39-63 up to (not including) the opponent's reply, for one env
template <int NW, int CN, int CK>
__device__ __forceinline__ SpAgent sp_agent_half(const MnkGeom& g, MnkEnv<NW>& e, int64_t action, bool pending,
                                                 int64_t& side, const int64_t* forced_side, uint64_t seed,
                                                 uint64_t step, uint64_t env, int64_t i, int32_t* err, bool strict) {
  SpAgent a;
  a.reward = 0.0f; a.term = false; a.was_reset = pending;
  if (pending) {
    env_clear<NW>(e);  // wrapper:41 env.reset(reset_idxs)
    side = forced_side ? (forced_side[i] & 1) : (int64_t)(mnk_rand_u32(seed, env, step, MNK_STREAM_SIDE) >> 31);  // :43-45
  } else {
    const MnkPly ply = env_play<NW, CN, CK>(g, e, action, strict);  // :51
    if (ply.err) mnk_report(err, ply.err, i);
    a.reward = ply.win ? 1.0f : 0.0f;  // :53
    a.term = ply.done;                 // :54
  }
  // :46 / :56-59 -> _opponent_move_if_needed: reply where it is not the agent's turn (:74-77)
  a.need_opp = (a.was_reset || !a.term) && ((int64_t)(e.meta & 1u) != side);
  return a;
}

// ------------------------------------------------------------------ pre: reset-or-agent-ply, the opponent's view
// DRAW: the agent's action is drawn from its policy head's logits (sa; the mask is the one the agent acted on)
template <int NW, int CN, int CK, typename DRAW = NoDraw>
__global__ void __launch_bounds__(256)
k_selfplay_pre(MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, const int64_t* actions, MnkSample sa,
               const uint8_t* pending, int64_t* agent_side, const int64_t* forced_side, uint64_t seed,
               uint64_t step, const uint64_t* step_dev, int64_t env_id0, float* rewards, uint8_t* terminated,
               uint8_t* sp_flags, void* opp_obs, int obs_dtype, uint8_t* opp_mask, int32_t* err, uint32_t flags,
               int vec_ok, int envs_per_block, int stage_span) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  if (step_dev) step += *step_dev;
  const int B = envs_per_block, NT = blockDim.x, tid = threadIdx.x;
  const int64_t env0 = (int64_t)blockIdx.x * B;
  const int64_t i = env0 + tid;
  const int64_t left = N - env0;
  const int nb = left < B ? (int)left : B;
  const bool emit = opp_obs || opp_mask;
  MnkStage st = mnk_stage_carve(lds_raw, g, B);
  if (emit) mnk_stage_tables<CN>(st, g, B, tid, NT);
  const int* drawn = nullptr;
  if constexpr (DRAW::ON) drawn = mnk_draw_block<DRAW>(sa, lds_raw + stage_span, env0, nb, N, B, tid, NT);
  if (tid < B && i < N) {
    MnkEnv<NW> e;
    env_load<NW>(e, planes, meta, N, g.W, i);
    int64_t side = agent_side[i];
    const bool pend = pending[i] != 0;
    int64_t action;
    if constexpr (DRAW::ON) action = drawn[tid];
    else action = actions[i];
    const SpAgent a = sp_agent_half<NW, CN, CK>(g, e, action, pend, side, forced_side, seed, step,
                                        (uint64_t)(env_id0 + i), i, err, (flags & MNK_STEP_STRICT) != 0);
    env_store<NW>(e, planes, meta, N, g.W, i);
    if (pend) agent_side[i] = side;
    rewards[i] = a.reward;
    terminated[i] = a.term ? 1 : 0;
    sp_flags[i] = (a.need_opp ? MNK_SP_NEED_OPP : 0u) | (a.was_reset ? MNK_SP_WAS_RESET : 0u);
    if (emit) {
      // wrapper:83-89: the mover sees itself in channel 0
      const bool white_to_move = (e.meta & 1u) != 0;
      if (white_to_move) mnk_stage_put<NW>(st, g, B, tid, e.p[1], e.p[0], !a.need_opp);
      else mnk_stage_put<NW>(st, g, B, tid, e.p[0], e.p[1], !a.need_opp);
    }
  }
  if (emit)
    mnk_write_out<NW, CN, CK>(st, g, B, nb, mnk_obs_slab(opp_obs, obs_dtype, env0, g.C), obs_dtype,
                              opp_mask ? opp_mask + env0 * g.C : nullptr, vec_ok, tid, NT);
}

// ------------------------------------------------------------------ post: the opponent's ply, merge, the agent's view
// DRAW: the opponent's reply is drawn from ITS policy head's logits (sa; the mask is the one `pre` wrote for it)
template <int NW, int CN, int CK, typename DRAW = NoDraw>
__global__ void __launch_bounds__(256)
k_selfplay_post(MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, const int64_t* opp_actions, MnkSample sa,
                const uint8_t* sp_flags, const int64_t* agent_side, float* rewards, uint8_t* terminated,
                uint8_t* pending, void* obs, int obs_dtype, uint8_t* legal_mask, uint64_t* packed_obs, int32_t* err,
                MnkEpisodes ep, uint32_t flags, int vec_ok, int envs_per_block, int stage_span) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ unsigned int lds_ep[MNK_STATS_COUNTERS];
  const int B = envs_per_block, NT = blockDim.x, tid = threadIdx.x;
  const int64_t env0 = (int64_t)blockIdx.x * B;
  const int64_t i = env0 + tid;
  const int64_t left = N - env0;
  const int nb = left < B ? (int)left : B;
  const bool emit = obs || legal_mask;
  MnkStage st = mnk_stage_carve(lds_raw, g, B);
  if (emit) mnk_stage_tables<CN>(st, g, B, tid, NT);
  if (ep.stats) {
    if (tid < MNK_STATS_COUNTERS) lds_ep[tid] = 0u;
    __syncthreads();
  }
  const int* drawn = nullptr;
  if constexpr (DRAW::ON) drawn = mnk_draw_block<DRAW>(sa, lds_raw + stage_span, env0, nb, N, B, tid, NT);
  if (tid < B && i < N) {
    MnkEnv<NW> e;
    env_load<NW>(e, planes, meta, N, g.W, i);
    const uint32_t f = sp_flags[i];
    float rew = rewards[i];
    bool term = terminated[i] != 0;
    if (f & MNK_SP_NEED_OPP) {
      int64_t action;
      if constexpr (DRAW::ON) action = drawn[tid];
      else action = opp_actions[i];
      const MnkPly ply = env_play<NW, CN, CK>(g, e, action, (flags & MNK_STEP_STRICT) != 0);  // wrapper:96
      if (ply.err) mnk_report(err, ply.err, i);
      else env_store<NW>(e, planes, meta, N, g.W, i);
      if (!(f & MNK_SP_WAS_RESET)) {  // :46 ignores the reply's outcome after a reset
        rew -= ply.win ? 1.0f : 0.0f;  // :62
        term = ply.done;               // :63
      }
      rewards[i] = rew;
      terminated[i] = term ? 1 : 0;
    }
    pending[i] = term ? 1 : 0;  // :65
    if (ep.stats) mnk_ep_account(ep, i, rew, term, lds_ep);
    const bool white = agent_side[i] == 1;  // :104-106
    if (emit) {
      if (white) mnk_stage_put<NW>(st, g, B, tid, e.p[1], e.p[0], true);
      else mnk_stage_put<NW>(st, g, B, tid, e.p[0], e.p[1], true);
    }
    if (packed_obs) mnk_packed_put<NW>(packed_obs, N, g.W, i, white ? e.p[1] : e.p[0], white ? e.p[0] : e.p[1]);
  }
  if (emit) {  // (synchronises: the episode counters in LDS are complete after it, too)
    mnk_write_out<NW, CN, CK>(st, g, B, nb, mnk_obs_slab(obs, obs_dtype, env0, g.C), obs_dtype,
                              legal_mask ? legal_mask + env0 * g.C : nullptr, vec_ok, tid, NT);
  } else if (ep.stats) {
    __syncthreads();
  }
  if (ep.stats) mnk_ep_flush(ep, lds_ep);
}

// ------------------------------------------------------------------ the whole wrapper.step in one launch
// when the opponent is RandomPolicy (policy.py:13-29).  DRAW: the AGENT's action is drawn from its logits -- an
// agent-step of "network agent against the uniformly random opponent" is then this one launch after the forward.
template <int NW, int CN, int CK, typename DRAW = NoDraw>
__global__ void __launch_bounds__(256)
k_selfplay_step_random(MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, const int64_t* actions, MnkSample sa,
                       uint8_t* pending, int64_t* agent_side, const int64_t* forced_side, uint64_t seed,
                       uint64_t step, const uint64_t* step_dev, int64_t env_id0, float* rewards,
                       uint8_t* terminated, void* obs, int obs_dtype, uint8_t* legal_mask, uint64_t* packed_obs,
                       int32_t* err, MnkEpisodes ep, uint32_t flags, int vec_ok, int envs_per_block, int stage_span) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  if (step_dev) step += *step_dev;
  __shared__ unsigned int lds_ep[MNK_STATS_COUNTERS];
  const int B = envs_per_block, NT = blockDim.x, tid = threadIdx.x;
  const int64_t env0 = (int64_t)blockIdx.x * B;
  const int64_t i = env0 + tid;
  const int64_t left = N - env0;
  const int nb = left < B ? (int)left : B;
  const bool emit = obs || legal_mask;
  MnkStage st = mnk_stage_carve(lds_raw, g, B);
  if (emit) mnk_stage_tables<CN>(st, g, B, tid, NT);
  if (ep.stats) {
    if (tid < MNK_STATS_COUNTERS) lds_ep[tid] = 0u;
    __syncthreads();
  }
  const int* drawn = nullptr;
  if constexpr (DRAW::ON) drawn = mnk_draw_block<DRAW>(sa, lds_raw + stage_span, env0, nb, N, B, tid, NT);
  if (tid < B && i < N) {
    MnkEnv<NW> e;
    env_load<NW>(e, planes, meta, N, g.W, i);
    int64_t side = agent_side[i];
    const bool pend = pending[i] != 0;
    const uint64_t env = (uint64_t)(env_id0 + i);
    int64_t action;
    if constexpr (DRAW::ON) action = drawn[tid];
    else action = actions[i];
    SpAgent a = sp_agent_half<NW, CN, CK>(g, e, action, pend, side, forced_side, seed, step, env, i, err,
                                          (flags & MNK_STEP_STRICT) != 0);
    if (a.need_opp) {
      const int oa = env_pick_legal<NW, CN>(g, e, mnk_rand_u32(seed, env, step, MNK_STREAM_OPP));
      const MnkPly ply = env_play<NW, CN, CK, true>(g, e, oa, false);
      if (!a.was_reset) {
        a.reward -= ply.win ? 1.0f : 0.0f;
        a.term = ply.done;
      }
    }
    env_store<NW>(e, planes, meta, N, g.W, i);
    if (pend) agent_side[i] = side;
    rewards[i] = a.reward;
    terminated[i] = a.term ? 1 : 0;
    pending[i] = a.term ? 1 : 0;
    if (ep.stats) mnk_ep_account(ep, i, a.reward, a.term, lds_ep);
    if (emit) {
      if (side == 1) mnk_stage_put<NW>(st, g, B, tid, e.p[1], e.p[0], true);
      else mnk_stage_put<NW>(st, g, B, tid, e.p[0], e.p[1], true);
    }
    if (packed_obs) mnk_packed_put<NW>(packed_obs, N, g.W, i, side == 1 ? e.p[1] : e.p[0], side == 1 ? e.p[0] : e.p[1]);
  }
  if (emit) {  // (synchronises: the episode counters in LDS are complete after it, too)
    mnk_write_out<NW, CN, CK>(st, g, B, nb, mnk_obs_slab(obs, obs_dtype, env0, g.C), obs_dtype,
                              legal_mask ? legal_mask + env0 * g.C : nullptr, vec_ok, tid, NT);
  } else if (ep.stats) {
    __syncthreads();
  }
  if (ep.stats) mnk_ep_flush(ep, lds_ep);
}

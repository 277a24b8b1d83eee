// mnk_selfplay_host.h -- host side of the fused self-play step kernels (mnk_selfplay_kernels.h): argument checks shared
// by the actions and the logits forms of the three C-ABI entry points, and the launcher.
#pragma once
#include "mnk_host.h"
#include "mnk_selfplay_kernels.h"

// ------------------------------------------------------------------ launchers shared by the two translation units
// everything one of the three step kernels takes besides the moves
struct MnkSpArgs {
  MnkGeom g;
  uint64_t* planes;
  uint32_t* meta;
  int64_t N;
  uint8_t* pending;          // pre: read; post / step_random: written
  int64_t* agent_side;
  const int64_t* forced_side;
  uint64_t seed, step;
  const uint64_t* step_dev;
  int64_t env_id0;
  float* rewards;
  uint8_t* terminated;
  uint8_t* sp_flags;         // pre: written; post: read
  void* obs;                 // pre: the opponent's view; post / step_random: the agent's
  int obs_dtype;
  uint8_t* mask;
  uint64_t* packed_obs;
  int32_t* err;
  MnkEpisodes ep;
  uint32_t flags;
};

enum { MNK_SP_PRE = 0, MNK_SP_POST = 1, MNK_SP_STEP_RANDOM = 2 };

// argument checks + MnkSpArgs of the three entry points (shared by their actions and their logits forms)
inline int mnk_sp_args_pre(MnkSpArgs* a, uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k, const uint8_t* pending,
                           int64_t* agent_side, const int64_t* forced_side, uint64_t seed, uint64_t step,
                           const uint64_t* step_dev, int64_t env_id0, float* rewards, uint8_t* terminated, uint8_t* sp_flags,
                           void* opp_obs, int obs_dtype, uint8_t* opp_mask, int32_t* err, uint32_t flags) {
  memset(a, 0, sizeof(*a));
  int rc = mnk_check_geom(m, n, k, &a->g);
  if (rc != MNK_OK) return rc;
  if (!planes || !meta || !pending || !agent_side || !rewards || !terminated || !sp_flags || N < 0 || !mnk_obs_dtype_ok(obs_dtype))
    return MNK_EINVAL;
  a->planes = planes; a->meta = meta; a->N = N; a->pending = const_cast<uint8_t*>(pending); a->agent_side = agent_side;
  a->forced_side = forced_side; a->seed = seed; a->step = step; a->step_dev = step_dev; a->env_id0 = env_id0;
  a->rewards = rewards; a->terminated = terminated; a->sp_flags = sp_flags; a->obs = opp_obs; a->obs_dtype = obs_dtype;
  a->mask = opp_mask; a->err = err; a->flags = flags;
  return MNK_OK;
}

inline int mnk_sp_args_post(MnkSpArgs* a, uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k, const uint8_t* sp_flags,
                            const int64_t* agent_side, float* rewards, uint8_t* terminated, uint8_t* pending, void* obs,
                            int obs_dtype, uint8_t* legal_mask, uint64_t* packed_obs, int32_t* err, float* ep_return,
                            int32_t* ep_length, int64_t* ep_stats, uint32_t flags) {
  memset(a, 0, sizeof(*a));
  int rc = mnk_check_geom(m, n, k, &a->g);
  if (rc != MNK_OK) return rc;
  if (!planes || !meta || !sp_flags || !agent_side || !rewards || !terminated || !pending || N < 0 || !mnk_obs_dtype_ok(obs_dtype))
    return MNK_EINVAL;
  if (ep_stats && (!ep_return || !ep_length)) return MNK_EINVAL;
  a->planes = planes; a->meta = meta; a->N = N; a->pending = pending; a->agent_side = const_cast<int64_t*>(agent_side);
  a->rewards = rewards; a->terminated = terminated; a->sp_flags = const_cast<uint8_t*>(sp_flags); a->obs = obs;
  a->obs_dtype = obs_dtype; a->mask = legal_mask; a->packed_obs = packed_obs; a->err = err;
  a->ep = MnkEpisodes{ep_return, ep_length, (unsigned long long*)ep_stats}; a->flags = flags;
  return MNK_OK;
}

inline int mnk_sp_args_step_random(MnkSpArgs* a, uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k, uint8_t* pending,
                                   int64_t* agent_side, const int64_t* forced_side, uint64_t seed, uint64_t step,
                                   const uint64_t* step_dev, int64_t env_id0, float* rewards, uint8_t* terminated, void* obs,
                                   int obs_dtype, uint8_t* legal_mask, uint64_t* packed_obs, int32_t* err, float* ep_return,
                                   int32_t* ep_length, int64_t* ep_stats, uint32_t flags) {
  memset(a, 0, sizeof(*a));
  int rc = mnk_check_geom(m, n, k, &a->g);
  if (rc != MNK_OK) return rc;
  if (!planes || !meta || !pending || !agent_side || !rewards || !terminated || N < 0 || !mnk_obs_dtype_ok(obs_dtype))
    return MNK_EINVAL;
  if (ep_stats && (!ep_return || !ep_length)) return MNK_EINVAL;
  a->planes = planes; a->meta = meta; a->N = N; a->pending = pending; a->agent_side = agent_side; a->forced_side = forced_side;
  a->seed = seed; a->step = step; a->step_dev = step_dev; a->env_id0 = env_id0; a->rewards = rewards;
  a->terminated = terminated; a->obs = obs; a->obs_dtype = obs_dtype; a->mask = legal_mask; a->packed_obs = packed_obs;
  a->err = err; a->ep = MnkEpisodes{ep_return, ep_length, (unsigned long long*)ep_stats}; a->flags = flags;
  return MNK_OK;
}

// geometry-independent part of a launch of one of the three step kernels
struct MnkSpLaunch {
  int B, NT, vec_ok;
  bool emit;
  dim3 grid, block;
};
inline MnkSpLaunch mnk_sp_launch_shape(const MnkSpArgs& a) {
  MnkSpLaunch l;
  l.B = mnk_block_envs(a.N);
  l.NT = mnk_block_threads();
  l.emit = a.obs || a.mask;
  l.vec_ok = (aligned16(a.obs) ? 1 : 0) | (aligned16(a.mask) ? 2 : 0);
  l.grid = dim3((unsigned)((a.N + l.B - 1) / l.B));
  l.block = dim3(l.NT);
  return l;
}

// one launch of kernel WHICH in its DRAW form; `moves` = the actions array of the NoDraw form
template <int WHICH, int NW, int CN, int CK, typename DRAW>
inline void mnk_launch_sp(const MnkSpArgs& a, const int64_t* moves, const MnkSample& sa, hipStream_t s) {
  const MnkSpLaunch l = mnk_sp_launch_shape(a);
  const int B = l.B, vec_ok = l.vec_ok;
  const size_t stage = l.emit ? mnk_stage_bytes(a.g.NW, a.g.C, B, a.g.n, mnk_geom_packed(a.g.n, a.g.k, a.g.NW, a.g.C)) : 0;
  size_t lds = stage;
  int span = 0;
  if constexpr (DRAW::ON) {
    span = (int)mnk_stage_span(stage);
    lds = (size_t)span + mnk_draw_lds_bytes<DRAW>(B, l.NT);
  }
  const dim3 grid = l.grid, block = l.block;
  if constexpr (WHICH == MNK_SP_PRE)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_selfplay_pre<NW, CN, CK, DRAW>), grid, block, lds, s, a.g, a.planes, a.meta, a.N, moves, sa,
                       a.pending, a.agent_side, a.forced_side, a.seed, a.step, a.step_dev, a.env_id0, a.rewards, a.terminated,
                       a.sp_flags, a.obs, a.obs_dtype, a.mask, a.err, a.flags, vec_ok, B, span);
  else if constexpr (WHICH == MNK_SP_POST)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_selfplay_post<NW, CN, CK, DRAW>), grid, block, lds, s, a.g, a.planes, a.meta, a.N, moves, sa,
                       a.sp_flags, a.agent_side, a.rewards, a.terminated, a.pending, a.obs, a.obs_dtype, a.mask, a.packed_obs,
                       a.err, a.ep, a.flags, vec_ok, B, span);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_selfplay_step_random<NW, CN, CK, DRAW>), grid, block, lds, s, a.g, a.planes, a.meta, a.N,
                       moves, sa, a.pending, a.agent_side, a.forced_side, a.seed, a.step, a.step_dev, a.env_id0, a.rewards,
                       a.terminated, a.obs, a.obs_dtype, a.mask, a.packed_obs, a.err, a.ep, a.flags, vec_ok, B, span);
}

// The same launch through the board's own run-time compiled variant (mnk_jit.hip), when there is one: `moves` != NULL =
// the NoDraw form, else the form that draws from sa (any row width: mnk_draw::Shape).  false = the caller launches the
// ahead-of-time kernel(s).
template <int WHICH>
inline bool mnk_launch_sp_jit(const MnkSpArgs& a, const int64_t* moves, const MnkSample& sa, hipStream_t s) {
  int kind = MNK_JK_SP_PRE + WHICH;
  if (!moves) kind = MNK_JK_SP_DRAW + 3 * (!sa.logits ? 2 : (sa.logits_dtype == MNK_LOGITS_BF16 ? 1 : 0)) + WHICH;
  const MnkSpLaunch l = mnk_sp_launch_shape(a);
  const int B = l.B, vec_ok = l.vec_ok;
  const size_t stage = l.emit ? mnk_stage_bytes(a.g.NW, a.g.C, B, a.g.n, mnk_packed_cells(a.g.n, a.g.C)) : 0;
  size_t lds = stage;
  int span = 0;
  if (!moves) {  // LDS of the draw behind the stage: as mnk_draw_lds_bytes<Draw<LT, C>>, with the shape of this row width
    const int C = a.g.C;
    const int rows = l.NT / mnk_draw::shape_lpr(C);
    const int ve = !sa.logits ? 16 : (sa.logits_dtype == MNK_LOGITS_BF16 ? 8 : 4);
    span = (int)mnk_stage_span(stage);
    lds = (size_t)span + (size_t)B * 4 + ((size_t)rows * C + 2 * ve + rows) * sizeof(float);
  }
  if (lds > MNK_MAX_DYNAMIC_LDS) return false;  // (31x31 with the draw folded in: 92 KB -- its two launches stay)
  hipFunction_t fn = mnk_jit_api_function(a.g, kind, a.N, s);
  if (!fn) return false;
  if constexpr (WHICH == MNK_SP_PRE)
    mnk_module_launch(&k_selfplay_pre<2, 0, 0, NoDraw>, fn, l.grid, l.block, lds, s, a.g, a.planes, a.meta, a.N, moves, sa,
                      a.pending, a.agent_side, a.forced_side, a.seed, a.step, a.step_dev, a.env_id0, a.rewards, a.terminated,
                      a.sp_flags, a.obs, a.obs_dtype, a.mask, a.err, a.flags, vec_ok, B, span);
  else if constexpr (WHICH == MNK_SP_POST)
    mnk_module_launch(&k_selfplay_post<2, 0, 0, NoDraw>, fn, l.grid, l.block, lds, s, a.g, a.planes, a.meta, a.N, moves, sa,
                      a.sp_flags, a.agent_side, a.rewards, a.terminated, a.pending, a.obs, a.obs_dtype, a.mask, a.packed_obs,
                      a.err, a.ep, a.flags, vec_ok, B, span);
  else
    mnk_module_launch(&k_selfplay_step_random<2, 0, 0, NoDraw>, fn, l.grid, l.block, lds, s, a.g, a.planes, a.meta, a.N,
                      moves, sa, a.pending, a.agent_side, a.forced_side, a.seed, a.step, a.step_dev, a.env_id0, a.rewards,
                      a.terminated, a.obs, a.obs_dtype, a.mask, a.packed_obs, a.err, a.ep, a.flags, vec_ok, B, span);
  return true;
}

// mnk_kernels.hip -- HIP kernels + C ABI of libmnk_hip.so (gfx950 / MI355X only).
//
// Replaces, behind include/mnk_hip.h, the ATen op sequences of the reference's
// env/torch_vector_mnk_env.py and selfplay/torch_self_play_wrapper.py (SURVEY.md
// section 2.1).  All of this is integer bit manipulation bounded by HBM traffic:
// no MFMA, no library calls.  Mapping: one lane per env for the game logic
// (coalesced 8-byte accesses over the env axis of the SoA state), one workgroup per
// B consecutive envs, and the same env -> workgroup map in every kernel so an env's
// state stays in the L2 of the XCD that touched it last.
#include "mnk_host.h"
#include "mnk_api_kernels.h"
#include "mnk_selfplay_host.h"

// ------------------------------------------------------------------ reset
__global__ void k_reset_idx(uint64_t* planes, uint32_t* meta, int64_t N, int W, const int64_t* idx, int64_t R,
                            int32_t* err) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= R) return;
  int64_t i = idx[j];
  if (i < 0) i += N;
  if (i < 0 || i >= N) { mnk_report(err, MNK_ERR_ACTION_RANGE, idx[j]); return; }
  for (int w = 0; w < 2 * W; ++w) planes[(int64_t)w * N + i] = 0ull;
  meta[i] = 0u;
}

__global__ void k_reset_mask(uint64_t* planes, uint32_t* meta, int64_t N, int W, const uint8_t* mask) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N || !mask[i]) return;
  for (int w = 0; w < 2 * W; ++w) planes[(int64_t)w * N + i] = 0ull;
  meta[i] = 0u;
}

// dense f32 -> packed; rare path (the writable env.boards view), one lane per env
__global__ void k_pack_boards(MnkGeom g, const float* boards, uint64_t* planes, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  for (int pl = 0; pl < 2; ++pl) {
    const float* src = boards + (i * 2 + pl) * g.C;
    uint64_t word = 0;
    int cur = 0;
    for (int r = 0; r < g.m; ++r)
      for (int c = 0; c < g.n; ++c) {
        const int bit = r * g.stride + c;
        if ((bit >> 6) != cur) {
          planes[(int64_t)(pl * g.W + cur) * N + i] = word;
          word = 0;
          cur = bit >> 6;
        }
        if (src[r * g.n + c] != 0.0f) word |= 1ull << (bit & 63);
      }
    planes[(int64_t)(pl * g.W + cur) * N + i] = word;
    for (int w = cur + 1; w < g.W; ++w) planes[(int64_t)(pl * g.W + w) * N + i] = 0ull;
  }
}

// ------------------------------------------------------------------ GAE (alg/rollout_buffer.py:60-80)
// one lane per env, reverse scan over T; every access is coalesced over the env axis.  The
// operation order and the f32 roundings are the reference's (built with -ffp-contract=off).
// The recurrence is a chain of two dependent f32 ops per step, but the loads do not depend on it: they are
// issued GAE_DEPTH steps ahead (24 loads in flight per lane) -- with one step's loads in flight at a time
// the loop ran at one HBM round trip per step (104 us for 256 x 65 536 instead of ~60 us of traffic).
template <int DEPTH>
struct GaeBatch {
  float r[DEPTH], v[DEPTH];
  uint8_t d[DEPTH];
};

// the DEPTH steps t, t-1, ..., t-DEPTH+1 of env i (all loads independent of the recurrence)
template <int DEPTH>
__device__ __forceinline__ void gae_load(GaeBatch<DEPTH>& b, const float* rewards, const float* values, const uint8_t* dones,
                                         int64_t N, int64_t i, int t) {
#pragma unroll
  for (int j = 0; j < DEPTH; ++j) {
    const int64_t o = (int64_t)(t - j) * N + i;
    b.r[j] = rewards[o];
    b.v[j] = values[o];
    b.d[j] = dones[o];
  }
}

// The recurrence is a chain of two dependent f32 ops per step, but the loads do not depend on it.  Round 1 issued one
// batch of 8 steps' loads, waited for it, computed, stored: one HBM round trip per batch (57-80 us for
// 256 x 65 536 against ~45 us of traffic).  Round 2: the next batch's loads are in flight while the current batch is
// computed and stored (two register sets): the round trip hides behind the arithmetic and the stores.  Round 4: DEPTH is
// a template parameter (MNK_GAE_DEPTH=8|16|32 for A/B): one lane per env gives 1 024 waves at 65 536 envs, and what
// bounds the kernel is the bytes those waves keep in flight (9 B per step and lane).
template <int DEPTH>
__global__ void __launch_bounds__(64)
k_gae(const float* rewards, const float* values, const uint8_t* dones, const float* last_values, int64_t N, int T,
      float gamma, float gamma_lambda, float* advantages, float* returns) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float run = 0.0f;
  float next_v = last_values[i];
  int t = T - 1;
  GaeBatch<DEPTH> cur, nxt;
  if (t >= DEPTH - 1) gae_load<DEPTH>(cur, rewards, values, dones, N, i, t);
  for (; t >= DEPTH - 1; t -= DEPTH) {
    const bool more = t - DEPTH >= DEPTH - 1;
    if (more) gae_load<DEPTH>(nxt, rewards, values, dones, N, i, t - DEPTH);
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      const int64_t o = (int64_t)(t - j) * N + i;
      const float nonterm = 1.0f - (cur.d[j] ? 1.0f : 0.0f);            // :72
      const float delta = cur.r[j] + gamma * next_v * nonterm - cur.v[j];  // :74
      run = delta + gamma_lambda * nonterm * run;                       // :75
      advantages[o] = run;                                              // :77
      returns[o] = run + cur.v[j];                                      // :79
      next_v = cur.v[j];
    }
    if (more) cur = nxt;
  }
  for (; t >= 0; --t) {
    const int64_t o = (int64_t)t * N + i;
    const float v = values[o];
    const float nonterm = 1.0f - (dones[o] ? 1.0f : 0.0f);
    const float delta = rewards[o] + gamma * next_v * nonterm - v;
    run = delta + gamma_lambda * nonterm * run;
    advantages[o] = run;
    returns[o] = run + v;
    next_v = v;
  }
}

// ------------------------------------------------------------------ measurement aid
// A write-only kernel with the store pattern of the rollout records (rec[t][row][N] u64: one wave = 64 consecutive
// envs walking t, 512-byte pieces at stride 8N) and no game logic: what the device sustains for this access pattern.
// bench.py times it next to the roofline (`roofline.measured_write_ceiling_GBps`); tools/exp_write_pattern.hip
// compares it with other layouts.
__global__ void __launch_bounds__(64) k_probe_record_writes(uint64_t* rec, int64_t N, int T, int rows, uint64_t seed) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= N) return;
  uint64_t v = seed + (uint64_t)i;
  uint64_t* p = rec + i;
  for (int t = 0; t < T; ++t)
    for (int r = 0; r < rows; ++r) {
      __builtin_nontemporal_store(v + (uint64_t)r, p);
      p += N;
    }
}

// ================================================================== C ABI
extern "C" {

int mnk_probe_record_writes(uint64_t* rec, int64_t N, int T, int rows, void* stream) {
  if (!rec || N < 0 || T < 0 || rows < 1) return MNK_EINVAL;
  if (N == 0 || T == 0) return MNK_OK;
  hipLaunchKernelGGL(k_probe_record_writes, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, (hipStream_t)stream, rec, N, T,
                     rows, 0x9E3779B97F4A7C15ull);
  return mnk_launch_status("probe_record_writes");
}

int mnk_abi_version(void) { return MNK_ABI_VERSION; }

int mnk_reload_config(void) {
  mnk_config(true);
  return MNK_OK;
}

int mnk_record_words(int m, int n) {
  MnkGeom g;
  if (mnk_check_geom(m, n, 1, &g) != MNK_OK) return 0;
  return g.NW;
}

int mnk_state_words(int m, int n) {
  if (m < 1 || n < 1 || n > 61) return 0;
  const int W = (m * (n + 1) + 63) / 64;
  return W <= MNK_MAX_W ? W : 0;
}

int mnk_geometry_supported(int m, int n, int k) {
  MnkGeom g;
  return mnk_check_geom(m, n, k, &g) == MNK_OK ? 1 : 0;
}

const char* mnk_last_launch_error(void) { return g_launch_err; }

int mnk_reset_all(uint64_t* planes, uint32_t* meta, int64_t N, int W, void* stream) {
  if (!planes || !meta || N < 0 || W < 1 || W > MNK_MAX_W) return MNK_EINVAL;
  if (N == 0) return MNK_OK;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(planes, 0, (size_t)2 * W * N * sizeof(uint64_t), s) != hipSuccess) return mnk_launch_status("reset_all");
  if (hipMemsetAsync(meta, 0, (size_t)N * sizeof(uint32_t), s) != hipSuccess) return mnk_launch_status("reset_all");
  return MNK_OK;
}

int mnk_reset_idx(uint64_t* planes, uint32_t* meta, int64_t N, int W, const int64_t* idx, int64_t R, int32_t* err,
                  void* stream) {
  if (!planes || !meta || N < 0 || R < 0 || W < 1 || W > MNK_MAX_W || (R > 0 && !idx)) return MNK_EINVAL;
  if (R == 0 || N == 0) return MNK_OK;
  const int B = 256;
  hipLaunchKernelGGL(k_reset_idx, dim3((unsigned)((R + B - 1) / B)), dim3(B), 0, (hipStream_t)stream, planes, meta, N,
                     W, idx, R, err);
  return mnk_launch_status("reset_idx");
}

int mnk_reset_mask(uint64_t* planes, uint32_t* meta, int64_t N, int W, const uint8_t* mask, void* stream) {
  if (!planes || !meta || !mask || N < 0 || W < 1 || W > MNK_MAX_W) return MNK_EINVAL;
  if (N == 0) return MNK_OK;
  const int B = 256;
  hipLaunchKernelGGL(k_reset_mask, dim3((unsigned)((N + B - 1) / B)), dim3(B), 0, (hipStream_t)stream, planes, meta, N,
                     W, mask);
  return mnk_launch_status("reset_mask");
}

int mnk_observe(const uint64_t* planes, const uint32_t* meta, int64_t N, int m, int n, const int64_t* flip_side,
                void* obs, int obs_dtype, uint8_t* legal_mask, int fix_empty_mask, uint64_t* packed_obs, void* stream) {
  (void)meta;
  MnkGeom g;
  int rc = mnk_geom_any_k(m, n, &g);
  if (rc != MNK_OK) return rc;
  if (!planes || N < 0 || !mnk_obs_dtype_ok(obs_dtype)) return MNK_EINVAL;
  if (N == 0 || (!obs && !legal_mask && !packed_obs)) return MNK_OK;
  const int B = mnk_block_envs(N);
  const int vec_ok = (aligned16(obs) ? 1 : 0) | (aligned16(legal_mask) ? 2 : 0);
  const dim3 grid((unsigned)((N + B - 1) / B)), block(mnk_block_threads(obs != nullptr));
  hipStream_t s = (hipStream_t)stream;
  const size_t lds_own = mnk_stage_bytes(g.NW, g.C, B, g.n, mnk_packed_cells(g.n, g.C));
  hipFunction_t fn = lds_own <= MNK_MAX_DYNAMIC_LDS ? mnk_jit_api_function(g, MNK_JK_OBSERVE, N, s) : nullptr;
  if (fn) {  // the board's own variant (mnk_jit.hip)
    mnk_module_launch(&k_observe<2, 0, 0>, fn, grid, block, lds_own, s,
                      g, planes, N, flip_side, obs, obs_dtype, legal_mask, fix_empty_mask, packed_obs, vec_ok, B);
    return mnk_launch_status("observe (run-time specialised)");
  }
  const size_t lds = mnk_stage_bytes(g.NW, g.C, B, g.n, mnk_geom_packed(g.n, g.k, g.NW, g.C));
  MNK_DISPATCH(g, hipLaunchKernelGGL(MNK_K(k_observe), grid, block, lds, s, g, planes, N,
                                         flip_side, obs, obs_dtype, legal_mask, fix_empty_mask, packed_obs, vec_ok, B));
  return mnk_launch_status("observe");
}

int mnk_unpack_boards(const uint64_t* planes, float* boards, int64_t N, int m, int n, void* stream) {
  if (!boards) return MNK_EINVAL;
  return mnk_observe(planes, nullptr, N, m, n, nullptr, boards, MNK_OBS_F32, nullptr, 0, nullptr, stream);
}

int mnk_pack_boards(const float* boards, uint64_t* planes, int64_t N, int m, int n, void* stream) {
  MnkGeom g;
  int rc = mnk_geom_any_k(m, n, &g);
  if (rc != MNK_OK) return rc;
  if (!boards || !planes || N < 0) return MNK_EINVAL;
  if (N == 0) return MNK_OK;
  const int B = 64;
  hipLaunchKernelGGL(k_pack_boards, dim3((unsigned)((N + B - 1) / B)), dim3(B), 0, (hipStream_t)stream, g, boards,
                     planes, N);
  return mnk_launch_status("pack_boards");
}

// the full-batch step kernel, with the action read from `actions` or (draw != NULL) drawn by the lane itself
static int mnk_launch_step_full(const MnkGeom& g, uint64_t* planes, uint32_t* meta, int64_t N, const int64_t* actions,
                                const MnkDraw* draw, float* rewards, uint8_t* dones, uint8_t* legal_mask, void* obs,
                                int obs_dtype, int32_t* err, uint32_t flags, hipStream_t s) {
  const int B = mnk_block_envs(N);
  const bool emit = legal_mask || obs;
  const int vec_ok = (aligned16(obs) ? 1 : 0) | (aligned16(legal_mask) ? 2 : 0);
  const dim3 grid((unsigned)((N + B - 1) / B)), block(mnk_block_threads(obs != nullptr));
  const MnkDraw none = {0, 0, nullptr, 0, 0, nullptr};
  const size_t lds_own = emit ? mnk_stage_bytes(g.NW, g.C, B, g.n, mnk_packed_cells(g.n, g.C)) : 0;
  hipFunction_t fn = lds_own <= MNK_MAX_DYNAMIC_LDS ? mnk_jit_api_function(g, draw ? MNK_JK_STEP_DRAW : MNK_JK_STEP, N, s) : nullptr;
  if (fn) {
    mnk_module_launch(&k_step_full<2, 0, 0, false>, fn, grid, block, lds_own, s, g, planes, meta, N, actions, draw ? *draw : none,
                      rewards, dones, legal_mask, obs, obs_dtype, err, flags, vec_ok, B);
    return mnk_launch_status(draw ? "step_random (run-time specialised)" : "step (run-time specialised)");
  }
  const size_t lds = emit ? mnk_stage_bytes(g.NW, g.C, B, g.n, mnk_geom_packed(g.n, g.k, g.NW, g.C)) : 0;
  if (draw)
    MNK_DISPATCH(g, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_step_full<NW, CN, CK, true>), grid, block, lds, s,
                                       g, planes, meta, N, actions, *draw, rewards, dones, legal_mask, obs, obs_dtype, err,
                                       flags, vec_ok, B));
  else
    MNK_DISPATCH(g, hipLaunchKernelGGL(HIP_KERNEL_NAME(k_step_full<NW, CN, CK, false>), grid, block, lds, s,
                                       g, planes, meta, N, actions, none, rewards, dones, legal_mask, obs, obs_dtype, err,
                                       flags, vec_ok, B));
  return mnk_launch_status(draw ? "step_random" : "step");
}

int mnk_step_random(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k, uint64_t seed, uint64_t step,
                    const uint64_t* step_dev, int64_t env_id0, int stream_id, int64_t* actions_out, float* rewards,
                    uint8_t* dones, uint8_t* legal_mask, void* obs, int obs_dtype, uint32_t flags, void* stream) {
  MnkGeom g;
  int rc = mnk_check_geom(m, n, k, &g);
  if (rc != MNK_OK) return rc;
  if (!planes || !meta || !rewards || !dones || N < 0 || stream_id < 0 || stream_id > 255 || !mnk_obs_dtype_ok(obs_dtype))
    return MNK_EINVAL;
  if (flags & ~MNK_STEP_AUTORESET) return MNK_EINVAL;  // a drawn move is legal by construction: nothing to be strict about
  if (N == 0) return MNK_OK;
  const MnkDraw draw = {seed, step, step_dev, env_id0, (uint32_t)stream_id, actions_out};
  return mnk_launch_step_full(g, planes, meta, N, nullptr, &draw, rewards, dones, legal_mask, obs, obs_dtype, nullptr,
                              flags, (hipStream_t)stream);
}

int mnk_step(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k, const int64_t* actions,
             const int64_t* active_idx, int64_t A, float* rewards, uint8_t* dones, uint8_t* legal_mask, void* obs,
             int obs_dtype, int32_t* err, uint32_t flags, void* stream) {
  MnkGeom g;
  int rc = mnk_check_geom(m, n, k, &g);
  if (rc != MNK_OK) return rc;
  if (!planes || !meta || !rewards || !dones || N < 0 || A < 0 || (A > 0 && !actions) || !mnk_obs_dtype_ok(obs_dtype))
    return MNK_EINVAL;
  if (!active_idx && A != N) return MNK_EINVAL;
  if (active_idx && (flags & MNK_STEP_AUTORESET)) return MNK_EINVAL;
  if (N == 0) return MNK_OK;
  hipStream_t s = (hipStream_t)stream;
  if (!active_idx)
    return mnk_launch_step_full(g, planes, meta, N, actions, nullptr, rewards, dones, legal_mask, obs, obs_dtype, err, flags, s);
  // subset: full-size zero rewards / dones (:75, :79), scatter the active ones, then a full observe
  if (hipMemsetAsync(rewards, 0, (size_t)N * sizeof(float), s) != hipSuccess) return mnk_launch_status("step_subset");
  if (hipMemsetAsync(dones, 0, (size_t)N, s) != hipSuccess) return mnk_launch_status("step_subset");
  if (A > 0) {
    const int B = 64;
    const dim3 grid((unsigned)((A + B - 1) / B));
    if (hipFunction_t fn = mnk_jit_api_function(g, MNK_JK_STEP_SUBSET, A, s))
      mnk_module_launch(&k_step_subset<2, 0, 0>, fn, grid, dim3(B), 0, s, g, planes, meta, N, actions, active_idx, A, rewards,
                        dones, err, flags);
    else
      MNK_DISPATCH(g, hipLaunchKernelGGL(MNK_K(k_step_subset), grid, dim3(B), 0, s, g, planes, meta, N, actions,
                                             active_idx, A, rewards, dones, err, flags));
    rc = mnk_launch_status("step_subset");
    if (rc != MNK_OK) return rc;
  }
  if (legal_mask || obs) return mnk_observe(planes, meta, N, m, n, nullptr, obs, obs_dtype, legal_mask, 0, nullptr, stream);
  return MNK_OK;
}

int mnk_sample_legal(const uint64_t* planes, int64_t N, int m, int n, uint64_t seed, uint64_t step,
                     const uint64_t* step_dev, int64_t env_id0, int stream_id, int64_t* actions, void* stream) {
  MnkGeom g;
  int rc = mnk_geom_any_k(m, n, &g);
  if (rc != MNK_OK) return rc;
  if (!planes || !actions || N < 0 || stream_id < 0 || stream_id > 255) return MNK_EINVAL;
  if (N == 0) return MNK_OK;
  const int B = 64;
  const dim3 grid((unsigned)((N + B - 1) / B));
  if (hipFunction_t fn = mnk_jit_api_function(g, MNK_JK_SAMPLE_LEGAL, N, (hipStream_t)stream))
    mnk_module_launch(&k_sample_legal<2, 0, 0>, fn, grid, dim3(B), 0, (hipStream_t)stream, g, planes, N, seed, step, step_dev,
                      env_id0, (uint32_t)stream_id, actions);
  else
    MNK_DISPATCH(g, hipLaunchKernelGGL(MNK_K(k_sample_legal), grid, dim3(B), 0, (hipStream_t)stream, g, planes, N, seed,
                                           step, step_dev, env_id0, (uint32_t)stream_id, actions));
  return mnk_launch_status("sample_legal");
}

int mnk_selfplay_pre(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k, const int64_t* actions,
                     const uint8_t* pending, int64_t* agent_side, const int64_t* forced_side, uint64_t seed,
                     uint64_t step, const uint64_t* step_dev, int64_t env_id0, float* rewards, uint8_t* terminated,
                     uint8_t* sp_flags, void* opp_obs, int obs_dtype, uint8_t* opp_mask, int32_t* err, uint32_t flags,
                     void* stream) {
  MnkSpArgs a;
  int rc = mnk_sp_args_pre(&a, planes, meta, N, m, n, k, pending, agent_side, forced_side, seed, step, step_dev, env_id0,
                           rewards, terminated, sp_flags, opp_obs, obs_dtype, opp_mask, err, flags);
  if (rc != MNK_OK) return rc;
  if (!actions) return MNK_EINVAL;
  if (N == 0) return MNK_OK;
  if (!mnk_launch_sp_jit<MNK_SP_PRE>(a, actions, MnkSample{}, (hipStream_t)stream))
    MNK_DISPATCH(a.g, mnk_launch_sp<MNK_SP_PRE, NW, CN, CK, NoDraw>(a, actions, MnkSample{}, (hipStream_t)stream));
  return mnk_launch_status("selfplay_pre");
}

int mnk_selfplay_post(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k, const int64_t* opp_actions,
                      const uint8_t* sp_flags, const int64_t* agent_side, float* rewards, uint8_t* terminated,
                      uint8_t* pending, void* obs, int obs_dtype, uint8_t* legal_mask, uint64_t* packed_obs, int32_t* err,
                      float* ep_return, int32_t* ep_length, int64_t* ep_stats, uint32_t flags, void* stream) {
  MnkSpArgs a;
  int rc = mnk_sp_args_post(&a, planes, meta, N, m, n, k, sp_flags, agent_side, rewards, terminated, pending, obs, obs_dtype,
                            legal_mask, packed_obs, err, ep_return, ep_length, ep_stats, flags);
  if (rc != MNK_OK) return rc;
  if (!opp_actions) return MNK_EINVAL;
  if (N == 0) return MNK_OK;
  if (!mnk_launch_sp_jit<MNK_SP_POST>(a, opp_actions, MnkSample{}, (hipStream_t)stream))
    MNK_DISPATCH(a.g, mnk_launch_sp<MNK_SP_POST, NW, CN, CK, NoDraw>(a, opp_actions, MnkSample{}, (hipStream_t)stream));
  return mnk_launch_status("selfplay_post");
}

int mnk_selfplay_step_random(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k, const int64_t* actions,
                             uint8_t* pending, int64_t* agent_side, const int64_t* forced_side, uint64_t seed,
                             uint64_t step, const uint64_t* step_dev, int64_t env_id0, float* rewards,
                             uint8_t* terminated, void* obs, int obs_dtype, uint8_t* legal_mask, uint64_t* packed_obs,
                             int32_t* err, float* ep_return, int32_t* ep_length, int64_t* ep_stats, uint32_t flags,
                             void* stream) {
  MnkSpArgs a;
  int rc = mnk_sp_args_step_random(&a, planes, meta, N, m, n, k, pending, agent_side, forced_side, seed, step, step_dev,
                                   env_id0, rewards, terminated, obs, obs_dtype, legal_mask, packed_obs, err, ep_return,
                                   ep_length, ep_stats, flags);
  if (rc != MNK_OK) return rc;
  if (!actions) return MNK_EINVAL;
  if (N == 0) return MNK_OK;
  if (!mnk_launch_sp_jit<MNK_SP_STEP_RANDOM>(a, actions, MnkSample{}, (hipStream_t)stream))
    MNK_DISPATCH(a.g, mnk_launch_sp<MNK_SP_STEP_RANDOM, NW, CN, CK, NoDraw>(a, actions, MnkSample{}, (hipStream_t)stream));
  return mnk_launch_status("selfplay_step_random");
}

int mnk_unpack_records(const uint64_t* rec_planes, const uint32_t* rec_meta, int64_t N, int T, int m, int n,
                       void* obs, int obs_dtype, uint8_t* masks, int64_t* actions, float* rewards, uint8_t* dones,
                       void* stream) {
  MnkGeom g;
  int rc = mnk_geom_any_k(m, n, &g);
  if (rc != MNK_OK) return rc;
  if (!rec_meta || N < 0 || T < 0 || T > 65535 || ((obs || masks) && !rec_planes) || !mnk_obs_dtype_ok(obs_dtype))
    return MNK_EINVAL;
  if (N == 0 || T == 0) return MNK_OK;
  const int B = mnk_block_envs(N * (int64_t)T);
  const bool emit = obs || masks;
  // slabs start at row t*N + env0: 16-byte alignment of every slab needs N*rowbytes % 16 == 0 too
  const bool obs_vec = aligned16(obs) && ((N * 2 * g.C * mnk_obs_bytes(obs_dtype)) % 16 == 0);
  const bool mask_vec = aligned16(masks) && ((N * g.C) % 16 == 0);
  const int vec_ok = (obs_vec ? 1 : 0) | (mask_vec ? 2 : 0);
  const dim3 grid((unsigned)((N + B - 1) / B), (unsigned)T);
  const size_t lds_own = emit ? mnk_stage_bytes(g.NW, g.C, B, g.n, mnk_packed_cells(g.n, g.C)) : 0;
  hipFunction_t fn = lds_own <= MNK_MAX_DYNAMIC_LDS ? mnk_jit_api_function(g, MNK_JK_UNPACK_RECORDS, N * (int64_t)T, (hipStream_t)stream) : nullptr;
  if (fn) {
    mnk_module_launch(&k_unpack_records<2, 0, 0>, fn, grid, dim3(mnk_block_threads()), lds_own, (hipStream_t)stream, g, rec_planes,
                      rec_meta, N, obs, obs_dtype, masks, actions, rewards, dones, vec_ok, B);
    return mnk_launch_status("unpack_records (run-time specialised)");
  }
  const size_t lds = emit ? mnk_stage_bytes(g.NW, g.C, B, g.n, mnk_geom_packed(g.n, g.k, g.NW, g.C)) : 0;
  MNK_DISPATCH(g, hipLaunchKernelGGL(MNK_K(k_unpack_records), grid, dim3(mnk_block_threads()), lds, (hipStream_t)stream, g, rec_planes,
                                         rec_meta, N, obs, obs_dtype, masks, actions, rewards, dones, vec_ok, B));
  return mnk_launch_status("unpack_records");
}

int mnk_gather_obs(const uint64_t* planes, int64_t T, int64_t N, int m, int n, const int64_t* idx, int64_t B,
                   void* obs, int obs_dtype, uint8_t* legal_mask, int fix_empty_mask, int32_t* err, void* stream) {
  MnkGeom g;
  int rc = mnk_geom_any_k(m, n, &g);
  if (rc != MNK_OK) return rc;
  if (!planes || T < 0 || N < 0 || B < 0 || (B > 0 && !idx) || !mnk_obs_dtype_ok(obs_dtype)) return MNK_EINVAL;
  if (B == 0 || (!obs && !legal_mask)) return MNK_OK;
  const int E = mnk_block_envs(B);
  const int vec_ok = (aligned16(obs) ? 1 : 0) | (aligned16(legal_mask) ? 2 : 0);
  const dim3 grid((unsigned)((B + E - 1) / E));
  const size_t lds_own = mnk_stage_bytes(g.NW, g.C, E, g.n, mnk_packed_cells(g.n, g.C));
  hipFunction_t fn = lds_own <= MNK_MAX_DYNAMIC_LDS ? mnk_jit_api_function(g, MNK_JK_GATHER_OBS, B, (hipStream_t)stream) : nullptr;
  if (fn) {
    mnk_module_launch(&k_gather_obs<2, 0, 0>, fn, grid, dim3(mnk_block_threads()),
                      lds_own, (hipStream_t)stream, g, planes, T, N, idx, B,
                      obs, obs_dtype, legal_mask, fix_empty_mask, err, vec_ok, E);
    return mnk_launch_status("gather_obs (run-time specialised)");
  }
  const size_t lds = mnk_stage_bytes(g.NW, g.C, E, g.n, mnk_geom_packed(g.n, g.k, g.NW, g.C));
  MNK_DISPATCH(g, hipLaunchKernelGGL(MNK_K(k_gather_obs), grid, dim3(mnk_block_threads()), lds, (hipStream_t)stream, g,
                                     planes, T, N, idx, B, obs, obs_dtype, legal_mask, fix_empty_mask, err, vec_ok, E));
  return mnk_launch_status("gather_obs");
}

int mnk_gae(const float* rewards, const float* values, const uint8_t* dones, const float* last_values, int64_t N,
            int T, float gamma, float gamma_lambda, float* advantages, float* returns, void* stream) {
  if (!rewards || !values || !dones || !last_values || !advantages || !returns || N < 0 || T < 0) return MNK_EINVAL;
  if (N == 0 || T == 0) return MNK_OK;
  const int B = 64;
  const dim3 grid((unsigned)((N + B - 1) / B));
  const int depth = mnk_config().gae_depth;
#define MNK_GAE(D) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gae<D>), grid, dim3(B), 0, (hipStream_t)stream, rewards, values, dones, \
                                      last_values, N, T, gamma, gamma_lambda, advantages, returns)
  // (A/B at 256 x 65 536, profiles/r04_exp_gae.log: depth 8 49.9 us, 16 59.4, 32 52.7 -- more loads in flight per wave do
  // not help: 8 stays)
  if (depth == 16) MNK_GAE(16);
  else if (depth == 32) MNK_GAE(32);
  else MNK_GAE(8);
#undef MNK_GAE
  return mnk_launch_status("gae");
}

}  // extern "C"

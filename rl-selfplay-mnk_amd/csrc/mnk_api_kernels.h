// mnk_api_kernels.h -- the API-level kernels of the env (gfx950 only): step, step_subset, observe, RandomPolicy's draw,
// records -> RolloutBuffer layout, minibatch gather.  Device code only (no host includes): mnk_kernels.hip instantiates
// them ahead of time for the built-in boards and in generic form, and mnk_jit.hip hands this very text to hiprtc to
// compile a board's own variant (compile-time shifts, packed write-out) once the board is hot.
//
// Mapping: one lane per env for the game logic (coalesced 8-byte accesses over the env axis of the SoA state), one
// workgroup per B consecutive envs, and the same env -> workgroup map in every kernel so an env's state stays in the L2
// of the XCD that touched it last.
#pragma once
#include "mnk_device.h"
#include "mnk_emit.h"

// the view (channel 0, channel 1) of env i as packed planes u64[2][W][N]: what PackedRolloutBuffer stores
template <int NW>
__device__ __forceinline__ void mnk_packed_put(uint64_t* packed, int64_t N, int W, int64_t i, const uint32_t (&ch0)[NW],
                                               const uint32_t (&ch1)[NW]) {
  plane_store<NW>(ch0, packed, N, W, i);
  plane_store<NW>(ch1, packed + (int64_t)W * N, N, W, i);
}

// ------------------------------------------------------------------ step (full batch, fused write-out)
// DRAW: the lane draws its own uniformly random legal move (RandomPolicy, policy.py:18-29) instead of reading
// actions[i] -- mnk_step_random, BASELINE.json config 2 in one launch per ply
struct MnkDraw {
  uint64_t seed, step;
  const uint64_t* step_dev;
  int64_t env_id0;
  uint32_t stream_id;
  int64_t* actions_out;  // optional: the moves played
};

template <int NW, int CN, int CK, bool DRAW>
__global__ void __launch_bounds__(256)
k_step_full(MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, const int64_t* actions, MnkDraw draw, float* rewards,
            uint8_t* dones, uint8_t* legal_mask, void* obs, int obs_dtype, int32_t* err, uint32_t flags, int vec_ok,
            int envs_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int B = envs_per_block, NT = blockDim.x, tid = threadIdx.x;
  const int64_t env0 = (int64_t)blockIdx.x * B;
  const int64_t i = env0 + tid;
  const bool emit = (legal_mask != nullptr) || (obs != nullptr);
  MnkStage st = mnk_stage_carve(lds_raw, g, B);
  if (emit) mnk_stage_tables<CN>(st, g, B, tid, NT);
  if (tid < B && i < N) {
    MnkEnv<NW> e;
    env_load<NW>(e, planes, meta, N, g.W, i);
    MnkPly ply;
    if constexpr (DRAW) {
      const uint64_t step = draw.step + (draw.step_dev ? *draw.step_dev : 0ull);
      const int a = env_pick_legal<NW, CN>(g, e, mnk_rand_u32(draw.seed, (uint64_t)(draw.env_id0 + i), step, draw.stream_id));
      if (draw.actions_out) draw.actions_out[i] = a;
      ply = env_play<NW, CN, CK, true>(g, e, a, false);
    } else {
      ply = env_play<NW, CN, CK>(g, e, actions[i], (flags & MNK_STEP_STRICT) != 0);
    }
    if ((flags & MNK_STEP_AUTORESET) && ply.done) env_clear<NW>(e);  // :34-44 for the envs of nonzero(done)
    if (ply.err) mnk_report(err, ply.err, i);
    else env_store<NW>(e, planes, meta, N, g.W, i);
    rewards[i] = ply.win ? 1.0f : 0.0f;   // :75-77
    dones[i] = ply.done ? 1 : 0;          // :79-80
    if (emit) mnk_stage_put<NW>(st, g, B, tid, e.p[0], e.p[1], false);
  }
  if (emit) {
    const int64_t left = N - env0;
    const int nb = left < B ? (int)left : B;
    mnk_write_out<NW, CN, CK>(st, g, B, nb, mnk_obs_slab(obs, obs_dtype, env0, g.C), obs_dtype,
                              legal_mask ? legal_mask + env0 * g.C : nullptr, vec_ok, tid, NT);
  }
}

// step_subset: lane j plays env active_idx[j]; rewards / dones were zero-filled by the launcher
template <int NW, int CN, int CK>
__global__ void __launch_bounds__(256)
k_step_subset(MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, const int64_t* actions,
              const int64_t* active_idx, int64_t A, float* rewards, uint8_t* dones, int32_t* err,
              uint32_t flags) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A) return;
  int64_t i = active_idx[j];
  if (i < 0) i += N;
  if (i < 0 || i >= N) { mnk_report(err, MNK_ERR_ACTION_RANGE, active_idx[j]); return; }
  MnkEnv<NW> e;
  env_load<NW>(e, planes, meta, N, g.W, i);
  MnkPly ply = env_play<NW, CN, CK>(g, e, actions[j], (flags & MNK_STEP_STRICT) != 0);
  if (ply.err) { mnk_report(err, ply.err, i); return; }
  env_store<NW>(e, planes, meta, N, g.W, i);
  rewards[i] = ply.win ? 1.0f : 0.0f;
  dones[i] = ply.done ? 1 : 0;
}

// ------------------------------------------------------------------ observe / unpack
template <int NW, int CN, int CK>
__global__ void __launch_bounds__(256)
k_observe(MnkGeom g, const uint64_t* planes, int64_t N, const int64_t* flip_side, void* obs, int obs_dtype,
          uint8_t* legal_mask, int fix_empty, uint64_t* packed_obs, int vec_ok, int envs_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int B = envs_per_block, NT = blockDim.x, tid = threadIdx.x;
  const int64_t env0 = (int64_t)blockIdx.x * B;
  const int64_t i = env0 + tid;
  MnkStage st = mnk_stage_carve(lds_raw, g, B);
  mnk_stage_tables<CN>(st, g, B, tid, NT);
  if (tid < B && i < N) {
    uint32_t p0[NW], p1[NW];
    plane_load<NW>(p0, planes, N, g.W, i);
    plane_load<NW>(p1, planes + (int64_t)g.W * N, N, g.W, i);
    const bool flip = flip_side && flip_side[i] == 1;  // wrapper:104-106
    if (flip) mnk_stage_put<NW>(st, g, B, tid, p1, p0, fix_empty != 0);
    else mnk_stage_put<NW>(st, g, B, tid, p0, p1, fix_empty != 0);
    if (packed_obs) mnk_packed_put<NW>(packed_obs, N, g.W, i, flip ? p1 : p0, flip ? p0 : p1);
  }
  if (!obs && !legal_mask) return;  // packed planes only (workgroup-uniform)
  const int64_t left = N - env0;
  const int nb = left < B ? (int)left : B;
  mnk_write_out<NW, CN, CK>(st, g, B, nb, mnk_obs_slab(obs, obs_dtype, env0, g.C), obs_dtype,
                            legal_mask ? legal_mask + env0 * g.C : nullptr, vec_ok, tid, NT);
}

// ------------------------------------------------------------------ RandomPolicy
template <int NW, int CN, int CK>
__global__ void __launch_bounds__(256)
k_sample_legal(MnkGeom g, const uint64_t* planes, int64_t N, uint64_t seed, uint64_t step, const uint64_t* step_dev,
               int64_t env_id0, uint32_t stream_id, int64_t* actions) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (step_dev) step += *step_dev;  // device-resident part of the step counter (graph replays)
  MnkEnv<NW> e;
  plane_load<NW>(e.p[0], planes, N, g.W, i);
  plane_load<NW>(e.p[1], planes + (int64_t)g.W * N, N, g.W, i);
  e.meta = 0u;
  const uint32_t x = mnk_rand_u32(seed, (uint64_t)(env_id0 + i), step, stream_id);
  actions[i] = env_pick_legal<NW, CN>(g, e, x);
}

// ------------------------------------------------------------------ records -> RolloutBuffer layout
template <int NW, int CN, int CK>
__global__ void __launch_bounds__(256)
k_unpack_records(MnkGeom g, const uint64_t* rec_planes, const uint32_t* rec_meta, int64_t N, void* obs, int obs_dtype,
                 uint8_t* masks, int64_t* actions, float* rewards, uint8_t* dones, int vec_ok, int envs_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int B = envs_per_block, NT = blockDim.x, tid = threadIdx.x;
  const int64_t t = blockIdx.y;
  const int64_t env0 = (int64_t)blockIdx.x * B;
  const int64_t i = env0 + tid;
  const bool emit = obs || masks;
  MnkStage st = mnk_stage_carve(lds_raw, g, B);
  if (emit) mnk_stage_tables<CN>(st, g, B, tid, NT);
  if (tid < B && i < N) {
    const uint32_t mw = rec_meta[t * N + i];
    if (actions) actions[t * N + i] = (int64_t)(mw & MNK_REC_ACTION_MASK);
    if (rewards) rewards[t * N + i] = (float)(int8_t)((mw >> MNK_REC_REWARD_SHIFT) & 0xFFu);
    if (dones) dones[t * N + i] = (uint8_t)((mw >> MNK_REC_DONE_BIT) & 1u);
    if (emit) {
      uint32_t p0[NW], p1[NW];
      // a record is already in the mover's view (mover's words low, other side's high): channel 0 = p0
      rec_load<NW>(p0, p1, rec_planes + t * g.NW * N, N, g.NW, i);
      mnk_stage_put<NW>(st, g, B, tid, p0, p1, false);
    }
  }
  if (emit) {
    const int64_t left = N - env0;
    const int nb = left < B ? (int)left : B;
    const int64_t row0 = t * N + env0;
    mnk_write_out<NW, CN, CK>(st, g, B, nb, mnk_obs_slab(obs, obs_dtype, row0, g.C), obs_dtype,
                              masks ? masks + row0 * g.C : nullptr, vec_ok, tid, NT);
  }
}

// ------------------------------------------------------------------ minibatch gather from packed observations
// alg/rollout_buffer.py:82-113 (get_data_loader) indexes f32 [T*N, 2, m, n] observations and bool masks with a
// random permutation -- 729 B read + 729 B written per sample at 9x9.  Here the buffer keeps the packed planes
// (32 B per sample) and this kernel expands the drawn samples straight into the network's input layout:
// sample j = flat id idx[j] = t*N + i; lane j fetches its planes (a 32-byte random gather), the workgroup
// writes its contiguous slab of observations and masks through the LDS stage.
template <int NW, int CN, int CK>
__global__ void __launch_bounds__(256)
k_gather_obs(MnkGeom g, const uint64_t* planes, int64_t T, int64_t N, const int64_t* idx, int64_t B_total, void* obs,
             int obs_dtype, uint8_t* legal_mask, int fix_empty, int32_t* err, int vec_ok, int envs_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int B = envs_per_block, NT = blockDim.x, tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * B;
  const int64_t j = row0 + tid;
  MnkStage st = mnk_stage_carve(lds_raw, g, B);
  mnk_stage_tables<CN>(st, g, B, tid, NT);
  if (tid < B && j < B_total) {
    int64_t flat = idx[j];
    if (flat < 0) flat += T * N;
    uint32_t p0[NW], p1[NW];
    if (flat < 0 || flat >= T * N) {
      mnk_report(err, MNK_ERR_ACTION_RANGE, idx[j]);
#pragma unroll
      for (int w = 0; w < NW; ++w) p0[w] = p1[w] = 0u;
    } else {
      const int64_t t = flat / N, i = flat - t * N;
      const uint64_t* base = planes + t * 2 * g.W * N;
      plane_load<NW>(p0, base, N, g.W, i);
      plane_load<NW>(p1, base + (int64_t)g.W * N, N, g.W, i);
    }
    mnk_stage_put<NW>(st, g, B, tid, p0, p1, fix_empty != 0);
  }
  const int64_t left = B_total - row0;
  const int nb = left < B ? (int)left : B;
  mnk_write_out<NW, CN, CK>(st, g, B, nb, mnk_obs_slab(obs, obs_dtype, row0, g.C), obs_dtype,
                            legal_mask ? legal_mask + row0 * g.C : nullptr, vec_ok, tid, NT);
}

/*
 * mnk_hip.h -- C ABI of libmnk_hip.so, the MI355X (gfx950) implementation of the
 * vectorized MNK self-play rollout path.
 *
 * The reference (michal-szadkowski/rl-selfplay-mnk) has no FFI layer: its boundary
 * is the duck-typed Python surface of TorchVectorMnkEnv / TorchSelfPlayWrapper.
 * The Python classes in rl-selfplay-mnk_amd/{env,selfplay}/ keep that surface and
 * call the functions below through ctypes; every entry point cites the reference
 * code it replaces (paths relative to the reference's src/).
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name says host; buffers are owned
 *     by the caller (torch tensors on the Python side) and must be contiguous;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     every function only enqueues work on it: no allocation, no synchronisation,
 *     no host<->device copy, so all of them may be captured into a hipGraph;
 *   - return value: MNK_OK (0) or a negative MNK_E* code for an argument error that
 *     was detected on the host (nothing is enqueued then);
 *   - data-dependent errors (action out of range, illegal move in strict mode) are
 *     recorded on the device in `err` = int32[2] {code, global env id}; first error
 *     wins, the word is sticky until the caller clears it.  The offending env is
 *     left untouched, all other envs proceed;
 *   - single writer per state buffer: calls that touch the same state must be
 *     ordered on one stream (or by events).  No global mutable state in the library.
 *
 * Packed state:
 *   planes  u64[2][W][N]   plane 0 = black, 1 = white; cell (r,c) is bit r*(n+1)+c of the
 *                          W-word little-endian bit string; column n of each row is a
 *                          guard column that is always 0;  W = mnk_state_words(m, n)
 *   meta    u32[N]         bit 0 = side to move (0 black, 1 white), bits 1..31 = move count
 *
 * Rollout record of one position (what mnk_rollout_random / mnk_replay_actions write per ply):
 *   rows    u64[R][N]      row w = (32-bit word w of the MOVER's plane) | (word w of the other side's plane) << 32,
 *                          same bit numbering as above; R = mnk_record_words(m, n) = ceil(m*(n+1)/32).
 *                          Which colour the mover is says MNK_REC_SIDE_BIT of the ply's meta word (0 = black).
 *                          This is the view the policy sees (channel 0 = own stones, wrapper.py:99-106).
 *                          No padding at any board size: 24 B at 9x9 where two state planes take 32 B --
 *                          the rollout is bound by these stores.
 */
#ifndef MNK_HIP_H
#define MNK_HIP_H

#ifndef __HIPCC_RTC__ /* hiprtc (the run-time specialisation of the rollout kernel) has no libc headers */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define MNK_ABI_VERSION 6

/* status codes (host-side argument checks) */
#define MNK_OK 0
#define MNK_EINVAL -1   /* null pointer / negative size */
#define MNK_EGEOM -2    /* unsupported board geometry (see mnk_geometry_supported) */
#define MNK_ELAUNCH -3  /* hipLaunchKernel failed (hipGetLastError text via mnk_last_launch_error) */
#define MNK_ECOMM -4    /* an RCCL call failed or librccl could not be resolved (text via mnk_comm_last_error) */

/* device-side error codes written to err[0] */
#define MNK_ERR_NONE 0
#define MNK_ERR_ACTION_RANGE 1 /* action outside [-m*n, m*n): the reference raises IndexError (env/torch_vector_mnk_env.py:68) */
#define MNK_ERR_ILLEGAL_MOVE 2 /* strict mode only: occupied cell, message of env/torch_vector_mnk_env.py:102-104 */

/* flags for mnk_step and the mnk_selfplay_* functions */
#define MNK_STEP_STRICT 1u /* refuse moves onto occupied cells (the behaviour tests/test_mnk_integration.py:68-81 expects) */
#define MNK_STEP_AUTORESET 2u /* mnk_step, full batch only: an env whose game this ply finished is reset in the same launch
                               * (env.reset(nonzero(done)), env/torch_vector_mnk_env.py:34-44) and the legal mask / observation
                               * written are those of the fresh board -- the raw loop "step; reset(done); observe" in one launch */

/* element type of the observations a kernel writes (`obs_dtype` next to every `obs` pointer).  A cell is exactly
 * 0 or 1, so every narrowing is lossless: obs.float() of a narrow observation equals the f32 one bit for bit.
 * F32 is what the reference hands out (env/torch_vector_mnk_env.py:17, :52); BF16 is what its first convolution
 * computes in under alg/ppo.py:194 autocast (utils/hardware.py:38-41) -- the cast the caller does today is a separate
 * elementwise kernel over 648 B/env; U8 is 0/1 bytes.  Halves / quarters the dominant bytes of every API kernel. */
#define MNK_OBS_F32 0
#define MNK_OBS_BF16 1
#define MNK_OBS_U8 2

/* element type of the logits handed to mnk_sample_logits */
#define MNK_LOGITS_F32 0
#define MNK_LOGITS_BF16 1 /* what the reference's networks emit under alg/ppo.py:194 autocast on Ampere+ (utils/hardware.py:38-41) */

/* format of the action log (`act_bytes` of mnk_rollout_random / mnk_replay_actions / mnk_jit_compile_rollout) */
#define MNK_ACT_U8 1    /* one byte per action, boards of at most 256 cells: u32[ceil(T/4)][N] */
#define MNK_ACT_U16 2   /* 16 bits per action: u64[ceil(T/4)][N] */
#define MNK_ACT_BITS7 3 /* 7 bits per action, boards of at most 128 cells: a bit stream, ply p at bit 7p, in u32 words
                         * [mnk_action_log_words(MNK_ACT_BITS7, T)][N] -- 0.875 B per env-step, what the ranks of
                         * BASELINE.json configs 2-4 (9x9: 81 cells) put on xGMI */

#define MNK_ACT_U8P1 4  /* 9 bits per action, boards of 257 to 512 cells (19x19: 361; larger boards take MNK_ACT_U16): the low bytes as in MNK_ACT_U8,
                         * u32[ceil(T/4)][N], followed by a bit plane of bit 8 of every action, ply p at bit p % 32 of
                         * word [ceil(T/4) + p / 32][i], u32[ceil(T/32)][N] -- 1.125 B per env-step where MNK_ACT_U16 takes 2
                         * (BASELINE.json config 5's exchange) */

/* bytes of the opaque communicator id exchanged between ranks (= NCCL_UNIQUE_ID_BYTES) */
#define MNK_COMM_ID_BYTES 128

/* flags written by mnk_selfplay_pre for mnk_selfplay_post (u8 per env) */
#define MNK_SP_NEED_OPP 1u
#define MNK_SP_WAS_RESET 2u

/* rollout record meta word (u32 per env-step) */
#define MNK_REC_ACTION_MASK 0xFFFFu
#define MNK_REC_REWARD_SHIFT 16 /* i8 */
#define MNK_REC_DONE_BIT 24
#define MNK_REC_SIDE_BIT 25

/* rollout statistics: int64[MNK_STATS_REPLICAS][MNK_STATS_STRIDE]; a counter's value is the sum of
 * its replicas (column j of every row).  Replication keeps thousands of waves from serialising
 * their atomic adds on one address. */
#define MNK_STATS_REPLICAS 64
#define MNK_STATS_STRIDE 8
#define MNK_STATS_COUNTERS 5 /* episodes finished, black wins, white wins, draws, sum of episode lengths */

/* Philox streams (oracle/philox.py restates the generator) */
#define MNK_STREAM_MOVE 0
#define MNK_STREAM_OPP 1
#define MNK_STREAM_SIDE 2
#define MNK_STREAM_SAMPLE 3

int mnk_abi_version(void);
/* Developer knobs (MNK_ROLLOUT_PAIR, MNK_ROLLOUT_FORM, MNK_JIT, MNK_ROLLOUT_SADDR, MNK_EMIT_ENVS, MNK_EMIT_THREADS: A/B
 * timing and parity tests of every kernel form) are read from the environment once, at the first call that needs them;
 * this reads them again.  Not meant to race with launches from other threads. */
int mnk_reload_config(void);
/* W = ceil(m*(n+1)/64), or 0 when the geometry is unsupported */
int mnk_state_words(int m, int n);
/* R = ceil(m*(n+1)/32), rows of one rollout record; 0 when the geometry is unsupported */
int mnk_record_words(int m, int n);
/* 1 when 1 <= k <= min(m,n), 2 <= n <= 61 and W <= 16 (planes of up to 1 024 bits m*(n+1): 22x22, 25x25, 31x31, 16x61) */
int mnk_geometry_supported(int m, int n, int k);
const char* mnk_last_launch_error(void);

/* ---- env/torch_vector_mnk_env.py:34-44  reset(env_indices=None) ------------------- */
int mnk_reset_all(uint64_t* planes, uint32_t* meta, int64_t N, int W, void* stream);
/* idx: R int64 env indices (negative indices wrap like torch indexing); out-of-range -> err */
int mnk_reset_idx(uint64_t* planes, uint32_t* meta, int64_t N, int W, const int64_t* idx, int64_t R,
                  int32_t* err, void* stream);
/* mask: u8[N], non-zero = reset (fixed-shape form used by the fused paths) */
int mnk_reset_mask(uint64_t* planes, uint32_t* meta, int64_t N, int W, const uint8_t* mask, void* stream);

/* ---- env/torch_vector_mnk_env.py:55-84 + 106-119  step / step_subset / _check_wins ---
 * actions: int64[A]; active_idx: int64[A] ascending unique env ids, or NULL for the full
 * batch (then A must equal N).  rewards f32[N] / dones u8[N] are FULL-SIZE as in the
 * reference (:75-80): zero outside the active set.  legal_mask u8[N][m*n] and
 * obs f32[N][2][m][n] (absolute planes, env/torch_vector_mnk_env.py:46-53) are optional
 * (NULL = skip) and always cover all N envs. */
int mnk_step(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k,
             const int64_t* actions, const int64_t* active_idx, int64_t A,
             float* rewards, uint8_t* dones, uint8_t* legal_mask, void* obs, int obs_dtype,
             int32_t* err, uint32_t flags, void* stream);

/* ---- BASELINE.json config 2 in ONE launch per ply: RandomPolicy.act (selfplay/policy.py:18-29) -> env.step
 * (env/torch_vector_mnk_env.py:55-84, win scan :106-119) -> env.reset(nonzero(done)) (:34-44) -> observe (:46-53).
 * Every env draws its own uniformly random legal move -- Philox(seed, env_id0 + i, step [+ *step_dev], stream_id),
 * the draw of mnk_sample_legal -- plays it, and (flags & MNK_STEP_AUTORESET) restarts if the game ended; rewards /
 * dones / legal_mask / obs as in mnk_step; actions_out (optional) int64[N] receives the moves played.
 * T such launches with step = s .. s+T-1 and MNK_STEP_AUTORESET play the plies of mnk_rollout_random(T, step0 = s). */
int mnk_step_random(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k,
                    uint64_t seed, uint64_t step, const uint64_t* step_dev, int64_t env_id0, int stream_id,
                    int64_t* actions_out, float* rewards, uint8_t* dones, uint8_t* legal_mask, void* obs, int obs_dtype,
                    uint32_t flags, void* stream);

/* ---- env/torch_vector_mnk_env.py:46-53 observe() and
 *      selfplay/torch_self_play_wrapper.py:99-112 _get_canonical_obs() --------------------
 * flip_side: NULL -> absolute planes; else int64[N], envs with flip_side==1 get the two
 * planes swapped (the agent sees itself in channel 0).  fix_empty_mask != 0 sets
 * mask[i][0] = 1 for rows without a legal cell (wrapper:108-110). obs or mask may be NULL. */
int mnk_observe(const uint64_t* planes, const uint32_t* meta, int64_t N, int m, int n,
                const int64_t* flip_side, void* obs, int obs_dtype, uint8_t* legal_mask, int fix_empty_mask,
                uint64_t* packed_obs, void* stream);
/* packed_obs (optional, here and in mnk_selfplay_post / mnk_selfplay_step_random): the same view as `obs` as packed
 * planes u64[2][W][N], channel 0 = the viewer's stones -- 16*W B per env (32 B at 9x9) instead of the 8C + C B of
 * observation + mask; what alg.packed_rollout_buffer.PackedRolloutBuffer stores and mnk_gather_obs expands. */

/* dense (N,2,m,n) f32 <-> packed planes: backs the writable `env.boards` view
 * (tests/test_mnk_integration.py:57-58 pokes stones in).  A cell is a stone when != 0. */
int mnk_pack_boards(const float* boards, uint64_t* planes, int64_t N, int m, int n, void* stream);
int mnk_unpack_boards(const uint64_t* planes, float* boards, int64_t N, int m, int n, void* stream);

/* `step_dev` (functions that draw random numbers): optional device pointer to a u64 that is ADDED to `step`.
 * A captured hipGraph replays its kernel arguments verbatim; keeping the advancing part of the Philox step
 * counter in device memory (bumped by one more node of the graph) lets a captured agent-step be replayed.
 * NULL = the step is `step`. */

/* ---- selfplay/policy.py:13-29 RandomPolicy.act -------------------------------------------
 * One uniformly drawn legal cell per env, from Philox(seed, env_id0 + i, step, stream_id).
 * Rows without a legal cell draw uniformly over all cells (the 1e-8 guard of policy.py:21-24). */
int mnk_sample_legal(const uint64_t* planes, int64_t N, int m, int n, uint64_t seed, uint64_t step,
                     const uint64_t* step_dev, int64_t env_id0, int stream_id, int64_t* actions, void* stream);

/* ---- alg/architectures/cnn.py:69-79 (= resnet.py:84-95, transformer.py:80-91) + policy.py:46-52 + ppo.py:96-97
 * Masked categorical head fused with the draw: logits [N][C] of type `logits_dtype` (MNK_LOGITS_F32: float,
 * MNK_LOGITS_BF16: bf16 bit patterns; any additive normalisation), mask u8[N][C], C <= 1024.
 * logits == NULL: every logit is 0 -- a uniform draw over the legal cells (RandomPolicy, policy.py:13-29) that
 * reads only the mask.  deterministic != 0 -> argmax over legal cells (policy.py:48-49); else an inverse-CDF draw
 * from softmax(masked logits) with one Philox uniform per row (stream MNK_STREAM_SAMPLE).
 * All-masked row -> uniform over C (cnn.py:76-77).
 * logp (optional) = log-probability of the chosen action under the masked softmax (f32 arithmetic).
 * seed_dev (optional): device pointer to a u64 that REPLACES `seed` -- the sampler's Philox key in device memory, so a
 * sampler captured into a hipGraph can be re-keyed without a new capture (a fresh opponent before every rollout,
 * train.py:106-114).  Row i draws from Philox(seed, env_id0 + i, step [+ *step_dev], MNK_STREAM_SAMPLE). */
int mnk_sample_logits(const void* logits, int logits_dtype, const uint8_t* mask, int64_t N, int C, uint64_t seed,
                      const uint64_t* seed_dev, uint64_t step, const uint64_t* step_dev, int64_t env_id0, int deterministic,
                      int64_t* actions, float* logp, void* stream);

/* ---- selfplay/torch_self_play_wrapper.py:32-67 step(), split around the opponent forward ----
 * pre : envs with pending != 0 are reset instead of stepped (their action is ignored), get a
 *       fresh side (forced_side[i] if given, else the top bit of Philox(seed, env, step, SIDE));
 *       the others play the agent's ply.  Writes partial rewards / terminated, the per-env
 *       flags for `post`, and the opponent's view (itself in channel 0, wrapper:83-89) for
 *       every env; rows that need no reply carry their current position and are ignored later.
 * post: envs flagged NEED_OPP play opp_actions; zero-sum merge (wrapper:59-63: reward -= r_opp,
 *       terminated = done_opp, both skipped for freshly reset envs :46); pending = terminated;
 *       writes the agent's canonical observation and mask (wrapper:99-112). */
int mnk_selfplay_pre(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k,
                     const int64_t* actions, const uint8_t* pending, int64_t* agent_side,
                     const int64_t* forced_side, uint64_t seed, uint64_t step, const uint64_t* step_dev,
                     int64_t env_id0, float* rewards, uint8_t* terminated, uint8_t* sp_flags,
                     void* opp_obs, int obs_dtype, uint8_t* opp_mask, int32_t* err, uint32_t flags, void* stream);
int mnk_selfplay_post(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k,
                      const int64_t* opp_actions, const uint8_t* sp_flags, const int64_t* agent_side,
                      float* rewards, uint8_t* terminated, uint8_t* pending,
                      void* obs, int obs_dtype, uint8_t* legal_mask, uint64_t* packed_obs, int32_t* err,
                      float* ep_return, int32_t* ep_length, int64_t* ep_stats, uint32_t flags, void* stream);
/* Every output is caller-owned and may point INTO the rollout sink: obs / legal_mask at row t+1 of the
 * RolloutBuffer's observations / action_masks (alg/rollout_buffer.py:14-44), rewards / terminated at row t of its
 * rewards / dones -- the step then writes each agent-step once, where the reference writes it, reads it back and
 * writes it again in RolloutBuffer.add (alg/rollout_buffer.py:47-58: 7 copy_ per step). */
/* flags: MNK_STEP_STRICT makes an agent / opponent move onto an occupied cell an MNK_ERR_ILLEGAL_MOVE (the env is
 * left untouched) instead of the reference's silent overwrite (env/torch_vector_mnk_env.py:67-69).
 * ep_* (all three or none; NULL = off): device-side episode accounting replacing the host loop of
 * alg/ppo.py:110-120 (dones.any() + nonzero + tolist, two synchronisations per step).  ep_return f32[N] /
 * ep_length i32[N] carry the running return and length (agent-steps) of each env's current episode; when
 * an env terminates its episode is added to ep_stats = int64[MNK_STATS_REPLICAS][MNK_STATS_STRIDE]
 * {episodes, wins (return > 0), losses (< 0), draws, sum of lengths} and the two running values restart. */
/* Whole wrapper.step in ONE launch for a uniformly random opponent (RandomPolicy, policy.py:13-29):
 * pre + Philox legal draw (stream OPP) + post. */
int mnk_selfplay_step_random(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k,
                             const int64_t* actions, uint8_t* pending, int64_t* agent_side,
                             const int64_t* forced_side, uint64_t seed, uint64_t step, const uint64_t* step_dev,
                             int64_t env_id0, float* rewards, uint8_t* terminated, void* obs, int obs_dtype,
                             uint8_t* legal_mask, uint64_t* packed_obs,
                             int32_t* err, float* ep_return, int32_t* ep_length, int64_t* ep_stats,
                             uint32_t flags, void* stream);

/* ---- the step kernels with the masked draw folded in ("[masked sample + opp ply + zero-sum
 * merge + canonical obs]" in one launch).  Each takes, in place of the int64 moves of its plain form, what
 * mnk_sample_logits takes -- logits [N][C] (NULL = uniform over the mask), logits_dtype, mask u8[N][C], the SAMPLER's
 * Philox key and position (sample_seed / sample_seed_dev / sample_step / sample_step_dev / sample_env_id0: independent of
 * the wrapper's own seed / step / env_id0 that follow), deterministic -- draws every row's move in the kernel and plays it:
 *   mnk_selfplay_pre_logits          the AGENT's move from its policy head (selfplay/policy.py:46-52, cnn.py:69-79) with its
 *                                    log-probability (alg/ppo.py:96-97), then wrapper:39-59;
 *   mnk_selfplay_post_logits         the OPPONENT's reply from its head on the view `pre` wrote (wrapper:83-96), then the
 *                                    merge and the agent's canonical view (wrapper:59-65, :99-112);
 *   mnk_selfplay_step_random_logits  the agent's move as in pre, the uniformly random opponent, everything else: one launch
 *                                    per agent-step.
 * `actions` (required) / `logp` (optional) receive the drawn moves and their log-probabilities for ALL N rows (rows whose
 * move is not played -- pending resets, no reply needed -- still draw: the rollout buffer stores them, alg/ppo.py:104).
 * Bit-identical to mnk_sample_logits followed by the plain form.  One launch on 3x3x3, 9x9x5, 13x13x5, 15x15x5 and 19x19x5;
 * other boards take the two launches inside the call until the kernel is hot, then one launch of the board's own run-time
 * compiled variant (ABI 6, below).  A network-vs-network agent-step is 2 env-side launches (was 4). */
int mnk_selfplay_pre_logits(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k, const void* logits,
                            int logits_dtype, const uint8_t* mask, uint64_t sample_seed, const uint64_t* sample_seed_dev,
                            uint64_t sample_step, const uint64_t* sample_step_dev, int64_t sample_env_id0, int deterministic,
                            int64_t* actions, float* logp, const uint8_t* pending, int64_t* agent_side,
                            const int64_t* forced_side, uint64_t seed, uint64_t step, const uint64_t* step_dev,
                            int64_t env_id0, float* rewards, uint8_t* terminated, uint8_t* sp_flags, void* opp_obs,
                            int obs_dtype, uint8_t* opp_mask, int32_t* err, uint32_t flags, void* stream);
int mnk_selfplay_post_logits(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k, const void* opp_logits,
                             int logits_dtype, const uint8_t* opp_mask, uint64_t sample_seed, const uint64_t* sample_seed_dev,
                             uint64_t sample_step, const uint64_t* sample_step_dev, int64_t sample_env_id0, int deterministic,
                             int64_t* opp_actions, float* opp_logp, const uint8_t* sp_flags, const int64_t* agent_side,
                             float* rewards, uint8_t* terminated, uint8_t* pending, void* obs, int obs_dtype,
                             uint8_t* legal_mask, uint64_t* packed_obs, int32_t* err, float* ep_return, int32_t* ep_length,
                             int64_t* ep_stats, uint32_t flags, void* stream);
int mnk_selfplay_step_random_logits(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k, const void* logits,
                                    int logits_dtype, const uint8_t* mask, uint64_t sample_seed,
                                    const uint64_t* sample_seed_dev, uint64_t sample_step, const uint64_t* sample_step_dev,
                                    int64_t sample_env_id0, int deterministic, int64_t* actions, float* logp,
                                    uint8_t* pending, int64_t* agent_side, const int64_t* forced_side, uint64_t seed,
                                    uint64_t step, const uint64_t* step_dev, int64_t env_id0, float* rewards,
                                    uint8_t* terminated, void* obs, int obs_dtype, uint8_t* legal_mask, uint64_t* packed_obs,
                                    int32_t* err, float* ep_return, int32_t* ep_length, int64_t* ep_stats, uint32_t flags,
                                    void* stream);

/* ---- the random-policy rollout of BASELINE.json (RandomPolicy.act -> env.step -> env.reset(done)),
 * T plies per env in one launch with the state held in registers.
 * rec_planes u64[T][R][N]: the position BEFORE each ply (record rows, see the top of this file);
 * rec_meta u32[T][N]: MNK_REC_* word;
 * stats (optional) int64[MNK_STATS_REPLICAS][MNK_STATS_STRIDE] += {episodes finished, black wins,
 * white wins, draws, sum of episode lengths} spread over the replicas (sum the rows to read a counter).
 * rec_planes / rec_meta may be NULL together (state-only rollout); act_log may be NULL. */
int mnk_rollout_random(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k, int T,
                       uint64_t seed, uint64_t step0, int64_t env_id0,
                       uint64_t* rec_planes, uint32_t* rec_meta, int64_t* stats,
                       void* act_log, int act_bytes, void* stream);

/* Boards other than 3x3x3, 9x9x5, 13x13x5, 15x15x5 and 19x19x5 have no ahead-of-time specialisation of the rollout
 * kernel; mnk_rollout_random compiles one with hiprtc (about a second, once per board / record / log-width
 * combination and process) when a launch covers at least 2^20 env-steps -- environment MNK_JIT=1: always, MNK_JIT=0:
 * never (the kernels with run-time geometry then run, 3-5x slower).  Results are identical either way.  The first
 * such launch compiles and loads a code object: make it outside a hipGraph capture (later launches only enqueue).
 * mnk_jit_compile_rollout only compiles (no GPU needed): code object bytes, or a negative status with the
 * compiler's log in mnk_jit_last_error(). */
int64_t mnk_jit_compile_rollout(int m, int n, int k, int record, int act_bytes);
/* Round 4: two more kernels exist as run-time specialisations -- the replay of an action log (mnk_replay_actions) and the
 * two-lanes-per-env form of the rollout (batches of up to 32 768 envs).  Boards whose planes take more than 512 bits
 * (25x25, 31x31 ...) have NO ahead-of-time rollout / replay kernel: theirs are always compiled at run time (~2 s).
 * mnk_jit_compile_kernel is mnk_jit_compile_rollout for any of the three (kind below; act_bytes of a replay = the log
 * format it reads). */
#define MNK_JIT_ROLLOUT 0
#define MNK_JIT_REPLAY 1
#define MNK_JIT_ROLLOUT_PAIR 2
int64_t mnk_jit_compile_kernel(int m, int n, int k, int record, int act_bytes, int kind);
const char* mnk_jit_last_error(void);
/* ABI 6: the API-level kernels are specialised at run time too.  On a board without a built-in variant the kernels behind
 * mnk_step / mnk_step_random / mnk_observe / mnk_sample_legal / mnk_unpack_records / mnk_gather_obs / mnk_selfplay_* start
 * on generic code (run-time shift amounts, table write-out) and switch to the board's own variant -- compile-time
 * geometry, the packed write-out, and for mnk_selfplay_*_logits the draw folded into the step kernel for any row width --
 * once they are hot: 1 024 launches or 2^26 items of that kernel on that board in this process (about a second of hiprtc per
 * kernel, then a code object load).  MNK_JIT_API=1 (or MNK_JIT=1): at the first launch; =0: never.  Results are identical
 * either way.  Nothing is compiled while the launch's stream is being captured into a hipGraph: call mnk_jit_prepare
 * before the capture.  `kind` / the bits of `kinds`: */
#define MNK_JIT_API_STEP 0            /* mnk_step, full batch */
#define MNK_JIT_API_STEP_DRAW 1       /* mnk_step_random */
#define MNK_JIT_API_STEP_SUBSET 2     /* mnk_step with active_idx */
#define MNK_JIT_API_OBSERVE 3         /* mnk_observe, mnk_unpack_boards */
#define MNK_JIT_API_SAMPLE_LEGAL 4    /* mnk_sample_legal */
#define MNK_JIT_API_UNPACK_RECORDS 5  /* mnk_unpack_records */
#define MNK_JIT_API_GATHER_OBS 6      /* mnk_gather_obs */
#define MNK_JIT_API_SP_PRE 7          /* mnk_selfplay_pre */
#define MNK_JIT_API_SP_POST 8         /* mnk_selfplay_post */
#define MNK_JIT_API_SP_STEP 9         /* mnk_selfplay_step_random */
#define MNK_JIT_API_SP_DRAW 10        /* + 3 * logits form (0 f32, 1 bf16, 2 none) + (0 pre, 1 post, 2 step_random):
                                         mnk_selfplay_pre_logits / _post_logits / _step_random_logits */
#define MNK_JIT_API_COUNT 19
/* compiles only (no GPU needed): code object bytes, or a negative status with the log in mnk_jit_last_error() */
int64_t mnk_jit_compile_api(int m, int n, int k, int kind);
/* compiles and loads, on the current device, the variants named by the bits of `kinds` NOW (kinds == 0: of every kernel
 * launched on this board so far -- after a warm-up run, exactly what a capture is going to launch): the number ready, 0
 * for a board with a built-in variant or with MNK_JIT_API / MNK_JIT = 0, or a negative status */
int mnk_jit_prepare(int m, int n, int k, int64_t kinds);
/* Compiled code objects are kept on disk and reused by later processes: directory $MNK_JIT_CACHE (default
 * $XDG_CACHE_HOME/mnk_hip or ~/.cache/mnk_hip; "0" or "off": no cache), one file per (this build's embedded sources, hiprtc
 * version, options, kernel), checksummed, written by rename.  mnk_jit_stats: out4 = {programs compiled by hiprtc, code
 * objects read from the cache instead, written to it, failed compilations} of this process. */
int mnk_jit_stats(int64_t* out4);
/* 1 when the board's own variant of `kind` is loaded on the current device (what the next launch will run), else 0 */
int mnk_jit_api_ready(int m, int n, int k, int kind);

/* The multi-GPU exchange format.  A shard's rollout is a pure function of its chunk-start state and
 * its actions, so the action log, optionally written by mnk_rollout_random, is what ranks all-gather
 * (1-2 B per env-step instead of the 28 B packed record or the reference's 750 B RolloutBuffer row).
 * Layout: four plies per word, act_log u32[ceil(T/4)][N] (act_bytes MNK_ACT_U8: one byte per action, boards with
 * <= 256 cells) or u64[ceil(T/4)][N] (MNK_ACT_U16: 16 bits per action); the action of ply 4q+j is field j
 * (little-endian) of word [q][i]; fields past T are 0.  MNK_ACT_BITS7 (boards with <= 128 cells): 7 bits per action
 * as one bit stream per env, see above.  With a log, step0 must be a multiple of 4 (every chunk but the last a
 * multiple of 4 plies).  A receiver that replays every chunk of a shard in order holds that shard's chunk-start state
 * itself, so after the first chunk the log alone is the message (selfplay/random_rollout.py).
 * mnk_replay_actions re-plays a log from `planes`/`meta` (updated in place, like the rollout) and
 * rebuilds rec_planes / rec_meta bit-identical to what the sender recorded (both may be NULL to only
 * advance the state).  An action >= m*n in the log is reported through err. */
/* 32-bit words per env of a T-ply log in format `act_bytes` (0 for an unknown format): U8 ceil(T/4), U16 2 ceil(T/4),
 * BITS7 ceil(7 ceil(T/4) / 8), U8P1 ceil(T/4) + ceil(T/32) */
int mnk_action_log_words(int act_bytes, int T);
int mnk_replay_actions(uint64_t* planes, uint32_t* meta, int64_t N, int m, int n, int k, int T,
                       const void* act_log, int act_bytes, uint64_t* rec_planes, uint32_t* rec_meta,
                       int32_t* err, void* stream);

/* Unpack gathered records into the reference's RolloutBuffer layout (alg/rollout_buffer.py:14-44):
 * obs f32[T][N][2][m][n] from the mover's point of view, masks u8[T][N][C], actions i64[T][N],
 * rewards f32[T][N], dones u8[T][N].  Any output may be NULL. */
int mnk_unpack_records(const uint64_t* rec_planes, const uint32_t* rec_meta, int64_t N, int T, int m, int n,
                       void* obs, int obs_dtype, uint8_t* masks, int64_t* actions, float* rewards, uint8_t* dones,
                       void* stream);

/* ---- alg/rollout_buffer.py:82-113 get_data_loader: a shuffled minibatch straight from PACKED observations.
 * planes u64[T][2][W][N] (channel 0 = the viewer's stones, as the wrapper hands them out); idx int64[B] flat
 * sample ids t*N + i (negative ids wrap, out-of-range -> err).  Writes the network inputs of the B samples:
 * obs f32[B][2][m][n] and legal mask u8[B][m*n] (free cells; fix_empty_mask as in wrapper:108-110). */
int mnk_gather_obs(const uint64_t* planes, int64_t T, int64_t N, int m, int n, const int64_t* idx, int64_t B,
                   void* obs, int obs_dtype, uint8_t* legal_mask, int fix_empty_mask, int32_t* err, void* stream);

/* ---- alg/rollout_buffer.py:60-80 compute_advantages_and_returns (GAE), one lane per env ---- */
int mnk_gae(const float* rewards, const float* values, const uint8_t* dones, const float* last_values,
            int64_t N, int T, float gamma, float gamma_lambda, float* advantages, float* returns,
            void* stream);

/* ---- the exchange step of the sharded rollout: RCCL all-gather over xGMI ------------
 * The reference has no distributed code; what ranks exchange is the content of its RolloutBuffer
 * (alg/rollout_buffer.py:14-44) in the packed forms above -- either the records themselves or the message
 * "chunk-start planes | action log | chunk-start meta" that mnk_replay_actions expands on the receiver.
 * One process per GPU.  mnk_comm_unique_id (one rank, HOST buffer of MNK_COMM_ID_BYTES) -> the id travels to
 * the other ranks by any host channel (torch.distributed store, a file, MPI) -> every rank calls mnk_comm_init
 * with its rank on its current HIP device -> mnk_allgather_records enqueues ONE all-gather of `bytes` bytes per
 * rank on `stream` (recv holds nranks * bytes; rank r's message lands at recv + r * bytes; in-place allowed
 * when send == recv + rank * bytes) -> mnk_comm_destroy.  Only enqueues: ordering against the rollout kernel
 * is by stream / events, as for every other entry point.  RCCL is resolved from the librccl.so.1 the process
 * has already loaded (PyTorch-ROCm's), not linked.  Errors: MNK_ECOMM + mnk_comm_last_error(). */
int mnk_comm_unique_id(void* id_out_host);
int mnk_comm_init(void** comm_out, const void* id_host, int nranks, int rank);
int mnk_comm_destroy(void* comm);
int mnk_allgather_records(void* comm, const void* send, void* recv, int64_t bytes, void* stream);
/* The same exchange as one send and one receive per peer (ncclSend / ncclRecv in one group, peers in rotated order):
 * on the fully connected xGMI mesh every rank's message then travels once over each of its own links, where a ring
 * all-gather forwards it hop by hop.  Same arguments and result layout; send and recv must not overlap. */
int mnk_allgather_records_direct(void* comm, const void* send, void* recv, int64_t bytes, void* stream);
const char* mnk_comm_last_error(void);
/* NCCL_VERSION_CODE of the resolved library, 0 when none could be resolved */
int mnk_comm_version(void);

/* Measurement aid (bench.py: `roofline.measured_write_ceiling_GBps`): fills rec u64[T][rows][N] with a write-only
 * kernel that has the store pattern of the rollout records and no game logic -- the write rate the device sustains
 * for the access pattern mnk_rollout_random is bound by. */
int mnk_probe_record_writes(uint64_t* rec, int64_t N, int T, int rows, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MNK_HIP_H */

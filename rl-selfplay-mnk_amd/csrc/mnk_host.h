// mnk_host.h -- host-side helpers shared by the translation units of libmnk_hip.so:
// geometry construction, kernel-variant dispatch, launch-status bookkeeping.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <tuple>
#include <utility>

#include "mnk_device.h"
#include "mnk_emit.h"

// ------------------------------------------------------------------ host helpers
inline thread_local char g_launch_err[256] = "";

inline int mnk_make_geom(int m, int n, int k, MnkGeom* g) {
  if (m < 1 || n < 1 || k < 1 || k > m || k > n || n > 61) return MNK_EGEOM;
  const int bits = m * (n + 1);
  const int W = (bits + 63) / 64;
  if (W > MNK_MAX_W) return MNK_EGEOM;
  memset(g, 0, sizeof(*g));
  g->m = m; g->n = n; g->k = k;
  g->C = m * n; g->W = W; g->NW = (bits + 31) / 32; g->stride = n + 1;
  auto magic = [](uint32_t d) { return (uint32_t)((1ull << 32) / d + 1ull); };
  g->magic_n = n == 1 ? 0u : magic((uint32_t)n);  // n == 1: x / 1 handled below
  g->magic_stride = magic((uint32_t)(n + 1));
  g->magic_C = g->C == 1 ? 0u : magic((uint32_t)g->C);
  g->magic_2C = magic((uint32_t)(2 * g->C));
  for (int r = 0; r < m; ++r)
    for (int c = 0; c < n; ++c) {
      const int b = r * (n + 1) + c;
      g->valid[b >> 5] |= 1u << (b & 31);
    }
  return MNK_OK;
}

// division by 1 cannot use the 32-bit magic (2^32 + 1 overflows); boards with n == 1 or a
// single cell are degenerate and rejected instead of carrying a special case in every kernel
inline int mnk_check_geom(int m, int n, int k, MnkGeom* g) {
  int rc = mnk_make_geom(m, n, k, g);
  if (rc != MNK_OK) return rc;
  if (n < 2) return MNK_EGEOM;
  return MNK_OK;
}

// kernels that never look at k (observe, samplers, unpack): hand the dispatcher the k of the
// specialised variant of that board width so they take the compile-time-geometry path too
inline int mnk_geom_any_k(int m, int n, MnkGeom* g) {
  int rc = mnk_check_geom(m, n, 1, g);
  if (rc != MNK_OK) return rc;
  if (n == 3) g->k = 3;
  if (n == 9 || n == 13 || n == 15 || n == 19) g->k = 5;
  return MNK_OK;
}

inline int mnk_launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return MNK_OK;
  snprintf(g_launch_err, sizeof(g_launch_err), "%s: %s", what, hipGetErrorString(e));
  return MNK_ELAUNCH;
}

// Developer knobs from the environment (A/B timing, parity tests of every kernel form), read ONCE -- a launch used to
// cost four or five getenv() calls, which is nothing beside a 90 us rollout launch but not beside a 5 us step.
// mnk_reload_config() (C ABI; mnk_hip.reload_config() in Python) reads them again after the environment has changed.
struct MnkConfig {
  int pair_override = -1;  // MNK_ROLLOUT_PAIR=0/1: never / always two lanes per env (unset: by batch size)
  int form = 0;            // MNK_ROLLOUT_FORM=lane|pair|pairw|ws2|ws4 -> 1..5 (unset / unknown: 0)
  int jit = -1;            // MNK_JIT=0/1 (unset: run-time specialisation from 2^20 env-steps per launch)
  int jit_api = -1;        // MNK_JIT_API=0/1: the same for the API-level kernels alone (unset: what MNK_JIT says; both unset:
                           // a board's own variant is compiled once the kernel is hot, mnk_jit_api_function)
  bool saddr_off = false;  // MNK_ROLLOUT_SADDR=0: no 32-bit-offset record stores
  int emit_envs = 0;       // MNK_EMIT_ENVS=16|32|64|128: envs per workgroup of the write-out kernels (0: by batch size)
  int emit_threads = 0;    // MNK_EMIT_THREADS=64|128|256 (the kernels are __launch_bounds__(256); 0: by output set)
  int gae_depth = 0;       // MNK_GAE_DEPTH=8|16|32: steps of loads mnk_gae keeps in flight per batch (0: the default)
};
enum { MNK_FORM_NONE = 0, MNK_FORM_LANE, MNK_FORM_PAIR, MNK_FORM_PAIRW, MNK_FORM_WS2, MNK_FORM_WS4 };

inline MnkConfig mnk_read_config() {
  MnkConfig c;
  if (const char* v = getenv("MNK_ROLLOUT_PAIR")) c.pair_override = atoi(v) != 0 ? 1 : 0;
  if (const char* v = getenv("MNK_ROLLOUT_FORM")) {
    static const char* names[] = {"", "lane", "pair", "pairw", "ws2", "ws4"};
    for (int f = 1; f <= 5; ++f)
      if (!strcmp(v, names[f])) c.form = f;
  }
  if (const char* v = getenv("MNK_JIT")) c.jit = atoi(v) != 0 ? 1 : 0;
  c.jit_api = c.jit;
  if (const char* v = getenv("MNK_JIT_API")) c.jit_api = atoi(v) != 0 ? 1 : 0;
  if (const char* v = getenv("MNK_ROLLOUT_SADDR")) c.saddr_off = atoi(v) == 0;
  if (const char* v = getenv("MNK_EMIT_ENVS")) {
    const int t = atoi(v);
    c.emit_envs = (t == 16 || t == 32 || t == 64 || t == 128) ? t : 0;
  }
  if (const char* v = getenv("MNK_GAE_DEPTH")) {
    const int t = atoi(v);
    c.gae_depth = (t == 8 || t == 16 || t == 32) ? t : 0;
  }
  if (const char* v = getenv("MNK_EMIT_THREADS")) {
    const int t = atoi(v);
    c.emit_threads = (t == 64 || t == 128 || t == 256) ? t : 0;  // anything else would break the launch bounds
  }
  return c;
}

// (the first call initialises the function-local static: thread-safe; a reload is the caller's to serialise)
inline const MnkConfig& mnk_config(bool reload = false) {
  static MnkConfig cfg = mnk_read_config();
  if (reload) cfg = mnk_read_config();
  return cfg;
}

// envs per workgroup of the kernels with a write-out stage; `items` = envs (x plies for mnk_unpack_records) of the launch.
// 64 by default; 32 while that still leaves fewer than 1 024 workgroups (19x19x5 x 32 768 envs: 25.5 vs 26.9 us for
// the fused self-play step, tools/exp_kernels.py).  MNK_EMIT_ENVS=16|32|64 forces one (read once per process).
inline int mnk_block_envs(int64_t items) {
  if (const int forced = mnk_config().emit_envs) return forced;
  return items <= 32768 ? 32 : 64;
}

// threads per workgroup of those kernels: the first 64 lanes play their envs, then all waves of
// the workgroup sweep its output slab (more waves per SIMD to hide the LDS / store latency)
// 256 when an observation is written (1.3-1.7x faster than 64, tools/exp_emit.py); 128 when only the legal mask leaves
// (an eighth of the bytes: two waves sweep it as fast as four and start sooner -- mnk_step_random at 65 536 envs 4.77 ->
// 4.68 us per ply, at 262 144 envs 11.0 -> 9.5, tools/exp_one_launch.py)
// Never fewer threads than envs per workgroup: lane tid plays env env0 + tid, so with MNK_EMIT_ENVS=128 a 64-thread
// workgroup would leave envs 64..127 of every workgroup unplayed and emit them from an unfilled stage.
inline int mnk_block_threads(bool writes_obs = true) {
  const int envs = mnk_config().emit_envs;  // 0 unless forced; the default 32 / 64 never exceeds the defaults below
  int threads = writes_obs ? 256 : 128;
  if (const int forced = mnk_config().emit_threads) threads = forced;
  return threads < envs ? envs : threads;
}

// Kernel variants: NW = u32 register words per plane; CN / CK = compile-time board width and
// run length (0 = run time).  The boards people actually train on get fully specialised code
// (immediate shift amounts, unrolled run doubling); everything else takes the generic form.
#define MNK_CASE(NWv, CNv, CKv, ...)                      \
  {                                                       \
    constexpr int NW = NWv, CN = CNv, CK = CKv;           \
    __VA_ARGS__;                                          \
  }
#define MNK_DISPATCH(g, ...)                                                          \
  do {                                                                                \
    if ((g).n == 9 && (g).k == 5 && (g).NW == 3) MNK_CASE(3, 9, 5, __VA_ARGS__)       \
    else if ((g).n == 3 && (g).k == 3 && (g).NW == 1) MNK_CASE(1, 3, 3, __VA_ARGS__)  \
    else if ((g).n == 13 && (g).k == 5 && (g).NW == 6) MNK_CASE(6, 13, 5, __VA_ARGS__) \
    else if ((g).n == 15 && (g).k == 5 && (g).NW == 8) MNK_CASE(8, 15, 5, __VA_ARGS__) \
    else if ((g).n == 19 && (g).k == 5 && (g).NW == 12) MNK_CASE(12, 19, 5, __VA_ARGS__) \
    else if ((g).NW <= 2) MNK_CASE(2, 0, 0, __VA_ARGS__)                              \
    else if ((g).NW <= 4) MNK_CASE(4, 0, 0, __VA_ARGS__)                              \
    else if ((g).NW <= 8) MNK_CASE(8, 0, 0, __VA_ARGS__)                              \
    else if ((g).NW <= 16) MNK_CASE(16, 0, 0, __VA_ARGS__)                            \
    else MNK_CASE(32, 0, 0, __VA_ARGS__)                                              \
  } while (0)
#define MNK_K(name) HIP_KERNEL_NAME(name<NW, CN, CK>)
// the variants a board of more than 256 cells can have (MNK_ACT_U8P1, up to 512 cells): 19x19x5 and the generic 16- and
// 32-word forms
#define MNK_DISPATCH_LARGE(g, ...)                                                       \
  do {                                                                                   \
    if ((g).n == 19 && (g).k == 5 && (g).NW == 12) MNK_CASE(12, 19, 5, __VA_ARGS__)      \
    else if ((g).NW <= 16) MNK_CASE(16, 0, 0, __VA_ARGS__)                               \
    else MNK_CASE(32, 0, 0, __VA_ARGS__)                                                 \
  } while (0)
// the variants a board of at most 128 cells can have (the 7-bit action stream): 9x9x5, 3x3x3, generic up to 8 words
#define MNK_DISPATCH_SMALL(g, ...)                                                    \
  do {                                                                                \
    if ((g).n == 9 && (g).k == 5 && (g).NW == 3) MNK_CASE(3, 9, 5, __VA_ARGS__)       \
    else if ((g).n == 3 && (g).k == 3 && (g).NW == 1) MNK_CASE(1, 3, 3, __VA_ARGS__)  \
    else if ((g).NW <= 2) MNK_CASE(2, 0, 0, __VA_ARGS__)                              \
    else if ((g).NW <= 4) MNK_CASE(4, 0, 0, __VA_ARGS__)                              \
    else MNK_CASE(8, 0, 0, __VA_ARGS__)                                               \
  } while (0)


// may the one-lane rollout address its record stores with 32-bit lane offsets (SADDR form)?  Only while a wave is
// alone on its SIMD (where it measured faster) and a launch's record rows stay below 4 GiB
inline bool mnk_rollout_saddr_ok(const MnkGeom& g, int64_t N, int T) {
  if (mnk_config().saddr_off) return false;  // MNK_ROLLOUT_SADDR=0 switches the form off (A/B timing)
  return N <= 65536 && ((int64_t)T * g.NW + 1) * N * 8 < (1ll << 32);
}

// is `act` a log format this board can use?  (0 = no log)
inline bool mnk_act_format_ok(int act, int C) {
  return act == 0 || act == MNK_ACT_U16 || (act == MNK_ACT_U8 && C <= 256) || (act == MNK_ACT_BITS7 && C <= 128) ||
         (act == MNK_ACT_U8P1 && C > 256 && C <= 512);  // (9 bits per action; only the boards that need it have kernel variants)
}

// the draw as a launch of its own (mnk_sample.hip)
int mnk_launch_sample(const MnkSample& sa, int64_t N, int C, hipStream_t s);

// one-lane rollout variants that write the action log (mnk_rollout_log.hip); act_bytes is 1 or 2
void mnk_launch_rollout_log(const MnkGeom& g, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed,
                            uint64_t step0, int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, int64_t* stats,
                            void* act_log, int act_bytes, void* stream);

// two-lanes-per-env rollout variants (mnk_rollout_pair.hip); geometry must be one of the compile-time boards
void mnk_launch_rollout_pair(const MnkGeom& g, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed,
                             uint64_t step0, int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, int64_t* stats,
                             void* act_log, int act_bytes, void* stream);

// waves-per-env-group rollout variants (mnk_rollout_ws.hip); ws = 2 or 4
bool mnk_rollout_ws_supported(const MnkGeom& g, int act_bytes);
void mnk_launch_rollout_ws(const MnkGeom& g, int ws, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed,
                           uint64_t step0, int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, int64_t* stats,
                           void* act_log, int act_bytes, void* stream);

// The rollout / replay kernels of boards with more than 16 register words per plane (planes of more than 512 bits) exist
// as run-time specialisations only: their generic ahead-of-time forms took 20 minutes to compile for kernels nobody's
// default board runs.  The API-level kernels keep a generic 32-word variant.
#define MNK_DISPATCH16(g, ...)                                                        \
  do {                                                                                \
    if ((g).n == 9 && (g).k == 5 && (g).NW == 3) MNK_CASE(3, 9, 5, __VA_ARGS__)       \
    else if ((g).n == 3 && (g).k == 3 && (g).NW == 1) MNK_CASE(1, 3, 3, __VA_ARGS__)  \
    else if ((g).n == 13 && (g).k == 5 && (g).NW == 6) MNK_CASE(6, 13, 5, __VA_ARGS__) \
    else if ((g).n == 15 && (g).k == 5 && (g).NW == 8) MNK_CASE(8, 15, 5, __VA_ARGS__) \
    else if ((g).n == 19 && (g).k == 5 && (g).NW == 12) MNK_CASE(12, 19, 5, __VA_ARGS__) \
    else if ((g).NW <= 2) MNK_CASE(2, 0, 0, __VA_ARGS__)                              \
    else if ((g).NW <= 4) MNK_CASE(4, 0, 0, __VA_ARGS__)                              \
    else if ((g).NW <= 8) MNK_CASE(8, 0, 0, __VA_ARGS__)                              \
    else MNK_CASE(16, 0, 0, __VA_ARGS__)                                              \
  } while (0)
#define MNK_DISPATCH16_LARGE(g, ...)                                                     \
  do {                                                                                   \
    if ((g).n == 19 && (g).k == 5 && (g).NW == 12) MNK_CASE(12, 19, 5, __VA_ARGS__)      \
    else MNK_CASE(16, 0, 0, __VA_ARGS__)                                                 \
  } while (0)
hipFunction_t mnk_jit_replay_function(const MnkGeom& g, bool rec, int act);
hipFunction_t mnk_jit_rollout_pair_function(const MnkGeom& g, bool rec, int act);
int mnk_jit_launch_rollout_lanes(hipFunction_t fn, MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed,
                                 uint64_t step0, int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, int64_t* stats,
                                 void* act_log, void* stream, int lanes_per_env);
int mnk_jit_launch_replay(hipFunction_t fn, MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, int T, const void* act_log,
                          uint64_t* rec_planes, uint32_t* rec_meta, int32_t* err, void* stream);

// run-time specialised rollout kernels (mnk_jit.hip, hiprtc): nullptr when the compile failed
hipFunction_t mnk_jit_rollout_function(const MnkGeom& g, bool rec, int act, bool saddr);
int mnk_jit_launch_rollout(hipFunction_t fn, MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed,
                           uint64_t step0, int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, int64_t* stats,
                           void* act_log, void* stream);

// two lanes per env with the board split by words (mnk_rollout_pairw.hip): 19x19x5 and 15x15x5
bool mnk_rollout_pairw_supported(const MnkGeom& g);
void mnk_launch_rollout_pairw(const MnkGeom& g, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed,
                              uint64_t step0, int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, int64_t* stats,
                              void* act_log, int act_bytes, void* stream);

// dynamic LDS a launch may ask for without raising the function's limit (the packed write-out stage of a run-time
// specialised kernel is larger than the table form's: launches that would not fit stay on the ahead-of-time kernels)
#define MNK_MAX_DYNAMIC_LDS ((size_t)64 * 1024 - 256)  /* (the kernels also hold a few static words) */

// ------------------------------------------------------------------ run-time specialised API-level kernels (mnk_jit.hip)
// The kernels of mnk_api_kernels.h / mnk_selfplay_kernels.h, compiled by hiprtc with the board's NW / n / k (and, for the
// forms with a folded-in draw, its cell count) as template arguments.  Boards with a built-in variant never get here.
enum MnkJitApiKind {  // (the public names: MNK_JIT_API_* of include/mnk_hip.h)
  MNK_JK_STEP = MNK_JIT_API_STEP,                      // k_step_full<NW, CN, CK, false>
  MNK_JK_STEP_DRAW = MNK_JIT_API_STEP_DRAW,            // k_step_full<NW, CN, CK, true>
  MNK_JK_STEP_SUBSET = MNK_JIT_API_STEP_SUBSET,        // k_step_subset
  MNK_JK_OBSERVE = MNK_JIT_API_OBSERVE,                // k_observe          (never looks at k: compiled with CK = 0)
  MNK_JK_SAMPLE_LEGAL = MNK_JIT_API_SAMPLE_LEGAL,      // k_sample_legal     (CK = 0)
  MNK_JK_UNPACK_RECORDS = MNK_JIT_API_UNPACK_RECORDS,  // k_unpack_records   (CK = 0)
  MNK_JK_GATHER_OBS = MNK_JIT_API_GATHER_OBS,          // k_gather_obs       (CK = 0)
  MNK_JK_SP_PRE = MNK_JIT_API_SP_PRE,                  // k_selfplay_pre / _post / _step_random <NW, CN, CK, NoDraw>
  MNK_JK_SP_POST = MNK_JIT_API_SP_POST,
  MNK_JK_SP_STEP = MNK_JIT_API_SP_STEP,
  MNK_JK_SP_DRAW = MNK_JIT_API_SP_DRAW,  // + 3 * lt + which: <NW, CN, CK, Draw<LT, C>>, lt 0 f32 / 1 bf16 / 2 no logits
  MNK_JK_COUNT = MNK_JIT_API_COUNT
};
inline bool mnk_jit_kind_any_k(int kind) { return kind >= MNK_JK_OBSERVE && kind <= MNK_JK_GATHER_OBS; }

// The board's own variant of API kernel `kind`, or nullptr = launch the ahead-of-time kernel: the board has a built-in
// variant, MNK_JIT_API / MNK_JIT = 0, the kernel is not hot yet (fewer than 1 024 launches and 2^26 items on this board in
// this process; MNK_JIT_API / MNK_JIT = 1: compile at the first launch), `stream` is being captured and the variant does
// not exist yet (nothing is compiled or loaded under a capture), or the compilation failed (mnk_jit_last_error).
#define MNK_JIT_HOT_LAUNCHES 1024u
#define MNK_JIT_HOT_ITEMS (1ull << 26)
hipFunction_t mnk_jit_api_function(const MnkGeom& g, int kind, int64_t items, hipStream_t stream);

template <typename... P, size_t... I>
inline hipError_t mnk_module_launch_impl(hipFunction_t fn, dim3 grid, dim3 block, size_t lds, hipStream_t s,
                                         std::tuple<P...>& params, std::index_sequence<I...>) {
  void* ptrs[] = {(void*)&std::get<I>(params)...};
  return hipModuleLaunchKernel(fn, grid.x, grid.y, grid.z, block.x, block.y, block.z, (unsigned)lds, s, ptrs, nullptr);
}

// Launches a run-time compiled kernel.  `signature`: any ahead-of-time instantiation of the same kernel template -- its
// parameter list is the module function's, so every argument is converted to the exact parameter type here and a
// mismatch in the number of arguments does not compile.
template <typename... P, typename... A>
inline void mnk_module_launch(void (*signature)(P...), hipFunction_t fn, dim3 grid, dim3 block, size_t lds, hipStream_t s,
                              A&&... a) {
  (void)signature;
  static_assert(sizeof...(P) == sizeof...(A), "argument list differs from the kernel's parameter list");
  std::tuple<P...> params{static_cast<P>(a)...};
  (void)mnk_module_launch_impl(fn, grid, block, lds, s, params, std::index_sequence_for<P...>{});  // (mnk_launch_status reads the error)
}

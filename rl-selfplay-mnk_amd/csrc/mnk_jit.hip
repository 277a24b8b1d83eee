// mnk_jit.hip -- run-time specialisation of the fused rollout kernel with hiprtc (gfx950 / MI355X only).
//
// The rollout kernel is bound by its own instruction count, and a third to a half of those
// instructions depend on the board: shift amounts of the win scan, word counts, the divisions by n and n+1.  For the
// boards compiled in ahead of time (3x3x3, 9x9x5, 13x13x5, 15x15x5, 19x19x5) they are immediates; every other (m, n, k)
// used to run kernels with run-time shift amounts and wave-uniform loops at a third to a fifth of that rate.
// Here the first large launch on such a board compiles `mnk_rollout_lane.h` -- the very text hipcc compiles for the
// built-in boards, embedded in this library at build time -- with the board's NW / n / k as template arguments, and
// every later launch runs that code object.  One compile per (board, record?, log width) and process, a few seconds.
// The reference (env/torch_vector_mnk_env.py:7-32) takes any m, n, k; this keeps every board on the fast path.
#include <hip/hiprtc.h>

#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "mnk_host.h"
#include "_obj/mnk_jit_sources.inc"  // MNK_JIT_HEADER_COUNT, mnk_jit_header_names[], mnk_jit_header_texts[]

namespace {

struct Compiled {
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;
  size_t code_bytes = 0;
  bool failed = false;
};

std::mutex g_mu;
// (device, m, n, k, record, log format, saddr, kind: 0 rollout / 1 replay)
std::map<std::tuple<int, int, int, int, int, int, int, int>, Compiled> g_cache;
thread_local char g_jit_err[2048] = "";

// compiles the rollout (kind 0) or the replay kernel (kind 1) for this geometry; code object bytes in `code` (no GPU
// needed for this part)
bool compile(const MnkGeom& g, bool rec, int act, bool saddr, std::vector<char>& code, int kind = 0) {
  static const char* program =
      "#include \"mnk_rollout_lane.h\"\n";
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, program, "mnk_jit_rollout.hip", MNK_JIT_HEADER_COUNT, mnk_jit_header_texts,
                          mnk_jit_header_names) != HIPRTC_SUCCESS) {
    snprintf(g_jit_err, sizeof(g_jit_err), "hiprtcCreateProgram failed");
    return false;
  }
  const std::string d_nw = "-DMNK_JIT_NW=" + std::to_string(g.NW), d_cn = "-DMNK_JIT_CN=" + std::to_string(g.n),
                    d_ck = "-DMNK_JIT_CK=" + std::to_string(g.k), d_rec = "-DMNK_JIT_REC=" + std::to_string(rec ? 1 : 0),
                    d_act = "-DMNK_JIT_ACT=" + std::to_string(act), d_sa = "-DMNK_JIT_SADDR=" + std::to_string(saddr ? 1 : 0),
                    d_kind = "-DMNK_JIT_KIND=" + std::to_string(kind);
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", d_nw.c_str(), d_cn.c_str(),
                        d_ck.c_str(), d_rec.c_str(), d_act.c_str(), d_sa.c_str(), d_kind.c_str()};
  const hiprtcResult rc = hiprtcCompileProgram(prog, (int)(sizeof(opts) / sizeof(opts[0])), opts);
  if (rc != HIPRTC_SUCCESS) {
    size_t n = 0;
    hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) hiprtcGetProgramLog(prog, &log[0]);
    snprintf(g_jit_err, sizeof(g_jit_err), "hiprtc: %s\n%.1800s", hiprtcGetErrorString(rc), log.c_str());
    hiprtcDestroyProgram(&prog);
    return false;
  }
  size_t n = 0;
  hiprtcGetCodeSize(prog, &n);
  code.resize(n);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  return n > 0;
}

}  // namespace

// kernel of this geometry, compiled on first use; nullptr (and mnk_jit_last_error) when that failed -- the caller
// then stays on the ahead-of-time generic kernel
static hipFunction_t jit_function(const MnkGeom& g, bool rec, int act, bool saddr, int kind) {
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return nullptr;  // a code object is loaded into one device's context
  std::lock_guard<std::mutex> lock(g_mu);
  Compiled& c = g_cache[std::make_tuple(device, g.m, g.n, g.k, rec ? 1 : 0, act, saddr ? 1 : 0, kind)];
  if (c.fn || c.failed) return c.fn;
  std::vector<char> code;
  if (!compile(g, rec, act, saddr, code, kind)) { c.failed = true; return nullptr; }
  if (hipModuleLoadData(&c.module, code.data()) != hipSuccess ||
      hipModuleGetFunction(&c.fn, c.module, kind == 1 ? "mnk_jit_replay" : (kind == 2 ? "mnk_jit_rollout_pair" : "mnk_jit_rollout")) != hipSuccess) {
    snprintf(g_jit_err, sizeof(g_jit_err), "hipModuleLoadData / hipModuleGetFunction failed: %s",
             hipGetErrorString(hipGetLastError()));
    c.failed = true;
    c.fn = nullptr;
    return nullptr;
  }
  c.code_bytes = code.size();
  return c.fn;
}

hipFunction_t mnk_jit_rollout_function(const MnkGeom& g, bool rec, int act, bool saddr) {
  return jit_function(g, rec, act, saddr, 0);
}

// the two-lanes-per-env form of the rollout (batches of up to 32 768 envs; byte / 16-bit logs or none)
hipFunction_t mnk_jit_rollout_pair_function(const MnkGeom& g, bool rec, int act) { return jit_function(g, rec, act, false, 2); }

// the replay of an action log in format `act` (boards of more than 16 register words have no ahead-of-time variant)
hipFunction_t mnk_jit_replay_function(const MnkGeom& g, bool rec, int act) { return jit_function(g, rec, act, false, 1); }

int mnk_jit_launch_replay(hipFunction_t fn, MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, int T, const void* act_log,
                          uint64_t* rec_planes, uint32_t* rec_meta, int32_t* err, void* stream) {
  void* args[] = {&g, &planes, &meta, &N, &T, &act_log, &rec_planes, &rec_meta, &err};
  const unsigned grid = (unsigned)((N + 63) / 64);
  if (hipModuleLaunchKernel(fn, grid, 1, 1, 64, 1, 1, 0, (hipStream_t)stream, args, nullptr) != hipSuccess)
    return mnk_launch_status("replay_actions (run-time specialised)");
  return MNK_OK;
}

int mnk_jit_launch_rollout(hipFunction_t fn, MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed,
                           uint64_t step0, int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, int64_t* stats,
                           void* act_log, void* stream) {
  return mnk_jit_launch_rollout_lanes(fn, g, planes, meta, N, T, seed, step0, env_id0, rec_planes, rec_meta, stats, act_log,
                                      stream, 1);
}

// lanes_per_env: 1 (64 envs per workgroup) or 2 (the pair form: 32 envs per workgroup)
int mnk_jit_launch_rollout_lanes(hipFunction_t fn, MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed,
                                 uint64_t step0, int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, int64_t* stats,
                                 void* act_log, void* stream, int lanes_per_env) {
  void* args[] = {&g, &planes, &meta, &N, &T, &seed, &step0, &env_id0, &rec_planes, &rec_meta, &stats, &act_log};
  const int per_group = 64 / lanes_per_env;
  const unsigned grid = (unsigned)((N + per_group - 1) / per_group);
  if (hipModuleLaunchKernel(fn, grid, 1, 1, 64, 1, 1, 0, (hipStream_t)stream, args, nullptr) != hipSuccess)
    return mnk_launch_status("rollout_random (run-time specialised)");
  return MNK_OK;
}

extern "C" {

const char* mnk_jit_last_error(void) { return g_jit_err; }

// Compiles (does not load) the rollout kernel of a geometry: code object size in bytes, or a negative MNK_E* code.
// Needs no GPU -- the build check and the CPU test suite use it.
int64_t mnk_jit_compile_rollout(int m, int n, int k, int record, int act_bytes) {
  MnkGeom g;
  const int rc = mnk_check_geom(m, n, k, &g);
  if (rc != MNK_OK) return rc;
  if (!mnk_act_format_ok(act_bytes, g.C)) return MNK_EINVAL;
  std::vector<char> code;
  if (!compile(g, record != 0, act_bytes, record != 0, code)) return MNK_ELAUNCH;  // the form a 65 536-env launch uses
  return (int64_t)code.size();
}

// the same for any of the run-time specialised kernels: kind MNK_JIT_ROLLOUT (one lane per env), MNK_JIT_REPLAY
// (mnk_replay_actions; act_bytes = the log format it reads) or MNK_JIT_ROLLOUT_PAIR (two lanes per env)
int64_t mnk_jit_compile_kernel(int m, int n, int k, int record, int act_bytes, int kind) {
  MnkGeom g;
  const int rc = mnk_check_geom(m, n, k, &g);
  if (rc != MNK_OK) return rc;
  if (kind < MNK_JIT_ROLLOUT || kind > MNK_JIT_ROLLOUT_PAIR || !mnk_act_format_ok(act_bytes, g.C)) return MNK_EINVAL;
  if (kind == MNK_JIT_REPLAY && act_bytes == 0) return MNK_EINVAL;
  if (kind == MNK_JIT_ROLLOUT_PAIR && (act_bytes == MNK_ACT_BITS7 || act_bytes == MNK_ACT_U8P1)) return MNK_EINVAL;
  std::vector<char> code;
  if (!compile(g, record != 0, act_bytes, kind == MNK_JIT_ROLLOUT && record != 0, code, kind)) return MNK_ELAUNCH;
  return (int64_t)code.size();
}

}  // extern "C"

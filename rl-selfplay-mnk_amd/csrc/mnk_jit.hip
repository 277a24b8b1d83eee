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
#include <errno.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
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

// ---- code objects on disk -------------------------------------------------------------------------------------------
// A board's first hot kernel waits 0.3-1.6 s for hiprtc, in every process.  Compiled code objects are therefore kept in
// $MNK_JIT_CACHE (default $XDG_CACHE_HOME/mnk_hip or ~/.cache/mnk_hip; "0" / "off" / no home directory: no cache), one
// file per (embedded sources, hiprtc version, program, options, kernel name), named by the FNV-1a hash of all of those:
// another build of the library or another ROCm never reads this build's files.  A file that is short, carries another key
// or fails its checksum is ignored and rewritten; files appear by rename, so a reader never sees half of one.
std::atomic<int64_t> g_stat_compiled{0}, g_stat_hits{0}, g_stat_stores{0}, g_stat_failed{0};

uint64_t fnv1a(uint64_t h, const void* p, size_t n) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001B3ull; }
  return h;
}
uint64_t fnv1a(uint64_t h, const std::string& s) { return fnv1a(fnv1a(h, s.data(), s.size()), "\0", 1); }

uint64_t sources_hash() {
  static const uint64_t h = [] {
    uint64_t x = 0xCBF29CE484222325ull;
    int major = 0, minor = 0;
    hiprtcVersion(&major, &minor);
    const int v[3] = {MNK_ABI_VERSION, major, minor};
    x = fnv1a(x, v, sizeof(v));
    for (int i = 0; i < MNK_JIT_HEADER_COUNT; ++i) {
      x = fnv1a(x, std::string(mnk_jit_header_names[i]));
      x = fnv1a(x, std::string(mnk_jit_header_texts[i]));
    }
    return x;
  }();
  return h;
}

std::string cache_dir() {
  if (const char* v = getenv("MNK_JIT_CACHE")) {
    if (!*v || !strcmp(v, "0") || !strcmp(v, "off")) return "";
    return v;
  }
  if (const char* x = getenv("XDG_CACHE_HOME")) if (*x) return std::string(x) + "/mnk_hip";
  if (const char* h = getenv("HOME")) if (*h) return std::string(h) + "/.cache/mnk_hip";
  return "";
}

bool make_dirs(const std::string& dir) {
  for (size_t i = 1; i <= dir.size(); ++i)
    if (i == dir.size() || dir[i] == '/') {
      const std::string part = dir.substr(0, i);
      if (mkdir(part.c_str(), 0700) != 0 && errno != EEXIST) return false;
    }
  return true;
}

struct CacheHeader {
  char magic[8];  // "MNKJIT1"
  uint64_t key, check;
  uint32_t name_len, reserved;
  uint64_t code_len;
};

std::string cache_path(const std::string& dir, uint64_t key) {
  char name[40];
  snprintf(name, sizeof(name), "/%016llx.co", (unsigned long long)key);
  return dir + name;
}

bool cache_load(uint64_t key, std::vector<char>& code, std::string& lowered) {
  const std::string dir = cache_dir();
  if (dir.empty()) return false;
  FILE* f = fopen(cache_path(dir, key).c_str(), "rb");
  if (!f) return false;
  CacheHeader h;
  bool ok = fread(&h, sizeof(h), 1, f) == 1 && !memcmp(h.magic, "MNKJIT1", 8) && h.key == key && h.name_len < 4096 &&
            h.code_len > 0 && h.code_len < (1ull << 30);
  if (ok) {
    lowered.assign(h.name_len, '\0');
    code.resize(h.code_len);
    ok = (h.name_len == 0 || fread(&lowered[0], h.name_len, 1, f) == 1) && fread(code.data(), h.code_len, 1, f) == 1 &&
         fgetc(f) == EOF;
    if (ok) ok = fnv1a(fnv1a(0xCBF29CE484222325ull, lowered.data(), lowered.size()), code.data(), code.size()) == h.check;
  }
  fclose(f);
  if (ok) g_stat_hits++;
  return ok;
}

void cache_store(uint64_t key, const std::vector<char>& code, const std::string& lowered) {
  const std::string dir = cache_dir();
  if (dir.empty() || code.empty() || !make_dirs(dir)) return;
  CacheHeader h;
  memset(&h, 0, sizeof(h));
  memcpy(h.magic, "MNKJIT1", 8);
  h.key = key;
  h.name_len = (uint32_t)lowered.size();
  h.code_len = code.size();
  h.check = fnv1a(fnv1a(0xCBF29CE484222325ull, lowered.data(), lowered.size()), code.data(), code.size());
  const std::string path = cache_path(dir, key);
  char tmp[64];
  snprintf(tmp, sizeof(tmp), ".tmp.%ld.%llx", (long)getpid(), (unsigned long long)key);
  const std::string tpath = dir + "/" + tmp;
  FILE* f = fopen(tpath.c_str(), "wb");
  if (!f) return;
  const bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && (lowered.empty() || fwrite(lowered.data(), lowered.size(), 1, f) == 1) &&
                  fwrite(code.data(), code.size(), 1, f) == 1;
  if (fclose(f) != 0 || !ok || rename(tpath.c_str(), path.c_str()) != 0) { unlink(tpath.c_str()); return; }
  g_stat_stores++;
}

// key of one compilation: the embedded sources + everything handed to hiprtc for it
uint64_t compile_key(const char* program, const char* const* opts, int nopts, const std::string& expr) {
  uint64_t k = fnv1a(sources_hash(), std::string(program));
  for (int i = 0; i < nopts; ++i) k = fnv1a(k, std::string(opts[i]));
  return fnv1a(k, expr);
}

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
  const int nopts = (int)(sizeof(opts) / sizeof(opts[0]));
  const uint64_t key = compile_key(program, opts, nopts, "");
  std::string no_name;
  if (cache_load(key, code, no_name)) { hiprtcDestroyProgram(&prog); return true; }
  const hiprtcResult rc = hiprtcCompileProgram(prog, nopts, opts);
  if (rc != HIPRTC_SUCCESS) {
    g_stat_failed++;
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
  if (n > 0) {
    g_stat_compiled++;
    cache_store(key, code, no_name);
  }
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

// ------------------------------------------------------------------ API-level kernels (round 4)
// The kernels behind mnk_step / mnk_observe / mnk_sample_legal / mnk_unpack_records / mnk_gather_obs / mnk_selfplay_* of
// a board without a built-in variant ran generic code: wave-uniform loops over run-time shift amounts and the table form
// of the write-out, 1.3-1.4x slower than a compiled board of the same size.  Here hiprtc instantiates the same templates
// (mnk_api_kernels.h, mnk_selfplay_kernels.h: the text hipcc compiles, embedded at build time) with the board's NW / n / k
// -- one kernel per program, named by a hiprtc name expression -- which also puts the board on the packed write-out
// (mnk_emit.h) and lets the self-play step kernels fold the masked draw in for ANY row width (mnk_draw::Shape).
namespace {

struct ApiEntry {
  Compiled c;
  uint64_t launches = 0, items = 0;
};
// (device, m, n, k or 0, kind)
std::map<std::tuple<int, int, int, int, int>, ApiEntry> g_api_cache;

// the template-id of API kernel `kind` on this board, e.g. "k_step_full<5, 12, 5, false>"
std::string api_kernel_name(const MnkGeom& g, int kind) {
  const int ck = mnk_jit_kind_any_k(kind) ? 0 : g.k;
  const std::string geo = std::to_string(g.NW) + ", " + std::to_string(g.n) + ", " + std::to_string(ck);
  switch (kind) {
    case MNK_JK_STEP: return "k_step_full<" + geo + ", false>";
    case MNK_JK_STEP_DRAW: return "k_step_full<" + geo + ", true>";
    case MNK_JK_STEP_SUBSET: return "k_step_subset<" + geo + ">";
    case MNK_JK_OBSERVE: return "k_observe<" + geo + ">";
    case MNK_JK_SAMPLE_LEGAL: return "k_sample_legal<" + geo + ">";
    case MNK_JK_UNPACK_RECORDS: return "k_unpack_records<" + geo + ">";
    case MNK_JK_GATHER_OBS: return "k_gather_obs<" + geo + ">";
    default: break;
  }
  static const char* which[] = {"k_selfplay_pre", "k_selfplay_post", "k_selfplay_step_random"};
  if (kind >= MNK_JK_SP_PRE && kind <= MNK_JK_SP_STEP) return std::string(which[kind - MNK_JK_SP_PRE]) + "<" + geo + ", NoDraw>";
  static const char* lts[] = {"float", "uint16_t", "void"};
  const int d = kind - MNK_JK_SP_DRAW;
  return std::string(which[d % 3]) + "<" + geo + ", Draw<" + lts[d / 3] + ", " + std::to_string(g.C) + "> >";
}

// compiles API kernel `kind` for this geometry (no GPU needed): code object in `code`, mangled name in `lowered`
bool compile_api(const MnkGeom& g, int kind, std::vector<char>& code, std::string& lowered) {
  static const char* program = "#include \"mnk_selfplay_kernels.h\"\n";  // (includes mnk_api_kernels.h)
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, program, "mnk_jit_api.hip", MNK_JIT_HEADER_COUNT, mnk_jit_header_texts,
                          mnk_jit_header_names) != HIPRTC_SUCCESS) {
    snprintf(g_jit_err, sizeof(g_jit_err), "hiprtcCreateProgram failed");
    return false;
  }
  const std::string expr = "&" + api_kernel_name(g, kind);
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"};
  const int nopts = (int)(sizeof(opts) / sizeof(opts[0]));
  const uint64_t key = compile_key(program, opts, nopts, expr);
  if (cache_load(key, code, lowered)) { hiprtcDestroyProgram(&prog); return true; }
  bool ok = hiprtcAddNameExpression(prog, expr.c_str()) == HIPRTC_SUCCESS;
  hiprtcResult rc = HIPRTC_SUCCESS;
  if (ok) rc = hiprtcCompileProgram(prog, nopts, opts);
  if (!ok || rc != HIPRTC_SUCCESS) {
    g_stat_failed++;
    size_t n = 0;
    hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) hiprtcGetProgramLog(prog, &log[0]);
    snprintf(g_jit_err, sizeof(g_jit_err), "hiprtc (%s): %s\n%.1700s", expr.c_str(), ok ? hiprtcGetErrorString(rc) : "name expression rejected",
             log.c_str());
    hiprtcDestroyProgram(&prog);
    return false;
  }
  const char* name = nullptr;
  if (hiprtcGetLoweredName(prog, expr.c_str(), &name) != HIPRTC_SUCCESS || !name) {
    snprintf(g_jit_err, sizeof(g_jit_err), "hiprtcGetLoweredName(%s) failed", expr.c_str());
    hiprtcDestroyProgram(&prog);
    return false;
  }
  lowered = name;
  size_t n = 0;
  hiprtcGetCodeSize(prog, &n);
  code.resize(n);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  if (n > 0) {
    g_stat_compiled++;
    cache_store(key, code, lowered);
  }
  return n > 0;
}

bool api_kind_ok(const MnkGeom& g, int kind) {
  if (kind < 0 || kind >= MNK_JK_COUNT) return false;
  return kind < MNK_JK_SP_DRAW || g.C <= 1024;
}

// compile + load on the current device (g_mu held)
hipFunction_t api_build(ApiEntry& e, const MnkGeom& g, int kind) {
  std::vector<char> code;
  std::string lowered;
  if (!compile_api(g, kind, code, lowered)) { e.c.failed = true; return nullptr; }
  if (hipModuleLoadData(&e.c.module, code.data()) != hipSuccess ||
      hipModuleGetFunction(&e.c.fn, e.c.module, lowered.c_str()) != hipSuccess) {
    snprintf(g_jit_err, sizeof(g_jit_err), "hipModuleLoadData / hipModuleGetFunction(%s) failed: %s", lowered.c_str(),
             hipGetErrorString(hipGetLastError()));
    e.c.failed = true;
    e.c.fn = nullptr;
    return nullptr;
  }
  e.c.code_bytes = code.size();
  return e.c.fn;
}

}  // namespace

hipFunction_t mnk_jit_api_function(const MnkGeom& g, int kind, int64_t items, hipStream_t stream) {
  const int jit = mnk_config().jit_api;
  if (jit == 0 || mnk_geom_builtin(g.n, g.k, g.NW) || !api_kind_ok(g, kind)) return nullptr;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_mu);
  ApiEntry& e = g_api_cache[std::make_tuple(device, g.m, g.n, mnk_jit_kind_any_k(kind) ? 0 : g.k, kind)];
  if (e.c.fn || e.c.failed) return e.c.fn;
  e.launches += 1;
  e.items += (uint64_t)(items > 0 ? items : 0);
  // not hot yet?  (A compilation costs 0.3-0.9 s and buys ~5 us per launch: it pays for itself after ~10^5 launches -- any
  // training run, no smoke test.  1 024 launches into a run is early enough to lose nothing and late enough to spare
  // short scripts the wait.)
  if (jit != 1 && e.launches < MNK_JIT_HOT_LAUNCHES && e.items < MNK_JIT_HOT_ITEMS) return nullptr;
  if (stream) {  // nothing is compiled or loaded while the stream is being captured into a graph
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
  }
  return api_build(e, g, kind);
}

extern "C" {

// Compiles (does not load) API kernel `kind` (MnkJitApiKind, mnk_host.h) of a geometry: code object size in bytes or a
// negative MNK_E* code.  Needs no GPU: the build check and the CPU test suite use it.
int64_t mnk_jit_compile_api(int m, int n, int k, int kind) {
  MnkGeom g;
  const int rc = mnk_check_geom(m, n, k, &g);
  if (rc != MNK_OK) return rc;
  if (!api_kind_ok(g, kind)) return MNK_EINVAL;
  std::vector<char> code;
  std::string lowered;
  if (!compile_api(g, kind, code, lowered)) return MNK_ELAUNCH;
  return (int64_t)code.size();
}

// Compiles and loads, on the current device, the board's own variants of the API kernels named by the bits of
// `kinds` (bit MnkJitApiKind) now instead of when they get hot -- before a stream capture, where nothing can be compiled.
// kinds == 0: every kernel that has been launched on this board so far (on this device, in this process): a warm-up run
// followed by mnk_jit_prepare(m, n, k, 0) prepares exactly what the capture is going to launch.  Returns the number of
// variants ready, 0 for a board with a built-in variant or with MNK_JIT_API / MNK_JIT = 0, or a negative MNK_E* code
// (mnk_jit_last_error says which kernel failed).
int mnk_jit_prepare(int m, int n, int k, int64_t kinds) {
  MnkGeom g;
  const int rc = mnk_check_geom(m, n, k, &g);
  if (rc != MNK_OK) return rc;
  if (mnk_config().jit_api == 0 || mnk_geom_builtin(g.n, g.k, g.NW)) return 0;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return MNK_ELAUNCH;
  std::lock_guard<std::mutex> lock(g_mu);
  int ready = 0;
  for (int kind = 0; kind < MNK_JK_COUNT; ++kind) {
    if (!api_kind_ok(g, kind)) continue;
    const auto key = std::make_tuple(device, g.m, g.n, mnk_jit_kind_any_k(kind) ? 0 : g.k, kind);
    if (kinds == 0) {
      auto it = g_api_cache.find(key);
      if (it == g_api_cache.end() || (it->second.launches == 0 && !it->second.c.fn)) continue;
    } else if (!((kinds >> kind) & 1)) {
      continue;
    }
    ApiEntry& e = g_api_cache[key];
    if (!e.c.fn && !e.c.failed) api_build(e, g, kind);
    if (!e.c.fn) return MNK_ELAUNCH;
    ++ready;
  }
  return ready;
}

// is the board's own variant of API kernel `kind` loaded on the current device?  (1 / 0; what a launch would use)
int mnk_jit_api_ready(int m, int n, int k, int kind) {
  MnkGeom g;
  if (mnk_check_geom(m, n, k, &g) != MNK_OK || !api_kind_ok(g, kind)) return 0;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return 0;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_api_cache.find(std::make_tuple(device, g.m, g.n, mnk_jit_kind_any_k(kind) ? 0 : g.k, kind));
  return it != g_api_cache.end() && it->second.c.fn ? 1 : 0;
}

// what the run-time compiler has done in this process: out[0] programs compiled by hiprtc, out[1] code objects taken from
// the cache on disk instead ($MNK_JIT_CACHE), out[2] written to it, out[3] failed compilations
int mnk_jit_stats(int64_t* out4) {
  if (!out4) return MNK_EINVAL;
  out4[0] = g_stat_compiled; out4[1] = g_stat_hits; out4[2] = g_stat_stores; out4[3] = g_stat_failed;
  return MNK_OK;
}

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

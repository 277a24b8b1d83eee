// mnk_device.h -- device-side building blocks shared by the MNK kernels (gfx950 only).
//
// One lane owns one env.  A board plane is WT u64 words in registers (guard-column
// layout, see include/mnk_hip.h); the K-in-a-row test of the reference
// (env/torch_vector_mnk_env.py:106-119: three conv2d's with ones / eye stencils over
// the mover's whole plane, "> k - 0.1") becomes four shift-AND chains on that bit
// string -- shifts 1 (row), n+1 (column), n+2 (diagonal), n (anti-diagonal) -- which
// is exact integer arithmetic, so it reproduces the f32 sums of 0/1 values bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MNK_MAX_W 8

struct MnkGeom {
  int m, n, k;
  int C;        // m*n cells = number of actions
  int W;        // u64 words per plane in memory
  int stride;   // n+1 bits per board row (guard column included)
  uint32_t magic_n;       // x / n      == __umulhi(x, magic_n)      for x*n      < 2^32
  uint32_t magic_stride;  // x / (n+1)
  uint32_t magic_C;       // x / C
  uint32_t magic_2C;      // x / (2C)
  uint64_t valid[MNK_MAX_W];  // 1 on real cells, 0 on guard / padding bits
};

__device__ __forceinline__ uint32_t mnk_div(uint32_t x, uint32_t magic) { return __umulhi(x, magic); }

// ---------------------------------------------------------------- multi-word bit strings
template <int WT>
__device__ __forceinline__ void bs_shr(uint64_t (&x)[WT], int s) {
  // s is wave-uniform (depends on the geometry only)
  while (s > 63) {
#pragma unroll
    for (int w = 0; w < WT; ++w) x[w] = (x[w] >> 63) | (w + 1 < WT ? (x[w + 1] << 1) : 0ull);
    s -= 63;
  }
  if (s == 0) return;
#pragma unroll
  for (int w = 0; w < WT; ++w) x[w] = (x[w] >> s) | (w + 1 < WT ? (x[w + 1] << (64 - s)) : 0ull);
}

// does the bit string hold a run of >= k set bits spaced d apart?
template <int WT>
__device__ __forceinline__ bool bs_has_run(const uint64_t (&b)[WT], int d, int k) {
  uint64_t x[WT], t[WT];
#pragma unroll
  for (int w = 0; w < WT; ++w) x[w] = b[w];
  int len = 1;  // x marks the starts of runs of >= len
  while (2 * len <= k) {
#pragma unroll
    for (int w = 0; w < WT; ++w) t[w] = x[w];
    bs_shr<WT>(t, len * d);
#pragma unroll
    for (int w = 0; w < WT; ++w) x[w] &= t[w];
    len *= 2;
  }
  if (len < k) {
#pragma unroll
    for (int w = 0; w < WT; ++w) t[w] = x[w];
    bs_shr<WT>(t, (k - len) * d);
#pragma unroll
    for (int w = 0; w < WT; ++w) x[w] &= t[w];
  }
  uint64_t any = 0;
#pragma unroll
  for (int w = 0; w < WT; ++w) any |= x[w];
  return any != 0;
}

// env/torch_vector_mnk_env.py:106-119 on one plane
template <int WT>
__device__ __forceinline__ bool mnk_plane_wins(const MnkGeom& g, const uint64_t (&b)[WT]) {
  bool hit = bs_has_run<WT>(b, 1, g.k);
  hit |= bs_has_run<WT>(b, g.stride, g.k);
  hit |= bs_has_run<WT>(b, g.stride + 1, g.k);
  hit |= bs_has_run<WT>(b, g.n, g.k);
  return hit;
}

// position of the r-th (0-based) set bit of v; r < popcount(v)
__device__ __forceinline__ int select_bit64(uint64_t v, int r) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  int c = __popc(lo);
  int pos = 0;
  uint32_t x = lo;
  if (r >= c) { r -= c; x = hi; pos = 32; }
  c = __popc(x & 0xFFFFu);
  if (r >= c) { r -= c; x >>= 16; pos += 16; }
  c = __popc(x & 0xFFu);
  if (r >= c) { r -= c; x >>= 8; pos += 8; }
  c = __popc(x & 0xFu);
  if (r >= c) { r -= c; x >>= 4; pos += 4; }
  c = __popc(x & 0x3u);
  if (r >= c) { r -= c; x >>= 2; pos += 2; }
  if (r >= (int)(x & 1u)) pos += 1;
  return pos;
}

template <int WT>
__device__ __forceinline__ int bs_popcount(const uint64_t (&x)[WT]) {
  int c = 0;
#pragma unroll
  for (int w = 0; w < WT; ++w) c += __popcll(x[w]);
  return c;
}

// bit index of the r-th set bit of the multi-word string; r < popcount
template <int WT>
__device__ __forceinline__ int bs_select(const uint64_t (&x)[WT], int r) {
  uint64_t word = x[0];
  int base = 0;
  bool found = false;
#pragma unroll
  for (int w = 0; w < WT; ++w) {
    int c = __popcll(x[w]);
    bool here = !found && r < c;
    if (here) { word = x[w]; base = 64 * w; found = true; }
    if (!found) r -= c;
  }
  return base + select_bit64(word, r);
}

// ---------------------------------------------------------------- Philox4x32-10
struct Philox4 { uint32_t v[4]; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// counter layout of oracle/philox.py: ctr = (env_lo, env_hi, q_lo, stream | q_hi24 << 8)
__device__ __forceinline__ Philox4 mnk_rng_block(uint64_t seed, uint64_t env, uint64_t q, uint32_t stream) {
  return philox4x32_10((uint32_t)env, (uint32_t)(env >> 32), (uint32_t)q,
                       stream | ((uint32_t)((q >> 32) & 0xFFFFFFu) << 8),
                       (uint32_t)seed, (uint32_t)(seed >> 32));
}

__device__ __forceinline__ uint32_t philox_word(const Philox4& b, uint32_t i) {
  uint32_t lo = (i & 1u) ? b.v[1] : b.v[0];
  uint32_t hi = (i & 1u) ? b.v[3] : b.v[2];
  return (i & 2u) ? hi : lo;
}

// scalar streams: one u32 per (env, step)
__device__ __forceinline__ uint32_t mnk_rand_u32(uint64_t seed, uint64_t env, uint64_t step, uint32_t stream) {
  Philox4 b = mnk_rng_block(seed, env, step >> 2, stream);
  return philox_word(b, (uint32_t)(step & 3));
}

// ---------------------------------------------------------------- one env in registers
template <int WT>
struct MnkEnv {
  uint64_t p[2][WT];
  uint32_t meta;  // bit0 side to move, bits 1.. move count
};

template <int WT>
__device__ __forceinline__ void env_load(MnkEnv<WT>& e, const uint64_t* planes, const uint32_t* meta, int64_t N,
                                         int W, int64_t i) {
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int w = 0; w < WT; ++w) e.p[pl][w] = (w < W) ? planes[((int64_t)(pl * W + w)) * N + i] : 0ull;
  e.meta = meta[i];
}

template <int WT>
__device__ __forceinline__ void env_store(const MnkEnv<WT>& e, uint64_t* planes, uint32_t* meta, int64_t N, int W,
                                          int64_t i) {
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int w = 0; w < WT; ++w)
      if (w < W) planes[((int64_t)(pl * W + w)) * N + i] = e.p[pl][w];
  meta[i] = e.meta;
}

template <int WT>
__device__ __forceinline__ void env_clear(MnkEnv<WT>& e) {
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int w = 0; w < WT; ++w) e.p[pl][w] = 0ull;
  e.meta = 0u;
}

template <int WT>
__device__ __forceinline__ void env_legal(const MnkGeom& g, const MnkEnv<WT>& e, uint64_t (&legal)[WT]) {
#pragma unroll
  for (int w = 0; w < WT; ++w) legal[w] = ~(e.p[0][w] | e.p[1][w]) & g.valid[w];
}

// uniform legal cell from one u32 (oracle/philox.py pick_legal; selfplay/policy.py:18-29)
template <int WT>
__device__ __forceinline__ int env_pick_legal(const MnkGeom& g, const MnkEnv<WT>& e, uint32_t x) {
  uint64_t legal[WT];
  env_legal<WT>(g, e, legal);
  int nl = bs_popcount<WT>(legal);
  if (nl == 0) return (int)__umulhi(x, (uint32_t)g.C);
  int r = (int)__umulhi(x, (uint32_t)nl);
  int bit = bs_select<WT>(legal, r);
  return bit - (int)mnk_div((uint32_t)bit, g.magic_stride);
}

struct MnkPly {
  bool win, done;
  int err;  // MNK_ERR_* (0 = fine); on error the env is left untouched
};

// env/torch_vector_mnk_env.py:60-84 for one env.
template <int WT>
__device__ __forceinline__ MnkPly env_play(const MnkGeom& g, MnkEnv<WT>& e, int64_t action, bool strict) {
  MnkPly out;
  out.win = false; out.done = false; out.err = 0;
  const int C = g.C;
  int64_t a64 = action < 0 ? action + C : action;  // torch indexing wraps negatives (:68)
  if (a64 < 0 || a64 >= C) { out.err = 1; return out; }
  const uint32_t a = (uint32_t)a64;
  const uint32_t bit = a + mnk_div(a, g.magic_n);  // row*(n+1) + col
  const int wsel = (int)(bit >> 6);
  const uint64_t one = 1ull << (bit & 63u);
  const uint32_t side = e.meta & 1u;
  if (strict) {
    uint64_t occ = 0;
#pragma unroll
    for (int w = 0; w < WT; ++w) occ |= (w == wsel) ? ((e.p[0][w] | e.p[1][w]) & one) : 0ull;
    if (occ) { out.err = 2; return out; }
  }
  uint64_t mine[WT];
#pragma unroll
  for (int w = 0; w < WT; ++w) {
    const uint64_t add = (w == wsel) ? one : 0ull;
    e.p[0][w] |= side ? 0ull : add;   // :68 boards[idx, player, r, c] = 1
    e.p[1][w] |= side ? add : 0ull;
    mine[w] = side ? e.p[1][w] : e.p[0][w];
  }
  const uint32_t moves = (e.meta >> 1) + 1u;           // :69
  out.win = mnk_plane_wins<WT>(g, mine);                // :71
  const bool draw = (moves >= (uint32_t)C) && !out.win;  // :72
  out.done = out.win || draw;                           // :73
  e.meta = (moves << 1) | (side ^ 1u);                  // :82 toggles even when finished
  return out;
}

__device__ __forceinline__ void mnk_report(int32_t* err, int code, int64_t env) {
  if (err && atomicCAS(&err[0], 0, code) == 0) err[1] = (int32_t)env;
}

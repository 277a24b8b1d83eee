// mnk_device.h -- device-side building blocks shared by the MNK kernels (gfx950 only).
//
// One lane owns one env.  A board plane is a bit string in the guard-column layout of
// include/mnk_hip.h, held in registers as NW 32-bit words (9x9: 90 bits = 3 VGPRs per
// plane; in memory it stays u64[W], the unused top half-word is simply never loaded).
// The K-in-a-row test of the reference (env/torch_vector_mnk_env.py:106-119: three
// conv2d's with ones / eye stencils over the mover's whole plane, "> k - 0.1") becomes
// four shift-AND chains on that bit string -- shifts 1 (row), n+1 (column), n+2
// (diagonal), n (anti-diagonal) -- exact integer arithmetic, so it reproduces the f32
// sums of 0/1 values bit for bit.  Multi-word shifts are v_alignbit_b32 per word.
//
// Templates: NW = register words per plane; CN / CK = board width / run length when
// they are compile-time constants (0 = read them from MnkGeom at run time).  The
// specialised forms turn every shift amount into an immediate and unroll the run
// doubling; the generic forms keep wave-uniform loops.
#pragma once
#ifdef __HIPCC_RTC__
// compiled at run time by hiprtc (mnk_jit.hip): the HIP device API is built in, libc headers do not exist
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef unsigned int uint32_t;
typedef unsigned long long uint64_t;
typedef signed char int8_t;
typedef int int32_t;
typedef long long int64_t;
typedef unsigned long long uintptr_t;
#include "mnk_hip.h"
#else
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mnk_hip.h"
#endif

#define MNK_MAX_W 16    // u64 words per plane in memory: boards of up to 1 024 bits m*(n+1) (25x25, 31x31)
#define MNK_MAX_NW 32   // u32 words per plane in registers

struct MnkGeom {
  int m, n, k;
  int C;        // m*n cells = number of actions
  int W;        // u64 words per plane in memory
  int NW;       // u32 words that actually hold board bits: ceil(m*(n+1)/32)
  int stride;   // n+1 bits per board row (guard column included)
  uint32_t magic_n;       // x / n      == __umulhi(x, magic_n)      for x*n      < 2^32
  uint32_t magic_stride;  // x / (n+1)
  uint32_t magic_C;       // x / C
  uint32_t magic_2C;      // x / (2C)
  uint32_t valid[MNK_MAX_NW];  // 1 on real cells, 0 on guard / padding bits
};

__device__ __forceinline__ uint32_t mnk_div(uint32_t x, uint32_t magic) { return __umulhi(x, magic); }

__host__ __device__ inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// A masked draw from a policy head's logits (mnk_sample_logits; the mnk_selfplay_*_logits entry points fold it into a
// step kernel): where the logits and the mask live, the sampler's Philox key and position, where the results go.
struct MnkSample {
  const void* logits;        // [N][C] f32 / bf16 bit patterns; NULL = all-zero logits (uniform over the mask)
  int logits_dtype;          // MNK_LOGITS_F32 / MNK_LOGITS_BF16
  const uint8_t* mask;       // [N][C]
  uint64_t seed;
  const uint64_t* seed_dev;  // optional device word that REPLACES seed (a captured graph's sampler can be re-keyed)
  uint64_t step;
  const uint64_t* step_dev;  // optional device word ADDED to step
  int64_t env_id0;           // Philox row id of row 0
  int deterministic;
  int64_t* actions;          // out [N]
  float* logp;               // out [N], optional
};

template <int CN>
__device__ __forceinline__ int geom_n(const MnkGeom& g) { return CN ? CN : g.n; }
template <int CK>
__device__ __forceinline__ int geom_k(const MnkGeom& g) { return CK ? CK : g.k; }

// ---------------------------------------------------------------- multi-word bit strings
// x >>= s for a wave-uniform s >= 0 (an immediate in the specialised kernels)
template <int NW>
__device__ __forceinline__ void bs_shr(uint32_t (&x)[NW], int s) {
  for (int q = s >> 5; q > 0; --q) {
#pragma unroll
    for (int w = 0; w < NW; ++w) x[w] = (w + 1 < NW) ? x[w + 1] : 0u;
  }
  const int r = s & 31;
  if (r) {
#pragma unroll
    for (int w = 0; w < NW; ++w) x[w] = __builtin_amdgcn_alignbit((w + 1 < NW) ? x[w + 1] : 0u, x[w], (uint32_t)r);
  }
}

// does the bit string hold k set bits spaced d apart?  log2(k) doubling steps.
template <int NW>
__device__ __forceinline__ bool bs_has_run(const uint32_t (&b)[NW], int d, int k) {
  uint32_t x[NW], t[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) x[w] = b[w];
  int len = 1;  // x marks the starts of runs of >= len
  while (2 * len <= k) {
#pragma unroll
    for (int w = 0; w < NW; ++w) t[w] = x[w];
    bs_shr<NW>(t, len * d);
#pragma unroll
    for (int w = 0; w < NW; ++w) x[w] &= t[w];
    len *= 2;
  }
  if (len + 1 == k) {
    // one stone short (k = 3, 5, 9, ...): AND with the string itself, shifted by len*d.  Same result as the
    // general step below, but the large shift moves zeros into the top words, so the compiler drops
    // everything that only fed those words (9x9x5: 12 instead of 15 ops for each of the three long strides)
#pragma unroll
    for (int w = 0; w < NW; ++w) t[w] = b[w];
    bs_shr<NW>(t, len * d);
#pragma unroll
    for (int w = 0; w < NW; ++w) x[w] &= t[w];
  } else if (len < k) {
#pragma unroll
    for (int w = 0; w < NW; ++w) t[w] = x[w];
    bs_shr<NW>(t, (k - len) * d);
#pragma unroll
    for (int w = 0; w < NW; ++w) x[w] &= t[w];
  }
  uint32_t any = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) any |= x[w];
  return any != 0;
}

// env/torch_vector_mnk_env.py:106-119 on one plane
template <int NW, int CN, int CK>
__device__ __forceinline__ bool mnk_plane_wins(const MnkGeom& g, const uint32_t (&b)[NW]) {
  const int n = geom_n<CN>(g), k = geom_k<CK>(g);
  bool hit = bs_has_run<NW>(b, 1, k);      // rows
  hit |= bs_has_run<NW>(b, n + 1, k);      // columns
  hit |= bs_has_run<NW>(b, n + 2, k);      // diagonals
  hit |= bs_has_run<NW>(b, n, k);          // anti-diagonals
  return hit;
}

// position of the r-th (0-based) set bit of x; r < popcount(x).
// A halving search in arithmetic only, as one block of gfx950 assembly: on a wave that is alone on its
// SIMD every instruction -- VALU, SALU or hazard s_nop -- costs one ~4-cycle issue slot
// (tools/exp_valu_rate.hip), and the v_cmp -> SGPR -> v_cndmask form the compiler makes of any C++
// spelling of this costs 8-9 slots per level.  Here a level is five:
//   t    = popcount(x[pos .. pos+S)) + nr      nr = -(r + 1), negative while bits remain
//   nr   = umax(nr, t)                         t < 0  <=>  the bit lies above the field: rank -= count
//   pos |= S & (t >> 31)                       ... and the position moves up
// The last level looks at one bit: the answer is pos + 1 unless r == 0 and that bit is set.
#define MNK_SELECT_LEVEL(S)                        \
  "v_bfe_u32 %[t], %[x], %[pos], " #S "\n\t"       \
  "v_bcnt_u32_b32 %[t], %[t], %[nr]\n\t"          \
  "v_max_u32 %[nr], %[nr], %[t]\n\t"              \
  "v_ashrrev_i32 %[t], 31, %[t]\n\t"              \
  "v_and_or_b32 %[pos], %[t], " #S ", %[pos]\n\t"
__device__ __forceinline__ int select_bit32(uint32_t x, int r) {
  uint32_t nr = ~(uint32_t)r, pos, t;
  asm("v_and_b32 %[t], 0xffff, %[x]\n\t"
      "v_bcnt_u32_b32 %[t], %[t], %[nr]\n\t"
      "v_max_u32 %[nr], %[nr], %[t]\n\t"
      "v_ashrrev_i32 %[t], 31, %[t]\n\t"
      "v_and_b32 %[pos], 16, %[t]\n\t"
      MNK_SELECT_LEVEL(8) MNK_SELECT_LEVEL(4) MNK_SELECT_LEVEL(2)
      "v_bfe_u32 %[t], %[x], %[pos], 1\n\t"
      "v_bitop3_b32 %[t], %[nr], %[t], %[t] bitop3:0x3f\n\t"  // ~(nr & bit): bit 0 clear only if r == 0 and the bit is set
      "v_and_or_b32 %[pos], %[t], 1, %[pos]"
      : [pos] "=&v"(pos), [t] "=&v"(t), [nr] "+v"(nr)
      : [x] "v"(x));
  return (int)pos;
}
#undef MNK_SELECT_LEVEL

template <int NW>
__device__ __forceinline__ int bs_popcount(const uint32_t (&x)[NW]) {
  int c = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) c += __popc(x[w]);
  return c;
}

// bit index of the r-th set bit of the multi-word string (r < popcount), and the string with only
// that bit set in `hot`.  The word is the last one whose prefix count is <= r: prefix counts never
// decrease, so the per-word tests are independent of each other.
template <int NW>
__device__ __forceinline__ int bs_select_hot(const uint32_t (&x)[NW], int r_, uint32_t (&hot)[NW]) {
  const uint32_t r = (uint32_t)r_;
  uint32_t word = x[0], base = 0, before = 0, pre = 0;
  bool past[NW + 1];
  past[0] = true;
  past[NW] = false;
#pragma unroll
  for (int w = 1; w < NW; ++w) {
    pre += (uint32_t)__popc(x[w - 1]);
    past[w] = r >= pre;
    word = past[w] ? x[w] : word;
    base = past[w] ? 32u * w : base;
    before = past[w] ? pre : before;
  }
  const uint32_t pos = (uint32_t)select_bit32(word, (int)(r - before));
  const uint32_t one = 1u << pos;
  // past[] is monotone (past[w + 1] implies past[w]): with a[w] = past[w] ? one : 0 the chosen word is where a[] drops,
  // hot[w] = a[w] ^ a[w + 1] -- one select and one xor per word boundary instead of two selects
  uint32_t a[NW + 1];
  a[0] = one;
  a[NW] = 0u;
#pragma unroll
  for (int w = 1; w < NW; ++w) a[w] = past[w] ? one : 0u;
#pragma unroll
  for (int w = 0; w < NW; ++w) hot[w] = a[w] ^ a[w + 1];
  return (int)(base + pos);
}

template <int NW>
__device__ __forceinline__ int bs_select(const uint32_t (&x)[NW], int r) {
  uint32_t hot[NW];
  return bs_select_hot<NW>(x, r, hot);
}

// ---------------------------------------------------------------- Philox4x32-10
struct Philox4 { uint32_t v[4]; };

// 32 x 32 -> 64 in one instruction (the compiler spells it v_mul_lo_u32 + v_mul_hi_u32: two issue slots)
__device__ __forceinline__ uint64_t mnk_mul_wide(uint32_t a, uint32_t m) {
  uint64_t product, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(product), "=s"(carry) : "v"(a), "s"(m));
  return product;
}

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = mnk_mul_wide(c0, M0), p1 = mnk_mul_wide(c2, M1);
    // three-input xor in one slot (v_bitop3_b32, truth table 0x96); left alone the compiler emits two v_xor_b32
    const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
    const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
    c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
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
template <int NW>
struct MnkEnv {
  uint32_t p[2][NW];
  uint32_t meta;  // bit0 side to move, bits 1.. move count
};

// planes u64[2][W][N] in memory <-> NW u32 words per plane in registers
// EXACT: the plane has exactly (NW+1)/2 memory words (true for the specialised boards), so the
// wave-uniform "does this word exist" tests fold away
template <int NW, bool EXACT = false>
__device__ __forceinline__ void plane_load(uint32_t (&x)[NW], const uint64_t* plane, int64_t N, int W, int64_t i) {
#pragma unroll
  for (int q = 0; q < (NW + 1) / 2; ++q) {
    const uint64_t v = (EXACT || q < W) ? plane[(int64_t)q * N + i] : 0ull;
    x[2 * q] = (uint32_t)v;
    if (2 * q + 1 < NW) x[2 * q + 1] = (uint32_t)(v >> 32);
  }
}

template <int NW, bool EXACT = false>
__device__ __forceinline__ void plane_store(const uint32_t (&x)[NW], uint64_t* plane, int64_t N, int W, int64_t i) {
#pragma unroll
  for (int q = 0; q < (NW + 1) / 2; ++q) {
    const uint64_t hi = (2 * q + 1 < NW) ? (uint64_t)x[2 * q + 1] : 0ull;
    if (EXACT || q < W) plane[(int64_t)q * N + i] = (uint64_t)x[2 * q] | (hi << 32);
  }
}

// ---------------------------------------------------------------- rollout records
// One recorded position is NW u64 rows of stride N: row w = first plane's word w | second plane's word w << 32
// (the 32-bit words of two planes, interleaved; the rollout records the mover's plane first).  No padding at any board size -- 9x9 takes 3 rows =
// 24 B where the state layout's two u64 planes take 32 -- and the rollout is bound by exactly these
// stores.  nw = the board's word count (MnkGeom::NW), used when NW is only an upper bound.
template <int NW, bool EXACT = false>
__device__ __forceinline__ void rec_store(const uint32_t (&p0)[NW], const uint32_t (&p1)[NW], uint64_t* rows, int64_t N,
                                          int nw, int64_t i) {
#pragma unroll
  for (int w = 0; w < NW; ++w)
    if (EXACT || w < nw) rows[(int64_t)w * N + i] = (uint64_t)p0[w] | ((uint64_t)p1[w] << 32);
}

template <int NW, bool EXACT = false>
__device__ __forceinline__ void rec_load(uint32_t (&p0)[NW], uint32_t (&p1)[NW], const uint64_t* rows, int64_t N, int nw,
                                         int64_t i) {
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const uint64_t v = (EXACT || w < nw) ? rows[(int64_t)w * N + i] : 0ull;
    p0[w] = (uint32_t)v;
    p1[w] = (uint32_t)(v >> 32);
  }
}

template <int NW, bool EXACT = false>
__device__ __forceinline__ void env_load(MnkEnv<NW>& e, const uint64_t* planes, const uint32_t* meta, int64_t N,
                                         int W, int64_t i) {
  plane_load<NW, EXACT>(e.p[0], planes, N, W, i);
  plane_load<NW, EXACT>(e.p[1], planes + (int64_t)W * N, N, W, i);
  e.meta = meta[i];
}

template <int NW, bool EXACT = false>
__device__ __forceinline__ void env_store(const MnkEnv<NW>& e, uint64_t* planes, uint32_t* meta, int64_t N, int W,
                                          int64_t i) {
  plane_store<NW, EXACT>(e.p[0], planes, N, W, i);
  plane_store<NW, EXACT>(e.p[1], planes + (int64_t)W * N, N, W, i);
  meta[i] = e.meta;
}

template <int NW>
__device__ __forceinline__ void env_clear(MnkEnv<NW>& e) {
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int w = 0; w < NW; ++w) e.p[pl][w] = 0u;
  e.meta = 0u;
}

template <int NW>
__device__ __forceinline__ void env_legal(const MnkGeom& g, const MnkEnv<NW>& e, uint32_t (&legal)[NW]) {
#pragma unroll
  for (int w = 0; w < NW; ++w) legal[w] = ~(e.p[0][w] | e.p[1][w]) & g.valid[w];
}

// uniform legal cell from one u32 (oracle/philox.py pick_legal; selfplay/policy.py:18-29)
template <int NW, int CN>
__device__ __forceinline__ int env_pick_legal(const MnkGeom& g, const MnkEnv<NW>& e, uint32_t x) {
  uint32_t legal[NW];
  env_legal<NW>(g, e, legal);
  const int nl = bs_popcount<NW>(legal);
  // a full board (nl == 0, poked states only) draws over all C cells like RandomPolicy's 1e-8 guard
  // (policy.py:21-24): the select then runs over the valid-cell string, whose r-th set bit is cell r
#pragma unroll
  for (int w = 0; w < NW; ++w) legal[w] = nl ? legal[w] : g.valid[w];
  const int r = (int)__umulhi(x, (uint32_t)(nl ? nl : g.C));
  const uint32_t bit = (uint32_t)bs_select<NW>(legal, r);
  return (int)(bit - (CN ? bit / (uint32_t)(CN + 1) : mnk_div(bit, g.magic_stride)));
}

struct MnkPly {
  bool win, done;
  int err;  // MNK_ERR_* (0 = fine); on error the env is left untouched
};

// env/torch_vector_mnk_env.py:60-84 for one env.
// TRUSTED: the action comes from our own legal-move sampler (always in [0, C)), so the range
// check -- a divergent branch around the whole ply -- is compiled out.
template <int NW, int CN, int CK, bool TRUSTED = false>
__device__ __forceinline__ MnkPly env_play(const MnkGeom& g, MnkEnv<NW>& e, int64_t action, bool strict) {
  MnkPly out;
  out.win = false; out.done = false; out.err = 0;
  const int C = g.C;
  uint32_t a;
  if (TRUSTED) {
    a = (uint32_t)action;
  } else {
    const int64_t a64 = action < 0 ? action + C : action;  // torch indexing wraps negatives (:68)
    if (a64 < 0 || a64 >= C) { out.err = 1; return out; }
    a = (uint32_t)a64;
  }
  const uint32_t bit = a + (CN ? a / (uint32_t)CN : mnk_div(a, g.magic_n));  // row*(n+1) + col
  const int wsel = (int)(bit >> 5);
  const uint32_t one = 1u << (bit & 31u);
  const uint32_t side = e.meta & 1u;
  if (strict) {
    uint32_t occ = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) occ |= (w == wsel) ? ((e.p[0][w] | e.p[1][w]) & one) : 0u;
    if (occ) { out.err = 2; return out; }
  }
  uint32_t mine[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const uint32_t add = (w == wsel) ? one : 0u;
    e.p[0][w] |= side ? 0u : add;   // :68 boards[idx, player, r, c] = 1
    e.p[1][w] |= side ? add : 0u;
    mine[w] = side ? e.p[1][w] : e.p[0][w];
  }
  const uint32_t moves = (e.meta >> 1) + 1u;                 // :69
  out.win = mnk_plane_wins<NW, CN, CK>(g, mine);              // :71
  const bool draw = (moves >= (uint32_t)C) && !out.win;       // :72
  out.done = out.win || draw;                                 // :73
  e.meta = (moves << 1) | (side ^ 1u);                        // :82 toggles even when finished
  return out;
}

__device__ __forceinline__ void mnk_report(int32_t* err, int code, int64_t env) {
  if (err && atomicCAS(&err[0], 0, code) == 0) err[1] = (int32_t)env;
}

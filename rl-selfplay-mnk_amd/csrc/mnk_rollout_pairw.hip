// mnk_rollout_pairw.hip -- the fused random rollout with a board SPLIT BY WORDS over two lanes (gfx950 / MI355X only).
// Its own translation unit so its variants compile in parallel with the other rollout kernels.
//
// For large boards at small batches (19x19x5 at 32 768 envs per GPU is BASELINE config 5).  There the one-lane
// kernel leaves half the SIMDs empty, and the two-lane form of mnk_rollout_pair.hip -- both lanes hold the WHOLE
// board and split the four scan directions -- executes 0.89x the one-lane instructions per lane: everything but the
// scan is done twice.  Here lane `role` of a pair holds only words [role*H, role*H + H) of both planes (H = ceil(NW / 2)),
// so legal string, popcounts, word pick, stone placement, record stores, side swap and reset all run on half the
// words, and the scan runs all four directions on half the words: a shift that reaches past a lane's last word
// takes the partner's low words with one DPP move each (the upper lane shifts zeros in).  Rank-select: each lane
// counts its own legal cells, one DPP swap gives both the pair's total, the lane whose interval holds the drawn
// rank selects inside its own words and hands the action to its partner.  ~0.6x the one-lane instructions per lane.
// Results are bit-identical to the one-lane kernel (parity tests run every form).
#include "mnk_host.h"
#include "mnk_rollout_lane.h"

namespace {

// partner lane's value (lane ^ 1): DPP quad_perm [1,0,3,2]
__device__ __forceinline__ uint32_t partner(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
}

template <int NW, int CN, int CK, bool RECORD, int ACT>
struct RolloutPairW {
  static_assert(NW >= 2 && CN != 0, "word-split pairs: compile-time boards of at least two words");
  static constexpr int H = (NW + 1) / 2;  // words per lane; the upper lane's last word is padding (always 0) when NW is odd
  const MnkGeom& g;
  int64_t N;
  uint32_t role;     // 0: words [0, H), 1: words [H, NW)
  uint32_t up_mask;  // role 0: all ones (the partner's words continue this lane's bit string), role 1: 0 (zeros above)
  uint32_t cur[H], oth[H], valid[H];
  uint32_t side, moves, moves_in = 0;
  // record stores address `wave-uniform row base + this lane's 32-bit byte offset` (global_store ... s[base]): one
  // 32-bit add per ply instead of a 64-bit pointer per row; the launcher keeps a launch's record rows below 4 GiB
  const char* rbase[H] = {};  // rec_planes[.][j][0]
  const char* mbase = nullptr;
  uint32_t roff = 0, moff = 0;  // byte offsets of rec_planes[t][role*H][i] (relative to row 0) / rec_meta[t][i]
  uint8_t* ra = nullptr;   // act_log[t / 4][i]
  uint64_t quad = 0;
  uint8_t* rh = nullptr;   // ACT 4 (MNK_ACT_U8P1): the bit plane of bit 8 behind the byte words, see mnk_rollout_lane.h
  uint32_t hi4 = 0, hiw = 0, hfill = 0;
  uint32_t acc_done = 0, acc_win = 0, acc_white = 0;

  __device__ __forceinline__ RolloutPairW(const MnkGeom& g_, int64_t N_, int64_t i, uint32_t role_, uint64_t* rec_planes,
                                          uint32_t* rec_meta, void* act_log)
      : g(g_), N(N_), role(role_), up_mask(role_ ? 0u : ~0u) {
    if (RECORD) {
#pragma unroll
      for (int j = 0; j < H; ++j) rbase[j] = (const char*)(rec_planes + (int64_t)j * N);
      mbase = (const char*)rec_meta;
      roff = ((uint32_t)i + role * (uint32_t)H * (uint32_t)N) * 8u;
      moff = (uint32_t)i * 4u;
    }
    if (ACT) ra = (uint8_t*)act_log + i * 4 * (ACT == 2 ? 2 : 1);
#pragma unroll
    for (int j = 0; j < H; ++j) valid[j] = role ? g.valid[H + j] : g.valid[j];  // g.valid is 0 past the board's last word
  }

  __device__ __forceinline__ void log_begin(void* act_log, int T, int64_t i) {
    if constexpr (ACT == 4) rh = (uint8_t*)act_log + ((int64_t)((T + 3) >> 2) * N + i) * 4;
  }

  // 32-bit word `gw` of a plane in memory (u64[W][N]); 0 past the board's last word
  __device__ __forceinline__ uint32_t load_word(const uint64_t* plane, uint32_t gw, int64_t i) const {
    const uint32_t q = (gw < (uint32_t)NW ? gw : 0u) >> 1;
    const uint64_t v = plane[(int64_t)q * N + i];
    const uint32_t half = (gw & 1u) ? (uint32_t)(v >> 32) : (uint32_t)v;
    return gw < (uint32_t)NW ? half : 0u;
  }

  // Memory holds u64 words; a lane holds the 32-bit words [role*H, role*H + H).  Load: the u64 word each 32-bit
  // word lives in (once per launch: a word shared by both lanes is simply read twice).
  __device__ __forceinline__ void load(const uint64_t* planes, const uint32_t* meta, int64_t i) {
    constexpr int W = (NW + 1) / 2;
    const uint32_t mw = meta[i];
    side = mw & 1u;
    moves = moves_in = mw >> 1;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const uint32_t gw = role * H + j;
      const uint32_t p0 = load_word(planes, gw, i), p1 = load_word(planes + (int64_t)W * N, gw, i);
      cur[j] = side ? p1 : p0;
      oth[j] = side ? p0 : p1;
    }
  }

  // Store: the lane that holds a u64 word's low half writes it; the high half is its own next word, or -- when the
  // u64 word straddles the two lanes (H odd) -- the partner's first word.
  __device__ __forceinline__ void store(uint64_t* planes, uint32_t* meta, int64_t i) const {
    constexpr int W = (NW + 1) / 2;
    uint32_t b[H], w[H];
#pragma unroll
    for (int j = 0; j < H; ++j) {
      b[j] = side ? oth[j] : cur[j];
      w[j] = side ? cur[j] : oth[j];
    }
    const uint32_t b_up = partner(b[0]) & up_mask, w_up = partner(w[0]) & up_mask;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const uint32_t gw = role * H + j;
      if ((gw & 1u) == 0u && gw < (uint32_t)NW) {
        const int nx = (j + 1 < H) ? j + 1 : 0;  // (index kept in range for the last word, where *_up is used)
        const uint32_t bh = (j + 1 < H) ? b[nx] : b_up;
        const uint32_t wh = (j + 1 < H) ? w[nx] : w_up;
        planes[(int64_t)(gw >> 1) * N + i] = (uint64_t)b[j] | ((uint64_t)bh << 32);
        planes[(int64_t)(W + (gw >> 1)) * N + i] = (uint64_t)w[j] | ((uint64_t)wh << 32);
      }
    }
    if (role == 0) meta[i] = (moves << 1) | side;
  }

  // out = the pair's NW-word string x, shifted right by S bits, this lane's words of it
  template <int S>
  __device__ __forceinline__ void shr(const uint32_t (&x)[H], uint32_t (&out)[H]) const {
    constexpr int Q = S >> 5, R = S & 31;
    // ext[j] = word role*H + j of the string, j < H + Q + 1: own words, then the partner's (zeros for the upper lane)
    uint32_t ext[H + Q + 1];
#pragma unroll
    for (int j = 0; j < H + Q + 1; ++j) {
      if (j < H) ext[j] = x[j];
      else if (j - H < H) ext[j] = partner(x[j - H]) & up_mask;
      else ext[j] = 0u;
    }
#pragma unroll
    for (int j = 0; j < H; ++j)
      out[j] = R ? __builtin_amdgcn_alignbit(ext[j + Q + 1], ext[j + Q], (uint32_t)R) : ext[j + Q];
  }

  // run-doubling scan of one direction (see bs_has_run): OR of this lane's words of the final string
  template <int D, int LEN = 1>
  __device__ __forceinline__ void run_steps(uint32_t (&x)[H]) const {
    uint32_t t[H];
    if constexpr (2 * LEN <= CK) {
      shr<LEN * D>(x, t);
#pragma unroll
      for (int j = 0; j < H; ++j) x[j] &= t[j];
      run_steps<D, 2 * LEN>(x);
    } else if constexpr (LEN + 1 == CK) {  // one stone short: AND with the plane itself shifted by LEN*D
      shr<LEN * D>(cur, t);
#pragma unroll
      for (int j = 0; j < H; ++j) x[j] &= t[j];
    } else if constexpr (LEN < CK) {
      shr<(CK - LEN) * D>(x, t);
#pragma unroll
      for (int j = 0; j < H; ++j) x[j] &= t[j];
    }
  }

  template <int D>
  __device__ __forceinline__ uint32_t run_bits() const {
    uint32_t x[H];
#pragma unroll
    for (int j = 0; j < H; ++j) x[j] = cur[j];
    run_steps<D>(x);
    uint32_t any = 0;
#pragma unroll
    for (int j = 0; j < H; ++j) any |= x[j];
    return any;
  }

  __device__ __forceinline__ void log_flush() {
    if constexpr (ACT == 4) {  // low bytes as one word, bit 8 of the four actions into the bit plane (both lanes keep the
      if (role == 0) *(uint32_t*)ra = (uint32_t)quad;  // cursors; the lower lane stores)
      ra += N * 4;
      quad = 0;
      hiw |= hi4 << hfill;
      hi4 = 0;
      hfill += 4u;
      if (hfill == 32u) {
        if (role == 0) *(uint32_t*)rh = hiw;
        rh += N * 4;
        hiw = 0;
        hfill = 0;
      }
      return;
    }
    if (role == 0) {
      if (ACT == 1) *(uint32_t*)ra = (uint32_t)quad;
      if (ACT == 2) *(uint64_t*)ra = quad;
    }
    ra += N * 4 * ACT;
    quad = 0;
  }

  __device__ __forceinline__ void log_finish(int T) {
    if (ACT && (T & 3)) log_flush();
    if constexpr (ACT == 4) {
      if (hfill && role == 0) *(uint32_t*)rh = hiw;
    }
  }

  // uniform legal cell from one u32 (oracle/philox.py pick_legal): the action (same in both lanes) and the cell's
  // bit as a one-hot string over this lane's words (all zero in the lane that does not hold it)
  // FAST: every game of the wave is consistent (stones == plies counted < C, mnk_rollout_lane.h): no full-board branch,
  // and `nlegal` (the pair's legal-cell count) stands in for the ply counter
  template <bool FAST = false>
  __device__ __forceinline__ int pick(uint32_t x, uint32_t (&hot)[H], uint32_t& nlegal) const {
    uint32_t legal[H];
#pragma unroll
    for (int j = 0; j < H; ++j) legal[j] = ~(cur[j] | oth[j]) & valid[j];
    uint32_t mine_n = (uint32_t)bs_popcount<H>(legal);
    uint32_t other_n = partner(mine_n);
    uint32_t n = mine_n + other_n;
    nlegal = n;
    if (!FAST && __builtin_amdgcn_ballot_w64(n == 0) != 0) {  // full board (poked states only): any cell, like RandomPolicy's 1e-8 guard
      const bool full = n == 0;
#pragma unroll
      for (int j = 0; j < H; ++j) legal[j] = full ? valid[j] : legal[j];
      mine_n = (uint32_t)bs_popcount<H>(legal);
      other_n = partner(mine_n);
      n = mine_n + other_n;
    }
    const uint32_t r = __umulhi(x, n);
    const uint32_t low_n = role ? other_n : mine_n;  // legal cells in the lower lane's words
    const bool holds = role ? (r >= low_n) : (r < low_n);
    const uint32_t local = holds ? (role ? r - low_n : r) : 0u;
    const uint32_t bit = (uint32_t)bs_select_hot<H>(legal, (int)local, hot) + role * 32u * H;
#pragma unroll
    for (int j = 0; j < H; ++j) hot[j] = holds ? hot[j] : 0u;
    const uint32_t a_here = bit - bit / (uint32_t)(CN + 1);
    const uint32_t a_there = partner(a_here);
    return (int)(holds ? a_here : a_there);
  }

  template <bool FAST = false>
  __device__ __forceinline__ void ply(uint32_t x, int field) {
    uint32_t hot[H];
    uint32_t nlegal;
    const int a = pick<FAST>(x, hot, nlegal);
    if constexpr (ACT == 4) {
      quad |= (uint64_t)((uint32_t)a & 0xFFu) << (8 * field);
      hi4 |= ((uint32_t)a >> 8) << field;
      if (field == 3) log_flush();
    } else if (ACT) {
      quad |= (uint64_t)(uint32_t)a << (8 * ACT * field);
      if (field == 3) log_flush();
    }
    if (RECORD) {  // this lane's rows of the position before the ply: mover's word | other side's word << 32
#pragma unroll
      for (int j = 0; j < H; ++j)
        if (2 * H == NW || j + 1 < H || role == 0)  // odd NW: the upper lane's last word is padding, not a row
          __builtin_nontemporal_store((uint64_t)cur[j] | ((uint64_t)oth[j] << 32), (uint64_t*)(rbase[j] + (uint64_t)roff));
      roff += (uint32_t)NW * (uint32_t)N * 8u;
    }
#pragma unroll
    for (int j = 0; j < H; ++j) cur[j] |= hot[j];                         // env:68
    if (!FAST) ++moves;                                                   // :69
    uint32_t hit = run_bits<1>() | run_bits<CN + 1>() | run_bits<CN + 2>() | run_bits<CN>();  // :71, this lane's words
    hit |= partner(hit);
    const uint32_t win = hit ? 1u : 0u;
    const uint32_t done = win | ((FAST ? nlegal == 1u : moves >= (uint32_t)g.C) ? 1u : 0u);  // :72-73
    if (RECORD) {
      if (role == 0)
        __builtin_nontemporal_store((uint32_t)a | (win << MNK_REC_REWARD_SHIFT) | (done << MNK_REC_DONE_BIT) | (side << MNK_REC_SIDE_BIT),
                                    (uint32_t*)(mbase + (uint64_t)moff));
      moff += (uint32_t)N * 4u;
    }
    acc_done += done;
    acc_win += win;
    acc_white += win & side;
#pragma unroll
    for (int j = 0; j < H; ++j) {  // the other side is to move (:82) -- or a fresh game
      const uint32_t c = cur[j];
      cur[j] = done ? 0u : oth[j];
      oth[j] = done ? 0u : c;
    }
    side = done ? 0u : (side ^ 1u);
    if (!FAST) moves = done ? 0u : moves;
  }

  // stones of the pair's board (both lanes get the sum)
  __device__ __forceinline__ uint32_t stones() const {
    uint32_t occ[H];
#pragma unroll
    for (int j = 0; j < H; ++j) occ[j] = cur[j] | oth[j];
    const uint32_t mine = (uint32_t)bs_popcount<H>(occ);
    return mine + partner(mine);
  }
  __device__ __forceinline__ bool consistent() const { return stones() == moves && moves < (uint32_t)g.C; }
  __device__ __forceinline__ void finish_fast() { moves = stones(); }
};

#ifndef MNK_PAIRW_FAST
#define MNK_PAIRW_FAST 1  // (0: build without the FAST loop, for A/B timing)
#endif

// the T plies of one lane pair
template <bool FAST, typename Lane>
__device__ __forceinline__ void pairw_plies(Lane& L, int T, uint64_t seed, uint64_t step0, uint64_t env, uint32_t role) {
  int t = 0;
  uint64_t step = step0;
  // unshared Philox until the step counter sits on a multiple of 8 (two blocks)
  for (; t < T && (step & 7); ++t, ++step)
    L.template ply<FAST>(mnk_rand_u32(seed, env, step, MNK_STREAM_MOVE), (int)(step & 3));
  for (; t + 8 <= T; t += 8, step += 8) {
    // lane `role` computes block (step/4 + role); the partner's four words arrive by DPP
    const Philox4 mine = mnk_rng_block(seed, env, (step >> 2) + role, MNK_STREAM_MOVE);
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t other = partner(mine.v[j]);
      lo[j] = role ? other : mine.v[j];  // block step/4
      hi[j] = role ? mine.v[j] : other;  // block step/4 + 1
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) L.template ply<FAST>(lo[j], j);
#pragma unroll
    for (int j = 0; j < 4; ++j) L.template ply<FAST>(hi[j], j);
  }
  for (; t < T; ++t, ++step) L.template ply<FAST>(mnk_rand_u32(seed, env, step, MNK_STREAM_MOVE), (int)(step & 3));
}

template <int NW, int CN, int CK, bool RECORD, int ACT>
__global__ void __launch_bounds__(64)
k_rollout_random_pairw(MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed, uint64_t step0,
                       int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, unsigned long long* stats,
                       void* act_log) {
  __shared__ unsigned int lds_stats[MNK_STATS_COUNTERS];
  if (threadIdx.x < MNK_STATS_COUNTERS) lds_stats[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t role = threadIdx.x & 1u;
  const int64_t i = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 1);  // env of this lane pair
  if (i < N) {
    RolloutPairW<NW, CN, CK, RECORD, ACT> L(g, N, i, role, rec_planes, rec_meta, act_log);
    L.log_begin(act_log, T, i);
    L.load(planes, meta, i);
    const uint64_t env = (uint64_t)(env_id0 + i);
    // every game of this wave consistent (any state the env produced itself): the loop without the full-board branch
    // and the ply counter (mnk_rollout_lane.h, FAST); a wave with a poked state plays the general loop, same results.
    // This form runs small batches, where the instruction count is the bound: it pays on every board it is built for.
    if (MNK_PAIRW_FAST && __builtin_amdgcn_ballot_w64(!L.consistent()) == 0) {
      pairw_plies<true>(L, T, seed, step0, env, role);
      L.finish_fast();
    } else {
      pairw_plies<false>(L, T, seed, step0, env, role);
    }
    L.log_finish(T);
    L.store(planes, meta, i);
    if (stats && role == 0) {
      const uint32_t len_sum = L.moves_in + (uint32_t)T - L.moves;
      if (L.acc_done) atomicAdd(&lds_stats[0], L.acc_done);
      if (L.acc_win - L.acc_white) atomicAdd(&lds_stats[1], L.acc_win - L.acc_white);
      if (L.acc_white) atomicAdd(&lds_stats[2], L.acc_white);
      if (L.acc_done - L.acc_win) atomicAdd(&lds_stats[3], L.acc_done - L.acc_win);
      if (len_sum) atomicAdd(&lds_stats[4], len_sum);
    }
  }
  __syncthreads();
  if (stats && threadIdx.x < MNK_STATS_COUNTERS && lds_stats[threadIdx.x])
    atomicAdd(&stats[(size_t)(blockIdx.x % MNK_STATS_REPLICAS) * MNK_STATS_STRIDE + threadIdx.x],
              (unsigned long long)lds_stats[threadIdx.x]);
}

}  // namespace

bool mnk_rollout_pairw_supported(const MnkGeom& g) {
  return (g.n == 19 && g.k == 5 && g.NW == 12) || (g.n == 15 && g.k == 5 && g.NW == 8) ||
         (g.n == 13 && g.k == 5 && g.NW == 6) || (g.n == 9 && g.k == 5 && g.NW == 3);
}

void mnk_launch_rollout_pairw(const MnkGeom& g, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed,
                              uint64_t step0, int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, int64_t* stats,
                              void* act_log, int act_bytes, void* stream) {
  const bool rec = rec_planes && rec_meta;
  const dim3 pgrid((unsigned)((N + 31) / 32));
#define MNK_PW(NWv, CNv, CKv, REC, ACTB)                                                                        \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rollout_random_pairw<NWv, CNv, CKv, REC, ACTB>), pgrid, dim3(64), 0,     \
                     (hipStream_t)stream, g, planes, meta, N, T, seed, step0, env_id0, rec_planes, rec_meta,    \
                     (unsigned long long*)stats, act_log)
#define MNK_PW_GEOM(REC, ACTB)                                \
  do {                                                        \
    if (g.n == 19) MNK_PW(12, 19, 5, REC, ACTB);              \
    else if (g.n == 15) MNK_PW(8, 15, 5, REC, ACTB);          \
    else if (g.n == 13) MNK_PW(6, 13, 5, REC, ACTB);          \
    else MNK_PW(3, 9, 5, REC, ACTB);                          \
  } while (0)
  // boards up to 256 cells log a byte per action; 19x19 = 361 needs two, or a byte and a bit (MNK_ACT_U8P1)
  if (act_bytes == MNK_ACT_U8P1) {
    if (rec) MNK_PW(12, 19, 5, true, 4);
    else MNK_PW(12, 19, 5, false, 4);
  } else if (rec && act_bytes == 1) MNK_PW_GEOM(true, 1);
  else if (rec && act_bytes == 2) MNK_PW_GEOM(true, 2);
  else if (rec) MNK_PW_GEOM(true, 0);
  else if (act_bytes == 1) MNK_PW_GEOM(false, 1);
  else if (act_bytes == 2) MNK_PW_GEOM(false, 2);
  else MNK_PW_GEOM(false, 0);
#undef MNK_PW_GEOM
#undef MNK_PW
}

// mnk_rollout_lane.h -- the fused random-policy rollout kernel template (gfx950 / MI355X only), shared by the
// translation units that instantiate it: mnk_rollout.hip (no action log; also the replay kernel),
// mnk_rollout_log.hip (the action-log variants) and mnk_rollout_pair.hip (two lanes per env).  Split so that
// the many instantiations compile in parallel.
// Device code only (no host headers): the same text is compiled ahead of time by hipcc for the built-in boards
// and at run time by hiprtc, with the board's geometry as template arguments, for every other board (mnk_jit.hip).
#pragma once
#include "mnk_device.h"
#include "mnk_pair_scan.h"

// ------------------------------------------------------------------ fused random rollout
// T plies per env in one launch; state lives in registers, HBM sees one load and one
// store of the state per launch plus the 28-byte (9x9) record of every ply.
// The loop is laid out for a wave that is alone on its SIMD (65 536 envs = 1024 waves =
// one per SIMD): no divergent branch, per-lane bookkeeping instead of per-ply ballots
// (scalar round trips), four plies per Philox block with the word picked at compile time.
// ACT = bytes per action of the optional action log (0 = none, 1 = u8, 2 = u16).  The log is stored
// four plies per word -- u32[ceil(T/4)][N] (ACT 1) or u64[ceil(T/4)][N] (ACT 2), action of ply 4q+j in
// field j of word [q][i] -- so a wave writes 256 / 512 contiguous bytes per store and the field position is a
// compile-time constant in the unrolled loop: +7 % kernel time.  (One byte store per lane per ply, and a packed
// word with a run-time field index, were both measured at +20 %.)
// ACT = 3 (MNK_ACT_BITS7, boards of at most 128 cells): the log is a stream of 7-bit actions, ply p at bit 7p, in u32
// words [ceil(7 * ceil(T/4) / 8)][N] -- 0.875 B per env-step on the wire.  Four plies still gather in `quad` at
// compile-time positions (28 bits, one 32-bit register); the quads then go through a 32-bit accumulator whose fill level
// is wave-uniform (every lane is at the same ply), so the shift amounts and the "first quad of a group of eight" branch
// are scalar: seven word stores per 32 plies, three 32-bit VALU operations per quad besides the store.  (A 64-bit
// accumulator -- the obvious spelling -- made every field shift a v_lshlrev_b64: 106.6 instead of 95 us per 256 plies.)
// One-lane form only.
// ACT = 4 (MNK_ACT_U8P1, boards of more than 256 cells): the low byte of every action exactly as ACT = 1, plus bit 8 of
// every action in a bit plane behind the byte words: four plies' bits gather at compile-time positions, the groups go
// into one word per 32 plies at a wave-uniform position -- one more VALU operation per ply and one more store per 32.
// PAIR: the two-lanes-per-env form for small batches (mnk_rollout_pair.hip): both lanes of a pair carry the env and
// pick the move redundantly; lane `role` scans two of the four directions (one DPP swap ORs the verdicts), writes
// half `role` of every record row and stores plane `role` of the final state.
// WS > 1: the waves-per-env-group form (mnk_rollout_ws.hip): a workgroup of WS waves carries the SAME 64 envs in
// every wave; wave `wrole` scans 4 / WS of the four directions with compile-time shifts, the verdicts meet in LDS
// (one 4-byte write, one s_barrier, one read per ply) and the record rows are split between the waves.  Move
// selection and the state update are redundant.  Twice / four times the waves of the one-lane form: for batches
// that leave SIMDs empty or alone with one wave, on boards whose scan dominates the ply (19x19).
// FAST (a template flag of ply / pick, one-lane form): every lane of the wave holds a CONSISTENT game -- as many stones
// on the board as plies counted, and fewer than C of them -- which is true of every state the env itself produces and
// stays true ply after ply (a ply adds one stone and one count; a finished game restarts at 0 / 0).  Then a board
// cannot be full when a ply begins, so the "no legal cell" branch (a compare and a scalar branch per ply) is gone;
// and the ply counter is redundant -- C minus the number of legal cells -- so its increment, its reset select and
// its store-back disappear from the loop (the draw test becomes "this was the last legal cell").  Poked states (stones
// placed through env.boards, counts set by hand) make the wave take the general loop for that launch: same results.
// SADDR (one-lane form only): the record stores address `uniform base + 32-bit lane offset` (global_store ... s[base])
// instead of a 64-bit pointer per lane -- two address instructions fewer per ply; one launch may then write at
// most 4 GiB of record rows (the launcher checks).
template <int NW, int CN, int CK, bool RECORD, int ACT = 0, bool PAIR = false, int WS = 1, bool SADDR = false>
struct RolloutLane {
  static constexpr bool EXACT = CN != 0;
  const MnkGeom& g;
  // The env in "mover first" form: cur = plane of the side to move, oth = the other plane.  A ply then ORs
  // one bit into cur, scans cur and swaps the two names (free: the loop is unrolled) -- no per-word selects
  // on the side bit as with (black, white) planes.  Black/white order is restored only where memory sees it.
  uint32_t cur[NW], oth[NW];
  uint32_t side, moves;
  int64_t N;
  // this lane's cursors into the record arrays; they advance by one ply's stride after every ply
  uint32_t role = 0;       // PAIR: 0 / 1 within the lane pair
  uint32_t wrole = 0;      // WS > 1: this wave's index in its workgroup (wave-uniform)
  uint32_t* vx = nullptr;  // WS > 1: LDS verdicts u32[2][64][WS] (double-buffered by ply parity)
  uint64_t* rp = nullptr;  // rec_planes[t][0][i]
  uint32_t* rm = nullptr;  // rec_meta[t][i]
  static constexpr bool OFF32 = SADDR || PAIR;  // the two-lane form always addresses its records this way
  const char* rbase[OFF32 ? NW : 1] = {};  // SADDR / PAIR: rec_planes[.][w][0] / rec_meta as wave-uniform bases ...
  const char* mbase = nullptr;
  uint32_t roff = 0, moff = 0;  // ... and this lane's byte offsets of rec_planes[t][0][i] / rec_meta[t][i]
  uint8_t* ra = nullptr;   // act_log[t / 4][i]  (ACT 3: the next word of the 7-bit stream, [w][i])
  uint64_t quad = 0;       // the actions of the current group of four plies
  uint32_t q32 = 0;        // ACT 3: the same, 4 x 7 bits; ACT 4: the low bytes of the group
  uint8_t* rh = nullptr;   // ACT 4: the next word of the bit plane of bit 8, [ceil(T/4) + p/32][i]
  uint32_t hi4 = 0, hiw = 0, hfill = 0;  // ACT 4: bit 8 of the group's actions, the word being filled, its fill (uniform)
  uint32_t lo = 0;         // ACT 3: the stream's accumulator: the bits of the word being filled ...
  uint32_t fill = 0;       // ... and how many of them are valid (wave-uniform: 0, 28, 24, ..., 4 between quads)
  // per-lane statistics, one add each per ply (T <= 65535 per launch): draws = done - wins, black wins =
  // wins - white wins; the summed length of the finished games needs no counter at all -- every ply adds one
  // to `moves` and a finished game takes its length out, so it is moves(start) + T - moves(end)
  uint32_t acc_done = 0, acc_win = 0, acc_white = 0;
  uint32_t moves_in = 0;
  uint32_t moves_parity = 0;  // WS > 1: plies played by this launch (selects the verdict buffer)

  __device__ __forceinline__ RolloutLane(const MnkGeom& g_, int64_t N_, int64_t i, uint64_t* rec_planes,
                                         uint32_t* rec_meta, void* act_log, uint32_t role_ = 0, uint32_t wrole_ = 0,
                                         uint32_t* vx_ = nullptr)
      : g(g_), N(N_), role(role_), wrole(wrole_), vx(vx_) {
    if (RECORD) {
      if (!PAIR) rp = rec_planes + i;
      rm = rec_meta + i;
      if (OFF32) {
#pragma unroll
        for (int w = 0; w < NW; ++w) rbase[w] = (const char*)(rec_planes + (int64_t)w * N);
        mbase = (const char*)rec_meta;
        roff = (uint32_t)i * 8u + (PAIR ? role * 4u : 0u);  // PAIR: lane `role` writes half `role` of every row
        moff = (uint32_t)i * 4u;
      }
    }
    static_assert((ACT != 3 && ACT != 4) || (!PAIR && WS == 1), "the bit-packed action logs are built into the one-lane form only");
    if (ACT) ra = (uint8_t*)act_log + i * 4 * (ACT == 2 ? 2 : 1);
  }

  // ACT 4: where the bit plane starts depends on the launch's length
  __device__ __forceinline__ void log_begin(void* act_log, int T, int64_t i) {
    if constexpr (ACT == 4) rh = (uint8_t*)act_log + ((int64_t)((T + 3) >> 2) * N + i) * 4;
  }

  __device__ __forceinline__ void load(const uint64_t* planes, const uint32_t* meta, int64_t i) {
    uint32_t p0[NW], p1[NW];
    plane_load<NW, EXACT>(p0, planes, N, g.W, i);
    plane_load<NW, EXACT>(p1, planes + (int64_t)g.W * N, N, g.W, i);
    const uint32_t mw = meta[i];
    side = mw & 1u;
    moves = moves_in = mw >> 1;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      cur[w] = side ? p1[w] : p0[w];
      oth[w] = side ? p0[w] : p1[w];
    }
  }

  // (black, white) planes of the current position to memory at `dst`
  __device__ __forceinline__ void store_planes(uint64_t* dst, int64_t i) const {
    uint32_t p0[NW], p1[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      p0[w] = side ? oth[w] : cur[w];
      p1[w] = side ? cur[w] : oth[w];
    }
    plane_store<NW, EXACT>(p0, dst, N, g.W, i);
    plane_store<NW, EXACT>(p1, dst + (int64_t)g.W * N, N, g.W, i);
  }

  // the position before a ply as one record -- mover's word | other side's word << 32, exactly the register
  // form, so no select on the side bit (it travels in the ply's meta word) -- cursor moves on to the next ply
  __device__ __forceinline__ void store_record() {
    if constexpr (PAIR) {  // lane 0 writes the mover's word of every row, lane 1 the other side's: 256 B per wave and store
#pragma unroll
      for (int w = 0; w < NW; ++w)
        __builtin_nontemporal_store(role ? oth[w] : cur[w], (uint32_t*)(rbase[w] + (uint64_t)roff));
      roff += (uint32_t)NW * (uint32_t)N * 8u;
      return;
    }
    if constexpr (WS > 1) {  // wave r writes rows [r*NW/WS, (r+1)*NW/WS): one taken uniform branch per ply
      static_assert(EXACT, "the waves-per-group form is built for the compile-time boards only");
#pragma unroll
      for (int r = 0; r < WS; ++r)
        if (wrole == (uint32_t)r) {
#pragma unroll
          for (int w = r * NW / WS; w < (r + 1) * NW / WS; ++w)
            __builtin_nontemporal_store((uint64_t)cur[w] | ((uint64_t)oth[w] << 32), rp + (int64_t)w * N);
        }
      rp += (int64_t)NW * N;
      return;
    }
    if constexpr (SADDR) {
#pragma unroll
      for (int w = 0; w < NW; ++w)
        if (EXACT || w < g.NW)
          __builtin_nontemporal_store((uint64_t)cur[w] | ((uint64_t)oth[w] << 32),
                                      (uint64_t*)(rbase[w] + (uint64_t)roff));
      roff += (uint32_t)g.NW * (uint32_t)N * 8u;
      return;
    }
#pragma unroll
    for (int w = 0; w < NW; ++w)
      if (EXACT || w < g.NW)
        __builtin_nontemporal_store((uint64_t)cur[w] | ((uint64_t)oth[w] << 32), rp + (int64_t)w * N);
    rp += (int64_t)g.NW * N;
  }

  __device__ __forceinline__ void store(uint64_t* planes, uint32_t* meta, int64_t i) const {
    if constexpr (PAIR) {  // lane `role` stores plane `role` (black = the mover's plane when black is to move)
      uint32_t mine[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) mine[w] = (role ^ side) ? oth[w] : cur[w];
      plane_store<NW, EXACT>(mine, planes + (int64_t)role * g.W * N, N, g.W, i);
    } else {
      store_planes(planes, i);
    }
    meta[i] = (moves << 1) | side;  // WS > 1: every wave of the group holds the same state; the kernel lets wave 0 store
  }

  __device__ __forceinline__ void log_flush() {
    if constexpr (ACT == 3) {  // 28 bits into the stream: the first quad of eight starts a word, every other completes one
      if (fill == 0u) {
        lo = q32;
        fill = 28u;
      } else {
        *(uint32_t*)ra = lo | (q32 << fill);
        ra += N * 4;
        lo = q32 >> (32u - fill);
        fill -= 4u;
      }
      q32 = 0;
      return;
    }
    if constexpr (ACT == 4) {  // the four low bytes as one word; the four high bits into the plane's current word
      *(uint32_t*)ra = q32;
      ra += N * 4;
      q32 = 0;
      hiw |= hi4 << hfill;
      hi4 = 0;
      hfill += 4u;
      if (hfill == 32u) {
        *(uint32_t*)rh = hiw;
        rh += N * 4;
        hiw = 0;
        hfill = 0;
      }
      return;
    }
    if (WS == 1 || wrole == 0) {
      if (ACT == 1) *(uint32_t*)ra = (uint32_t)quad;
      if (ACT == 2) *(uint64_t*)ra = quad;
    }
    ra += N * 4 * ACT;
    quad = 0;
  }

  // end of the launch: a partly filled group of four (T not a multiple of 4), then what is left in the stream
  __device__ __forceinline__ void log_finish(int T) {
    if (ACT && (T & 3)) log_flush();
    if constexpr (ACT == 3) {
      if (fill) *(uint32_t*)ra = lo;
    }
    if constexpr (ACT == 4) {
      if (hfill) *(uint32_t*)rh = hiw;
    }
  }

  // uniform legal cell from one u32 (oracle/philox.py pick_legal; selfplay/policy.py:18-29): the action, and
  // the cell's bit as a one-hot string
  template <bool FAST = false>
  __device__ __forceinline__ int pick(uint32_t x, uint32_t (&hot)[NW], uint32_t& nlegal) const {
    uint32_t legal[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) legal[w] = ~(cur[w] | oth[w]) & g.valid[w];
    const int nl = bs_popcount<NW>(legal);
    nlegal = (uint32_t)nl;
    int n = nl;
    // full board (poked states only): any cell, like RandomPolicy's 1e-8 guard -- the r-th valid cell is cell r.
    // A wave-uniform branch that is practically never taken: one compare and one scalar branch per ply.
    // FAST: a consistent game has a legal cell whenever a ply begins.
    if (!FAST && __builtin_amdgcn_ballot_w64(nl == 0) != 0) {
#pragma unroll
      for (int w = 0; w < NW; ++w) legal[w] = nl ? legal[w] : g.valid[w];
      n = nl ? nl : g.C;
    }
    const int r = (int)__umulhi(x, (uint32_t)n);
    const uint32_t bit = (uint32_t)bs_select_hot<NW>(legal, r, hot);
    return (int)(bit - (CN ? bit / (uint32_t)(CN + 1) : mnk_div(bit, g.magic_stride)));
  }

  // field = position of this ply inside its group of four (= step & 3; a compile-time constant in
  // the unrolled main loop, so the log costs one shift-or per ply and one wide store per four)
  template <bool FAST = false>
  __device__ __forceinline__ void ply(uint32_t x, int field) {
    uint32_t hot[NW];
    uint32_t nlegal;
    const int a = pick<FAST>(x, hot, nlegal);
    if constexpr (ACT == 3) {
      q32 |= (uint32_t)a << (7 * field);
      if (field == 3) log_flush();
    } else if constexpr (ACT == 4) {
      q32 |= ((uint32_t)a & 0xFFu) << (8 * field);
      hi4 |= ((uint32_t)a >> 8) << field;
      if (field == 3) log_flush();
    } else if (ACT) {
      quad |= (uint64_t)(uint32_t)a << (8 * ACT * field);
      if (field == 3) log_flush();
    }
    ply_hot<FAST>(a, hot, nlegal);
  }

  // one ply with a known-good action (from an action log the sampler wrote)
  __device__ __forceinline__ void ply_action(int a) {
    const uint32_t ua = (uint32_t)a;
    ply_bit(a, ua + (CN ? ua / (uint32_t)CN : mnk_div(ua, g.magic_n)));
  }

  __device__ __forceinline__ void ply_bit(int a, uint32_t bit) {
    const int wsel = (int)(bit >> 5);
    const uint32_t one = 1u << (bit & 31u);
    uint32_t hot[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) hot[w] = (w == wsel) ? one : 0u;
    ply_hot<false>(a, hot);
  }

  // env/torch_vector_mnk_env.py:60-84 for the mover, then env.reset(nonzero(done)) :34-44
  // FAST (see the top of this file): `moves` is not maintained -- the board is full after this ply iff the move took
  // the last legal cell (nlegal == 1); finish_fast() restores the counter from the board when the loop is over
  template <bool FAST = false>
  __device__ __forceinline__ void ply_hot(int a, const uint32_t (&hot)[NW], uint32_t nlegal = 0) {
    if (RECORD) store_record();
#pragma unroll
    for (int w = 0; w < NW; ++w) cur[w] |= hot[w];                    // :68
    if (!FAST) ++moves;                                               // :69
    uint32_t win;                                                     // :71
    if constexpr (PAIR) {
      // role 0 scans columns and rows, role 1 diagonals and anti-diagonals; paired so that the word parts of
      // the shift amounts agree wherever the board allows (n+1 with n+2, 1 with n)
      uint32_t hit = bs_run_bits_pair<NW, CK, CN + 1, CN + 2>(cur, role) | bs_run_bits_pair<NW, CK, 1, CN>(cur, role);
      hit |= pair_swap(hit);  // the partner's two directions
      win = hit ? 1u : 0u;
    } else if constexpr (WS > 1) {
      // this wave's share of the four directions (uniform branch, compile-time shifts inside), then the
      // workgroup's verdicts through LDS: [parity][lane][wave], so the read is one 8 / 16-byte load per lane
      bool hit = false;
      if constexpr (WS == 4) {
        if (wrole == 0) hit = bs_has_run<NW>(cur, 1, CK);
        else if (wrole == 1) hit = bs_has_run<NW>(cur, CN + 1, CK);
        else if (wrole == 2) hit = bs_has_run<NW>(cur, CN + 2, CK);
        else hit = bs_has_run<NW>(cur, CN, CK);
      } else {
        if (wrole == 0) hit = bs_has_run<NW>(cur, 1, CK) | bs_has_run<NW>(cur, CN + 1, CK);
        else hit = bs_has_run<NW>(cur, CN + 2, CK) | bs_has_run<NW>(cur, CN, CK);
      }
      uint32_t* slot = vx + ((moves_parity & 1u) * 64u + (threadIdx.x & 63u)) * WS;
      slot[wrole] = hit ? 1u : 0u;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS only: the record stores stay in flight
      uint32_t any = 0;
#pragma unroll
      for (int r = 0; r < WS; ++r) any |= slot[r];
      win = any;
      ++moves_parity;
    } else {
      win = mnk_plane_wins<NW, CN, CK>(g, cur) ? 1u : 0u;
    }
    const uint32_t done = win | ((FAST ? nlegal == 1u : moves >= (uint32_t)g.C) ? 1u : 0u);   // :72-73
    if (RECORD) {
      const uint32_t mword = (uint32_t)a | (win << MNK_REC_REWARD_SHIFT) | (done << MNK_REC_DONE_BIT) | (side << MNK_REC_SIDE_BIT);
      if constexpr (OFF32) {
        __builtin_nontemporal_store(mword, (uint32_t*)(mbase + (uint64_t)moff));
        moff += (uint32_t)N * 4u;
      } else {
        if (WS == 1 || wrole == WS - 1) __builtin_nontemporal_store(mword, rm);
        rm += N;
      }
    }
    acc_done += done;
    acc_win += win;
    acc_white += win & side;
    // the other side is to move (:82) -- or a fresh game
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const uint32_t c = cur[w];
      cur[w] = done ? 0u : oth[w];
      oth[w] = done ? 0u : c;
    }
    side = done ? 0u : (side ^ 1u);
    if (!FAST) moves = done ? 0u : moves;
  }

  // is this lane's game consistent (stones on the board == plies counted < C)?  -> the wave may take the FAST loop
  __device__ __forceinline__ bool consistent() const {
    uint32_t occ[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) occ[w] = cur[w] | oth[w];
    return (uint32_t)bs_popcount<NW>(occ) == moves && moves < (uint32_t)g.C;
  }

  // after a FAST loop: the ply counter of a consistent game is its stone count
  __device__ __forceinline__ void finish_fast() {
    uint32_t occ[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) occ[w] = cur[w] | oth[w];
    moves = (uint32_t)bs_popcount<NW>(occ);
  }
};

// the T plies of one lane: head (finish the Philox block the previous launch stopped in), groups of four plies on
// one Philox block each with the word picked at compile time, tail
template <bool FAST, typename Lane>
__device__ __forceinline__ void rollout_plies(Lane& L, int T, uint64_t seed, uint64_t step0, uint64_t env) {
  int t = 0;
  uint64_t step = step0;
  if (step & 3) {
    const Philox4 blk = mnk_rng_block(seed, env, step >> 2, MNK_STREAM_MOVE);
    for (; t < T && (step & 3); ++t, ++step)
      L.template ply<FAST>(philox_word(blk, (uint32_t)(step & 3)), (int)(step & 3));
  }
  for (; t + 4 <= T; t += 4, step += 4) {
    const Philox4 blk = mnk_rng_block(seed, env, step >> 2, MNK_STREAM_MOVE);
    L.template ply<FAST>(blk.v[0], 0);
    L.template ply<FAST>(blk.v[1], 1);
    L.template ply<FAST>(blk.v[2], 2);
    L.template ply<FAST>(blk.v[3], 3);
  }
  if (t < T) {
    const Philox4 blk = mnk_rng_block(seed, env, step >> 2, MNK_STREAM_MOVE);
    for (uint32_t j = 0; t < T; ++t, ++j) L.template ply<FAST>(philox_word(blk, j), (int)j);
  }
}

// the one-lane kernel's body: one wave of 64 envs per workgroup of 64 threads
template <int NW, int CN, int CK, bool RECORD, int ACT, bool SADDR = false>
__device__ __forceinline__ void rollout_random_body(const MnkGeom& g, uint64_t* planes, uint32_t* meta, int64_t N, int T,
                                                    uint64_t seed, uint64_t step0, int64_t env_id0,
                                                    uint64_t* rec_planes, uint32_t* rec_meta, unsigned long long* stats,
                                                    void* act_log) {
  // one full wave of 64 envs per workgroup: half-filled waves were measured and are slower
  // (gfx950 does not skip the idle half of a wave64), see DESIGN.md
  __shared__ unsigned int lds_stats[MNK_STATS_COUNTERS];
  if (threadIdx.x < MNK_STATS_COUNTERS) lds_stats[threadIdx.x] = 0u;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) {
    RolloutLane<NW, CN, CK, RECORD, ACT, false, 1, SADDR> L(g, N, i, rec_planes, rec_meta, act_log);
    L.log_begin(act_log, T, i);
    L.load(planes, meta, i);
    const uint64_t env = (uint64_t)(env_id0 + i);
    // every lane of this wave in a consistent game (any state the env produced itself): the FAST loop; a wave that
    // holds a poked state plays this launch on the general loop.  Same records either way.
    // Only where it measured faster (same-box A/B, us per 256 plies at 65 536 envs: 3x3x3 59.1 -> 53.4, 9x9x5 88.1 ->
    // 82.7, 7x9x7 106.9 -> 100.8, 13x13x5 143 -> 135.8): there the kernel is bound by its instruction count.  Where
    // the HBM write rate is the bound the second loop cost more than it saved (12x12x5 138 -> 147, 15x15x5 205 -> 218,
    // 19x19x5 267 -> 305: the compiler's schedule of the record stores changed), so those keep the one general loop.
    constexpr bool TRY_FAST = NW <= 3 || (NW == 6 && CN == 13);
    if (TRY_FAST && __builtin_amdgcn_ballot_w64(!L.consistent()) == 0) {
      rollout_plies<true>(L, T, seed, step0, env);
      L.finish_fast();
    } else {
      rollout_plies<false>(L, T, seed, step0, env);
    }
    L.log_finish(T);  // T not a multiple of 4: the last word is partly filled
    L.store(planes, meta, i);
    if (stats) {
      const uint32_t len_sum = L.moves_in + (uint32_t)T - L.moves;
      if (L.acc_done) atomicAdd(&lds_stats[0], L.acc_done);
      if (L.acc_win - L.acc_white) atomicAdd(&lds_stats[1], L.acc_win - L.acc_white);
      if (L.acc_white) atomicAdd(&lds_stats[2], L.acc_white);
      if (L.acc_done - L.acc_win) atomicAdd(&lds_stats[3], L.acc_done - L.acc_win);
      if (len_sum) atomicAdd(&lds_stats[4], len_sum);
    }
  }
  __syncthreads();
  // one global atomic per counter per wave, spread over MNK_STATS_REPLICAS cache lines: thousands
  // of adds on five addresses would serialise at ~11 ns each (measured: 56 us per launch)
  if (stats && threadIdx.x < MNK_STATS_COUNTERS && lds_stats[threadIdx.x])
    atomicAdd(&stats[(size_t)(blockIdx.x % MNK_STATS_REPLICAS) * MNK_STATS_STRIDE + threadIdx.x],
              (unsigned long long)lds_stats[threadIdx.x]);
}

// ------------------------------------------------------------------ two lanes per env (scan directions split)
// the T plies of one lane pair (FAST: see mnk_rollout_lane.h -- every game of the wave consistent)
template <bool FAST, typename Lane>
__device__ __forceinline__ void pair_plies(Lane& L, int T, uint64_t seed, uint64_t step0, uint64_t env, uint32_t role) {
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
      const uint32_t other = pair_swap(mine.v[j]);
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
__device__ __forceinline__ void rollout_random_pair_body(const MnkGeom& g, uint64_t* planes, uint32_t* meta, int64_t N, int T,
                                                         uint64_t seed, uint64_t step0, int64_t env_id0, uint64_t* rec_planes,
                                                         uint32_t* rec_meta, unsigned long long* stats, void* act_log) {
  __shared__ unsigned int lds_stats[MNK_STATS_COUNTERS];
  if (threadIdx.x < MNK_STATS_COUNTERS) lds_stats[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t role = threadIdx.x & 1u;
  const int64_t i = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 1);  // env of this lane pair
  if (i < N) {
    RolloutLane<NW, CN, CK, RECORD, ACT, true> L(g, N, i, rec_planes, rec_meta, act_log, role);
    L.load(planes, meta, i);
    const uint64_t env = (uint64_t)(env_id0 + i);
    // both lanes of a pair hold the whole board: the consistency test of the one-lane form applies as it is.  Small
    // boards only (what the launcher uses this form for: 3x3 and 9x9); the larger ones keep the one general loop.
    constexpr bool TRY_FAST = NW <= 3;
    if (TRY_FAST && __builtin_amdgcn_ballot_w64(!L.consistent()) == 0) {
      pair_plies<true>(L, T, seed, step0, env, role);
      L.finish_fast();
    } else {
      pair_plies<false>(L, T, seed, step0, env, role);
    }
    if (ACT && (T & 3)) L.log_flush();
    L.store(planes, meta, i);  // lane `role` stores plane `role`; the meta word is written by both
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

// ------------------------------------------------------------------ replay of an action log (one lane per env)
// The receiving side of the multi-GPU exchange: a shard's rollout is fully determined by its chunk-start state and its
// action log, so that is what crosses xGMI; this re-plays the log and rebuilds the full packed records, bit-identical to
// the sender's.  ACTB = the log format (MNK_ACT_U8 / _U16 / _BITS7 / _U8P1), a template parameter like everything else
// that shapes the ply loop.
template <int NW, int CN, int CK, bool RECORD, int ACTB>
__device__ __forceinline__ void replay_actions_body(const MnkGeom& g, uint64_t* planes, uint32_t* meta, int64_t N, int T,
                                                    const void* act_log, uint64_t* rec_planes, uint32_t* rec_meta,
                                                    int32_t* err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  RolloutLane<NW, CN, CK, RECORD> L(g, N, i, rec_planes, rec_meta, nullptr);
  L.load(planes, meta, i);
  if constexpr (ACTB == 3) {
    bool bad7 = false;
    auto play7 = [&](uint32_t a) {
      if (a >= (uint32_t)g.C) { bad7 = true; a = 0; }
      L.ply_action((int)a);
    };
    const int quads = (T + 3) >> 2, nwords = (7 * quads + 7) >> 3;
    const uint32_t* src = (const uint32_t*)act_log + i;
    uint32_t ahead = nwords ? src[0] : 0u;
    int w = 1;
    uint32_t cur = 0, have = 0;  // bits left over from the last word (low-aligned) and their number: 0, 4, ..., 28 (uniform)
    auto take_word = [&]() -> uint32_t {
      const uint32_t word = ahead;
      ahead = src[(int64_t)(w < nwords ? w : nwords - 1) * N];
      ++w;
      return word;
    };
    auto next_quad = [&]() -> uint32_t {  // 32-bit arithmetic only; seven words per eight quads
      uint32_t q;
      if (have == 28u) {
        q = cur;
        cur = 0u;
        have = 0u;
      } else {
        const uint32_t word = take_word();
        q = (cur | (word << have)) & 0x0FFFFFFFu;  // have == 0: cur == 0
        cur = word >> (28u - have);
        have += 4u;
      }
      return q;
    };
    int t = 0;
    for (; t + 4 <= T; t += 4) {
      const uint32_t q = next_quad();
      play7(q & 0x7Fu);
      play7((q >> 7) & 0x7Fu);
      play7((q >> 14) & 0x7Fu);
      play7((q >> 21) & 0x7Fu);
    }
    if (t < T) {
      uint32_t q = next_quad();
      for (; t < T; ++t, q >>= 7) play7(q & 0x7Fu);
    }
    if (bad7) mnk_report(err, MNK_ERR_ACTION_RANGE, i);
    L.store(planes, meta, i);
    return;
  }
  if constexpr (ACTB == 4) {  // MNK_ACT_U8P1: a word of four low bytes per group, a word of 32 high bits per 32 plies
    bool bad9 = false;
    auto play9 = [&](uint32_t a) {
      if (a >= (uint32_t)g.C) { bad9 = true; a = 0; }
      L.ply_action((int)a);
    };
    const int quads = (T + 3) >> 2, hwords = (T + 31) >> 5;
    const uint32_t* lo = (const uint32_t*)act_log + i;
    const uint32_t* hi = lo + (int64_t)quads * N;
    uint32_t ahead = quads ? lo[0] : 0u;
    uint32_t hbits = hwords ? hi[0] : 0u;
    int t = 0;
    for (int q = 0; t < T; ++q) {
      const uint32_t word = ahead;
      ahead = lo[(int64_t)(q + 1 < quads ? q + 1 : q) * N];
      if (q && (q & 7) == 0) hbits = hi[(int64_t)(q >> 3) * N];  // plies 4q .. 4q+3 are bits (4q .. 4q+3) % 32 of word q / 8
      const uint32_t h4 = hbits >> (4 * (q & 7));
      if (t + 4 <= T) {
        play9((word & 0xFFu) | ((h4 & 1u) << 8));
        play9(((word >> 8) & 0xFFu) | ((h4 & 2u) << 7));
        play9(((word >> 16) & 0xFFu) | ((h4 & 4u) << 6));
        play9((word >> 24) | ((h4 & 8u) << 5));
        t += 4;
      } else {
        for (int j = 0; t < T; ++t, ++j) play9(((word >> (8 * j)) & 0xFFu) | (((h4 >> j) & 1u) << 8));
      }
    }
    if (bad9) mnk_report(err, MNK_ERR_ACTION_RANGE, i);
    L.store(planes, meta, i);
    return;
  }
  constexpr uint32_t FIELD = ACTB == 1 ? 0xFFu : 0xFFFFu;
  auto fetch = [&](int q) -> uint64_t {
    if (ACTB == 1) return (uint64_t)((const uint32_t*)act_log)[(int64_t)q * N + i];
    return ((const uint64_t*)act_log)[(int64_t)q * N + i];
  };
  bool bad = false;
  auto play = [&](uint32_t a) {
    if (a >= (uint32_t)g.C) { bad = true; a = 0; }  // a log we did not write: flag it, keep the wave in step
    L.ply_action((int)a);
  };
  const int words = (T + 3) >> 2;
  uint64_t ahead = words ? fetch(0) : 0;
  int t = 0;
  for (int q = 0; t + 4 <= T; ++q, t += 4) {
    const uint64_t quad = ahead;
    ahead = fetch(q + 1 < words ? q + 1 : q);
    play((uint32_t)(quad >> (0 * 8 * ACTB)) & FIELD);
    play((uint32_t)(quad >> (1 * 8 * ACTB)) & FIELD);
    play((uint32_t)(quad >> (2 * 8 * ACTB)) & FIELD);
    play((uint32_t)(quad >> (3 * 8 * ACTB)) & FIELD);
  }
  for (uint64_t quad = ahead; t < T; ++t, quad >>= 8 * ACTB) play((uint32_t)quad & FIELD);  // a partly filled last word
  if (bad) mnk_report(err, MNK_ERR_ACTION_RANGE, i);
  L.store(planes, meta, i);
}


#ifdef MNK_JIT_NW
// run-time specialisation (mnk_jit.hip): this board's geometry arrives as macros on the hiprtc command line
#if !defined(MNK_JIT_KIND) || MNK_JIT_KIND == 0
extern "C" __global__ void __launch_bounds__(64)
mnk_jit_rollout(MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed, uint64_t step0,
                int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, unsigned long long* stats, void* act_log) {
  rollout_random_body<MNK_JIT_NW, MNK_JIT_CN, MNK_JIT_CK, MNK_JIT_REC != 0, MNK_JIT_ACT, MNK_JIT_SADDR != 0>(
      g, planes, meta, N, T, seed, step0, env_id0, rec_planes, rec_meta, stats, act_log);
}
#elif MNK_JIT_KIND == 2
// the two-lanes-per-env form (scan directions split): 32 envs per wave, for batches of up to 32 768 envs
extern "C" __global__ void __launch_bounds__(64)
mnk_jit_rollout_pair(MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed, uint64_t step0,
                     int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, unsigned long long* stats, void* act_log) {
  rollout_random_pair_body<MNK_JIT_NW, MNK_JIT_CN, MNK_JIT_CK, MNK_JIT_REC != 0, MNK_JIT_ACT>(
      g, planes, meta, N, T, seed, step0, env_id0, rec_planes, rec_meta, stats, act_log);
}
#else
// MNK_JIT_KIND == 1: the replay of an action log in format MNK_JIT_ACT
extern "C" __global__ void __launch_bounds__(64)
mnk_jit_replay(MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, int T, const void* act_log, uint64_t* rec_planes,
               uint32_t* rec_meta, int32_t* err) {
  replay_actions_body<MNK_JIT_NW, MNK_JIT_CN, MNK_JIT_CK, MNK_JIT_REC != 0, MNK_JIT_ACT>(g, planes, meta, N, T, act_log,
                                                                                         rec_planes, rec_meta, err);
}
#endif
#else
template <int NW, int CN, int CK, bool RECORD, int ACT, bool SADDR = false>
__global__ void __launch_bounds__(64)
k_rollout_random(MnkGeom g, uint64_t* planes, uint32_t* meta, int64_t N, int T, uint64_t seed, uint64_t step0,
                 int64_t env_id0, uint64_t* rec_planes, uint32_t* rec_meta, unsigned long long* stats,
                 void* act_log) {
  rollout_random_body<NW, CN, CK, RECORD, ACT, SADDR>(g, planes, meta, N, T, seed, step0, env_id0, rec_planes, rec_meta,
                                                      stats, act_log);
}
#endif

// mnk_emit.h -- LDS-staged write-out of observations and legal masks (gfx950 only).
//
// The game logic runs one lane per env on bit-packed boards, but the reference's
// consumers want dense row-major tensors: observation f32[N][2][m][n] (648 B/env at
// 9x9) and action_mask bool[N][m*n] (env/torch_vector_mnk_env.py:46-53).  Written by
// the env's own lane those would be 64 lanes x 648-byte stride -- uncoalesced.  So a
// workgroup parks the packed planes of its B envs in LDS (3*NW 32-bit words per env:
// channel 0, channel 1, legal cells), and then all its lanes sweep the workgroup's contiguous
// output slab in 16-byte vectors, each lane expanding the bits it needs from LDS.
// HBM sees only full-width coalesced stores; the slab of workgroup b is the byte range
// [b*B*rowbytes, (b+1)*B*rowbytes).
#pragma once
#include "mnk_device.h"

// LDS image: u32 stage[3*NW][B + 1] (odd row stride, see mnk_stage_stride) followed by u32 tab_obs[2C] and
// u32 tab_mask[C].  A table entry = (index of the word for env 0) << 5 | bit-in-word; lanes that expand
// neighbouring cells read the same word (a broadcast) or words of different rows (different banks).
struct MnkStage {
  uint32_t* words;    // [3*NW][B + 1]
  uint32_t* tab_obs;  // [2C]
  uint32_t* tab_mask; // [C]
  uint32_t* segs;     // packed write-out: [2B + 1][SW] channel strings, then [B + 1][SW] legal strings
};

// row stride of the stage in words: B + 1, not B.  With B = 64 = the number of LDS banks, word w of env e sat in bank
// e for every w, and the write-out -- where the lanes of a wave read a handful of DIFFERENT words of the SAME one or two
// envs -- serialised 5-way (9x9) to 8-way (19x19) on that one bank.  With the odd stride word w of env e sits in bank
// (w + e) mod 64: different words, different banks; lanes that want the same word are a broadcast.
__host__ __device__ inline int mnk_stage_stride(int B) { return B + 1; }

// ---- packed ("gap-free") strings, boards with a compile-time variant only -------------------------------------------
// The staged words still carry the guard column; every float / mask byte of the output then costs a table lookup, a
// word read and a wrap test (round 1: ~10 instructions and 2 LDS reads per cell; the bool mask, 1/8 of the bytes, took
// half as long as the f32 observation).  For the boards with compile-time geometry the write-out therefore runs on
// strings WITHOUT the guard bits: after the stage is filled, 3*B threads squeeze one string each (channel 0, channel 1,
// legal cells of one env; row r of n bits moves from bit r*(n+1) to bit r*n, all shifts immediates) into
//   segs  u32[2B + 1][SW]   segment 2*el + ch = channel ch of env el, C valid bits   (obs row of env el = its two segments)
//   msegs u32[ B + 1][SW]   segment el = legal cells of env el
// and 4 floats / 16 mask bytes of the output are then 4 / 16 CONSECUTIVE bits of a segment (continuing into the next
// segment where one ends): two LDS words and a funnel shift, spread to bytes with one multiply
// (nibble * 0x00204081 & 0x01010101) and, for the observation, to floats with v_cvt_f32_ubyte0..3.
__host__ __device__ inline int mnk_seg_words(int NW, int n) {  // SW: words of one segment (odd, >= 1 zero pad word)
  const int maxrows = 32 * NW / (n + 1);
  const int cw = (maxrows * n + 31) / 32;
  return (cw + 1) | 1;
}

// The boards the ahead-of-time dispatch (MNK_DISPATCH, mnk_host.h) has compile-time geometry for; every other board gets
// its compile-time variant from hiprtc once it is hot (mnk_jit.hip).
__host__ __device__ inline bool mnk_geom_builtin(int n, int k, int NW) {
  return (n == 9 && k == 5 && NW == 3) || (n == 3 && k == 3 && NW == 1) || (n == 13 && k == 5 && NW == 6) ||
         (n == 15 && k == 5 && NW == 8) || (n == 19 && k == 5 && NW == 12);
}

// does a kernel variant with compile-time geometry (CN = n) write out in the packed form on a board of C cells?
// (16 mask bytes must not span more than two envs; the squeeze moves a board row as one 32-bit word)
__host__ __device__ inline bool mnk_packed_cells(int n, int C) { return C >= 16 && n <= 31; }

// host side of the same rule for the ahead-of-time kernels: what MNK_DISPATCH will pick for this geometry
__host__ __device__ inline bool mnk_geom_packed(int n, int k, int NW, int C) {
  return mnk_geom_builtin(n, k, NW) && mnk_packed_cells(n, C);
}

// dynamic LDS of the write-out stage; `packed`: the launched variant has compile-time geometry and C >= 16
__host__ __device__ inline size_t mnk_stage_bytes(int NW, int C, int B, int n, bool packed) {
  size_t bytes = (size_t)3 * NW * mnk_stage_stride(B) * 4 + (size_t)3 * C * 4;
  if (packed) bytes += (size_t)(3 * B + 2) * mnk_seg_words(NW, n) * 4;
  return bytes;
}

// in a kernel: is this variant (compile-time board width CN, 0 = generic) on the packed write-out?
template <int CN>
__device__ __forceinline__ bool mnk_packed_form(const MnkGeom& g) { return CN != 0 && mnk_packed_cells(CN, g.C); }

__device__ __forceinline__ MnkStage mnk_stage_carve(void* lds, const MnkGeom& g, int B) {
  MnkStage s;
  s.words = (uint32_t*)lds;
  s.tab_obs = s.words + (size_t)3 * g.NW * mnk_stage_stride(B);
  s.tab_mask = s.tab_obs + 2 * g.C;
  s.segs = s.tab_obs + 3 * g.C;
  return s;
}

template <int CN>
__device__ __forceinline__ void mnk_stage_tables(const MnkStage& s, const MnkGeom& g, int B, int tid, int nthreads) {
  if (mnk_packed_form<CN>(g)) return;  // the packed write-out needs no cell tables
  for (int r = tid; r < 3 * g.C; r += nthreads) {
    const int plane = r >= 2 * g.C ? 2 : (r >= g.C ? 1 : 0);
    const uint32_t cell = (uint32_t)(r - plane * g.C);
    const uint32_t bit = cell + mnk_div(cell, g.magic_n);
    const uint32_t wq = (uint32_t)(plane * g.NW) + (bit >> 5);
    const uint32_t entry = ((wq * (uint32_t)mnk_stage_stride(B)) << 5) | (bit & 31u);
    s.tab_obs[r] = entry;  // tab_mask aliases tab_obs + 2C
  }
}

// one env's planes into the stage; ch0/ch1 already in the order the viewer wants
template <int NW>
__device__ __forceinline__ void mnk_stage_put(const MnkStage& s, const MnkGeom& g, int B, int el,
                                              const uint32_t (&ch0)[NW], const uint32_t (&ch1)[NW],
                                              bool fix_empty) {
  uint32_t any = 0;
  uint32_t legal[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    legal[w] = ~(ch0[w] | ch1[w]) & g.valid[w];
    any |= legal[w];
  }
  if (fix_empty && any == 0) legal[0] = 1u;  // wrapper:108-110  mask[invalid, 0] = True
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    if (w < g.NW) {
      const int S = mnk_stage_stride(B);
      s.words[(0 * g.NW + w) * S + el] = ch0[w];
      s.words[(1 * g.NW + w) * S + el] = ch1[w];
      s.words[(2 * g.NW + w) * S + el] = legal[w];
    }
  }
}

__device__ __forceinline__ uint32_t mnk_stage_bit(const uint32_t* st32, uint32_t entry, uint32_t el) {
  return (st32[(entry >> 5) + el] >> (entry & 31u)) & 1u;
}

// bytes per observation cell for MNK_OBS_F32 / MNK_OBS_BF16 / MNK_OBS_U8
__host__ __device__ inline int mnk_obs_bytes(int obs_dtype) { return obs_dtype == MNK_OBS_F32 ? 4 : (obs_dtype == MNK_OBS_BF16 ? 2 : 1); }
__host__ __device__ inline bool mnk_obs_dtype_ok(int obs_dtype) { return obs_dtype >= MNK_OBS_F32 && obs_dtype <= MNK_OBS_U8; }

// Table form (any board): a slab of nb rows x `row` cells starting at dst (16-byte aligned when vec), EB bytes per
// cell -- 4: f32 1.0 / 0.0 (observation), 2: bf16 1.0 / 0.0, 1: byte 1 / 0 (u8 observation, bool mask).  `tab` holds
// one entry per cell of a row (tab_obs: 2C entries, tab_mask: C entries), magic_row = the magic of `row`.
template <int EB>
__device__ __forceinline__ void mnk_emit_table(const MnkStage& s, const uint32_t* tab, uint32_t row, uint32_t magic_row,
                                               int nb, void* dst, bool vec, int tid, int nthreads) {
  constexpr uint32_t PER = 16 / EB, PW = 4 / EB;  // cells per 16-byte vector / per 32-bit word
  constexpr uint32_t ONE = EB == 4 ? 0x3F800000u : (EB == 2 ? 0x3F80u : 1u);
  const uint32_t* st32 = s.words;
  const uint32_t total = (uint32_t)nb * row;
  const uint32_t nvec = vec ? total / PER : 0u;
  for (uint32_t q = tid; q < nvec; q += nthreads) {
    const uint32_t e = q * PER;
    uint32_t el = mnk_div(e, magic_row);
    uint32_t rem = e - el * row;
    uint32_t out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t acc = 0;
#pragma unroll
      for (uint32_t b = 0; b < PW; ++b) {
        if (rem == row) { rem = 0; ++el; }
        acc |= (mnk_stage_bit(st32, tab[rem], el) * ONE) << (8 * EB * b);
        ++rem;
      }
      out[j] = acc;
    }
    reinterpret_cast<uint4*>(dst)[q] = make_uint4(out[0], out[1], out[2], out[3]);
  }
  for (uint32_t e = nvec * PER + tid; e < total; e += nthreads) {
    const uint32_t el = mnk_div(e, magic_row);
    const uint32_t v = mnk_stage_bit(st32, tab[e - el * row], el) * ONE;
    if (EB == 4) reinterpret_cast<uint32_t*>(dst)[e] = v;
    else if (EB == 2) reinterpret_cast<uint16_t*>(dst)[e] = (uint16_t)v;
    else reinterpret_cast<uint8_t*>(dst)[e] = (uint8_t)v;
  }
}


// ---------------------------------------------------------------- packed write-out (see mnk_seg_words above)
// one thread = one string: squeeze the guard bits out of NW staged words into CW words (+ zero pad up to SW)
template <int NW, int CN>
__device__ __forceinline__ void mnk_stage_squeeze(const MnkStage& s, int B, int tid) {
  constexpr int MAXROWS = 32 * NW / (CN + 1), CW = (MAXROWS * CN + 31) / 32, SW = (CW + 1) | 1;
  if (tid >= 3 * B) return;
  const int plane = tid / B, el = tid - plane * B, S = mnk_stage_stride(B);
  uint32_t x[NW + 1];
#pragma unroll
  for (int w = 0; w < NW; ++w) x[w] = s.words[(plane * NW + w) * S + el];
  x[NW] = 0u;
  uint32_t out[CW + 1];
#pragma unroll
  for (int w = 0; w <= CW; ++w) out[w] = 0u;
#pragma unroll
  for (int r = 0; r < MAXROWS; ++r) {
    constexpr uint32_t rowmask = (1u << CN) - 1u;
    const int sb = r * (CN + 1), q = sb >> 5, sh = sb & 31;
    const uint32_t bits = ((sh + CN <= 32) ? (x[q] >> sh) : __builtin_amdgcn_alignbit(x[q + 1], x[q], (uint32_t)sh)) & rowmask;
    const int db = r * CN, o = db >> 5, os = db & 31;
    out[o] |= bits << os;
    if (os + CN > 32) out[o + 1] |= bits >> (32 - os);
  }
  uint32_t* dst = s.segs + (size_t)(plane < 2 ? 2 * el + plane : 2 * B + 1 + el) * SW;
#pragma unroll
  for (int w = 0; w < SW; ++w) dst[w] = w < CW ? out[w] : 0u;
}

// 32 bits starting at bit `pos` of segment `sg` (C valid bits per segment), continuing with the first bits of
// segment sg + 1 where the segment ends within the first K bits
template <int K>
__device__ __forceinline__ uint32_t mnk_seg_bits(const uint32_t* segs, int SW, uint32_t sg, uint32_t pos, uint32_t C) {
  const uint32_t* p = segs + (size_t)sg * SW;
  const uint32_t w = pos >> 5;
  const uint32_t x = __builtin_amdgcn_alignbit(p[w + 1], p[w], pos & 31u);
  const uint32_t left = C - pos;  // bits of this segment from `pos` on (>= 1)
  const uint32_t next = p[SW];    // word 0 of the next segment (a zero pad segment follows the last one)
  const uint32_t joined = (x & ((1u << (left & 31u)) - 1u)) | (next << (left & 31u));
  return left < (uint32_t)K ? joined : x;
}

// 4 cells (low nibble) -> 4 bytes of 0 / 1
__device__ __forceinline__ uint32_t mnk_spread4(uint32_t nib) { return ((nib & 0xFu) * 0x00204081u) & 0x01010101u; }

// 8 cells (low byte) -> 4 words of two bf16 1.0 / 0.0 each
__device__ __forceinline__ uint32_t mnk_spread_bf16x2(uint32_t two) { return ((two & 1u) | ((two & 2u) << 15)) * 0x3F80u; }

// Packed form: `total` cells = consecutive segments of C valid bits each (segs: 2 per env for the observation, 1 per
// env for the legal mask), EB bytes per cell as in mnk_emit_table; 16 / EB cells per 16-byte store.
template <int NW, int CN, int EB>
__device__ __forceinline__ void mnk_emit_packed(const uint32_t* segs, const MnkGeom& g, uint32_t total, void* dst, bool vec,
                                                int tid, int nthreads) {
  constexpr int MAXROWS = 32 * NW / (CN + 1), CW = (MAXROWS * CN + 31) / 32, SW = (CW + 1) | 1;
  constexpr uint32_t PER = 16 / EB;
  const uint32_t C = (uint32_t)g.C;
  const uint32_t nvec = vec ? total / PER : 0u;
  for (uint32_t q = tid; q < nvec; q += nthreads) {
    const uint32_t e = q * PER, sg = mnk_div(e, g.magic_C), pos = e - sg * C;
    const uint32_t x = mnk_seg_bits<(int)PER>(segs, SW, sg, pos, C);
    // (wave-uniform slab base + 32-bit lane offset: global_store ... s[base])
    char* at = (char*)dst + (uint64_t)(q << 4);
    if constexpr (EB == 4) {
      const uint32_t b = mnk_spread4(x);
      *reinterpret_cast<float4*>(at) =
          make_float4((float)(b & 0xFFu), (float)((b >> 8) & 0xFFu), (float)((b >> 16) & 0xFFu), (float)(b >> 24));
    } else if constexpr (EB == 2) {
      *reinterpret_cast<uint4*>(at) = make_uint4(mnk_spread_bf16x2(x), mnk_spread_bf16x2(x >> 2), mnk_spread_bf16x2(x >> 4),
                                                 mnk_spread_bf16x2(x >> 6));
    } else {
      *reinterpret_cast<uint4*>(at) = make_uint4(mnk_spread4(x), mnk_spread4(x >> 4), mnk_spread4(x >> 8), mnk_spread4(x >> 12));
    }
  }
  for (uint32_t e = nvec * PER + tid; e < total; e += nthreads) {
    const uint32_t sg = mnk_div(e, g.magic_C), pos = e - sg * C;
    const uint32_t bit = (segs[(size_t)sg * SW + (pos >> 5)] >> (pos & 31u)) & 1u;
    if (EB == 4) reinterpret_cast<float*>(dst)[e] = (float)bit;
    else if (EB == 2) reinterpret_cast<uint16_t*>(dst)[e] = (uint16_t)(bit * 0x3F80u);
    else reinterpret_cast<uint8_t*>(dst)[e] = (uint8_t)bit;
  }
}

// The write-out of a workgroup whose stage has been filled by mnk_stage_put: call from ALL threads of the workgroup
// under a workgroup-uniform condition (it synchronises).  obs / mask may be NULL; they point at the workgroup's slab
// (obs: nb rows of 2C cells of mnk_obs_bytes(obs_dtype) bytes; mask: nb rows of C bytes).
template <int NW, int CN, int CK>
__device__ __forceinline__ void mnk_write_out(const MnkStage& s, const MnkGeom& g, int B, int nb, void* obs, int obs_dtype,
                                              uint8_t* mask, int vec_ok, int tid, int nthreads) {
  __syncthreads();  // the stage is complete
  const bool ovec = vec_ok & 1, mvec = (vec_ok >> 1) & 1;
  if constexpr (CN != 0 && CN <= 31) {
    if (mnk_packed_form<CN>(g)) {
      // the pad segments that follow the last channel / legal segment: read (never used) by the last groups
      constexpr int SW = ((((32 * NW / (CN + 1)) * CN + 31) / 32) + 1) | 1;
      if (tid < SW) {
        s.segs[(size_t)(2 * B) * SW + tid] = 0u;
        s.segs[(size_t)(3 * B + 1) * SW + tid] = 0u;
      }
      for (int t = tid; t < 3 * B; t += nthreads) mnk_stage_squeeze<NW, CN>(s, B, t);
      __syncthreads();
      const uint32_t cells = (uint32_t)nb * 2u * (uint32_t)g.C;
      if (obs) {
        if (obs_dtype == MNK_OBS_F32) mnk_emit_packed<NW, CN, 4>(s.segs, g, cells, obs, ovec, tid, nthreads);
        else if (obs_dtype == MNK_OBS_BF16) mnk_emit_packed<NW, CN, 2>(s.segs, g, cells, obs, ovec, tid, nthreads);
        else mnk_emit_packed<NW, CN, 1>(s.segs, g, cells, obs, ovec, tid, nthreads);
      }
      if (mask) mnk_emit_packed<NW, CN, 1>(s.segs + (size_t)(2 * B + 1) * SW, g, (uint32_t)nb * (uint32_t)g.C, mask, mvec, tid, nthreads);
      return;
    }
  }
  if (obs) {
    const uint32_t row = 2u * (uint32_t)g.C;
    if (obs_dtype == MNK_OBS_F32) mnk_emit_table<4>(s, s.tab_obs, row, g.magic_2C, nb, obs, ovec, tid, nthreads);
    else if (obs_dtype == MNK_OBS_BF16) mnk_emit_table<2>(s, s.tab_obs, row, g.magic_2C, nb, obs, ovec, tid, nthreads);
    else mnk_emit_table<1>(s, s.tab_obs, row, g.magic_2C, nb, obs, ovec, tid, nthreads);
  }
  if (mask) mnk_emit_table<1>(s, s.tab_mask, (uint32_t)g.C, g.magic_C, nb, mask, mvec, tid, nthreads);
}

// the workgroup's slab of an observation array: row env0 of rows of 2C cells
__device__ __forceinline__ void* mnk_obs_slab(void* obs, int obs_dtype, int64_t row0, int C) {
  return obs ? (void*)((char*)obs + row0 * 2 * C * mnk_obs_bytes(obs_dtype)) : nullptr;
}

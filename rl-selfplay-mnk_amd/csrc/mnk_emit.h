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
};

// row stride of the stage in words: B + 1, not B.  With B = 64 = the number of LDS banks, word w of env e sat in bank
// e for every w, and the write-out -- where the lanes of a wave read a handful of DIFFERENT words of the SAME one or two
// envs -- serialised 5-way (9x9) to 8-way (19x19) on that one bank.  With the odd stride word w of env e sits in bank
// (w + e) mod 64: different words, different banks; lanes that want the same word are a broadcast.
__host__ __device__ inline int mnk_stage_stride(int B) { return B + 1; }

__host__ __device__ inline size_t mnk_stage_bytes(int NW, int C, int B) {
  return (size_t)3 * NW * mnk_stage_stride(B) * 4 + (size_t)3 * C * 4;
}

__device__ __forceinline__ MnkStage mnk_stage_carve(void* lds, const MnkGeom& g, int B) {
  MnkStage s;
  s.words = (uint32_t*)lds;
  s.tab_obs = s.words + (size_t)3 * g.NW * mnk_stage_stride(B);
  s.tab_mask = s.tab_obs + 2 * g.C;
  return s;
}

__device__ __forceinline__ void mnk_stage_tables(const MnkStage& s, const MnkGeom& g, int B, int tid, int nthreads) {
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

// obs slab of this workgroup: nb envs x 2C floats starting at dst (16-byte aligned when vec)
__device__ __forceinline__ void mnk_emit_obs(const MnkStage& s, const MnkGeom& g, int nb, float* dst, bool vec,
                                             int tid, int nthreads) {
  const uint32_t* st32 = s.words;
  const uint32_t row = 2u * (uint32_t)g.C;
  const uint32_t total = (uint32_t)nb * row;
  const uint32_t nvec = vec ? (total >> 2) : 0u;
  for (uint32_t q = tid; q < nvec; q += nthreads) {
    const uint32_t e = q << 2;
    uint32_t el = mnk_div(e, g.magic_2C);
    uint32_t rem = e - el * row;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (rem == row) { rem = 0; ++el; }
      v[j] = (float)mnk_stage_bit(st32, s.tab_obs[rem], el);
      ++rem;
    }
    reinterpret_cast<float4*>(dst)[q] = make_float4(v[0], v[1], v[2], v[3]);
  }
  for (uint32_t e = (nvec << 2) + tid; e < total; e += nthreads) {
    const uint32_t el = mnk_div(e, g.magic_2C);
    dst[e] = (float)mnk_stage_bit(st32, s.tab_obs[e - el * row], el);
  }
}

// mask slab: nb envs x C bytes starting at dst
__device__ __forceinline__ void mnk_emit_mask(const MnkStage& s, const MnkGeom& g, int nb, uint8_t* dst, bool vec,
                                              int tid, int nthreads) {
  const uint32_t* st32 = s.words;
  const uint32_t row = (uint32_t)g.C;
  const uint32_t total = (uint32_t)nb * row;
  const uint32_t nvec = vec ? (total >> 4) : 0u;
  for (uint32_t q = tid; q < nvec; q += nthreads) {
    const uint32_t e = q << 4;
    uint32_t el = mnk_div(e, g.magic_C);
    uint32_t rem = e - el * row;
    uint32_t out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t acc = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (rem == row) { rem = 0; ++el; }
        acc |= mnk_stage_bit(st32, s.tab_mask[rem], el) << (8 * b);
        ++rem;
      }
      out[j] = acc;
    }
    reinterpret_cast<uint4*>(dst)[q] = make_uint4(out[0], out[1], out[2], out[3]);
  }
  for (uint32_t e = (nvec << 4) + tid; e < total; e += nthreads) {
    const uint32_t el = mnk_div(e, g.magic_C);
    dst[e] = (uint8_t)mnk_stage_bit(st32, s.tab_mask[e - el * row], el);
  }
}

// mnk_selfplay_draw.h -- host side of the step kernels that draw their moves from logits (mnk_selfplay_*_logits of the
// C ABI).  One translation unit per entry point (mnk_selfplay_{pre,post,step}_logits.hip) so that the 15 kernel variants of
// each -- five boards x {f32, bf16, no logits} -- compile in parallel.
#pragma once
#include "mnk_selfplay_host.h"

inline int mnk_sample_args_ok(const MnkSample& sa, int64_t N, int C) {
  if (!sa.mask || !sa.actions || C < 1 || C > 1024 || N > 0x7fffffffLL) return MNK_EINVAL;
  if (sa.logits_dtype != MNK_LOGITS_F32 && sa.logits_dtype != MNK_LOGITS_BF16) return MNK_EINVAL;
  return MNK_OK;
}

// Launches kernel WHICH with the draw folded in when the board has a compile-time draw shape (3x3x3, 9x9x5, 13x13x5,
// 15x15x5, 19x19x5: the boards the reference trains on and the usual Gomoku sizes) or a run-time compiled variant of its own
// (mnk_jit.hip: any other board, any row width); false = the caller takes two launches.
template <int WHICH>
inline bool mnk_launch_sp_fused(const MnkSpArgs& a, const MnkSample& sa, hipStream_t s) {
  const MnkGeom& g = a.g;
  if (mnk_launch_sp_jit<WHICH>(a, nullptr, sa, s)) return true;  // a board without a built-in variant, once it is hot
  if (g.m != g.n) return false;
  const int lt = !sa.logits ? 2 : (sa.logits_dtype == MNK_LOGITS_BF16 ? 1 : 0);
#define MNK_FUSED(NWv, CNv, CKv, Cv)                                                                            \
  do {                                                                                                          \
    if (lt == 0) mnk_launch_sp<WHICH, NWv, CNv, CKv, Draw<float, Cv>>(a, nullptr, sa, s);                       \
    else if (lt == 1) mnk_launch_sp<WHICH, NWv, CNv, CKv, Draw<uint16_t, Cv>>(a, nullptr, sa, s);               \
    else mnk_launch_sp<WHICH, NWv, CNv, CKv, Draw<void, Cv>>(a, nullptr, sa, s);                                \
    return true;                                                                                                \
  } while (0)
  if (g.n == 9 && g.k == 5) MNK_FUSED(3, 9, 5, 81);
  if (g.n == 3 && g.k == 3) MNK_FUSED(1, 3, 3, 9);
  if (g.n == 13 && g.k == 5) MNK_FUSED(6, 13, 5, 169);
  if (g.n == 15 && g.k == 5) MNK_FUSED(8, 15, 5, 225);
  if (g.n == 19 && g.k == 5) MNK_FUSED(12, 19, 5, 361);
#undef MNK_FUSED
  return false;
}

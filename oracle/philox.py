"""numpy Philox4x32-10 and the samplers built on it (test infrastructure).

The reference draws its randomness from torch's global generator
(``torch.multinomial`` in ``/root/reference/src/selfplay/policy.py:29``,
``torch.randint`` in ``src/selfplay/torch_self_play_wrapper.py:26,43-45``); that
stream depends on call shapes and cannot be replayed on a GPU.  The HIP kernels
instead use the counter-based Philox4x32-10 generator (Salmon et al., "Parallel
random numbers: as easy as 1, 2, 3", SC'11) keyed so that every (seed, env, step,
stream) has its own value regardless of how envs are sharded over GPUs
(SURVEY.md §8d "Synthetic inputs").  This file restates that generator and the
selection rules in numpy so the kernels can be checked bit for bit; the
distribution itself (uniform over legal cells = ``RandomPolicy``) is pinned by the
known-answer statistics in BASELINE.md §2.

Counter layout (must match ``mnk_rng_block`` in ``csrc/mnk_device.h``):
    key = (seed_lo32, seed_hi32)
    ctr = (env_lo32, env_hi32, q_lo32, stream | (q_hi24 << 8))
    every stream: q = step >> 2, output word = step & 3
"""
import numpy as np

STREAM_MOVE = 0    # mover's action in the raw rollout / the agent's action
STREAM_OPP = 1     # opponent's action in the fused self-play step
STREAM_SIDE = 2    # side draw at (auto)reset: top bit of the word
STREAM_SAMPLE = 3  # the one uniform per row of the masked-logits sampler (inverse-CDF draw)

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """All arguments broadcastable uint32 arrays; returns four uint32 arrays."""
    c0, c1, c2, c3, k0, k1 = np.broadcast_arrays(
        *[np.asarray(v, dtype=np.uint64) & _MASK32 for v in (c0, c1, c2, c3, k0, k1)]
    )
    for rnd in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        if rnd != 9:
            k0 = (k0 + np.uint64(_W0)) & _MASK32
            k1 = (k1 + np.uint64(_W1)) & _MASK32
    return tuple(v.astype(np.uint32) for v in (c0, c1, c2, c3))


def _block(seed, env_ids, q, stream):
    env_ids = np.asarray(env_ids, dtype=np.uint64)
    q = np.asarray(q, dtype=np.uint64)
    seed = np.uint64(seed)
    c3 = np.uint64(stream) | (((q >> np.uint64(32)) & np.uint64(0xFFFFFF)) << np.uint64(8))
    return philox4x32_10(
        env_ids & _MASK32, env_ids >> np.uint64(32), q & _MASK32, c3,
        seed & _MASK32, seed >> np.uint64(32),
    )


def rand_u32(seed: int, env_ids, step, stream: int) -> np.ndarray:
    """One u32 per env for scalar streams (MOVE / OPP / SIDE)."""
    step = np.asarray(step, dtype=np.uint64)
    blk = _block(seed, env_ids, step >> np.uint64(2), stream)
    word = np.broadcast_to(step & np.uint64(3), blk[0].shape)
    return np.choose(word.astype(np.int64), blk)


def mulhi32(x, y) -> np.ndarray:
    return ((np.asarray(x, dtype=np.uint64) * np.asarray(y, dtype=np.uint64)) >> np.uint64(32)).astype(np.int64)


def pick_legal(mask: np.ndarray, x: np.ndarray) -> np.ndarray:
    """Uniform legal cell from one u32 per row.

    r = floor(x * L / 2^32) with L = number of legal cells; the action is the r-th
    legal cell in action order.  A row with no legal cell (full board) draws
    uniformly over all C cells, which is what ``RandomPolicy``'s 1e-8 guard
    amounts to (policy.py:21-24).
    """
    mask = np.asarray(mask).astype(bool)
    nenv, c = mask.shape
    nl = mask.sum(axis=1)
    r = mulhi32(x, np.where(nl > 0, nl, c))
    rank = np.cumsum(mask, axis=1) - 1
    hit = mask & (rank == r[:, None])
    act = np.argmax(hit, axis=1)
    return np.where(nl > 0, act, r).astype(np.int64)


def draw_side(x: np.ndarray) -> np.ndarray:
    """Fresh agent side from one u32: its top bit (0 = black, 1 = white)."""
    return (np.asarray(x, dtype=np.uint32) >> np.uint32(31)).astype(np.int64)


def uniform_open01(x: np.ndarray) -> np.ndarray:
    """u32 -> f32 uniform in (0, 1): (top 24 bits + 0.5) * 2^-24"""
    return ((np.asarray(x, dtype=np.uint32) >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(
        2.0 ** -24
    )

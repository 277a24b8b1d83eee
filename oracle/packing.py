"""numpy bit-plane packing in the guard-column layout of the HIP state (test infrastructure).

Layout (SURVEY.md §8b "device state layout"): a board plane of m rows x n columns
is stored as W = ceil(m*(n+1)/64) little-endian u64 words; cell (r, c) is bit
``r*(n+1) + c``.  Column n of every row is a guard column that is always 0, so a
shift by 1, n, n+1 or n+2 bits can never carry a stone across a board edge.
Arrays are structure-of-arrays over the env axis: ``planes[2][W][N]``.

Dense boards here are the reference's ``(N, 2, m, n)`` planes
(``/root/reference/src/env/torch_vector_mnk_env.py:17``); a cell counts as a
stone when it is non-zero, which is how ``observe`` reads it (env:47).
"""
import numpy as np


def words_per_plane(m: int, n: int) -> int:
    return (m * (n + 1) + 63) // 64


def cell_bit_index(m: int, n: int) -> np.ndarray:
    """bit index of action a = r*n + c, for every a in [0, m*n)"""
    a = np.arange(m * n)
    return a + a // n


def pack_cells(cells: np.ndarray, m: int, n: int) -> np.ndarray:
    """cells: (N, m*n) array, non-zero = set  ->  (W, N) u64"""
    cells = np.asarray(cells)
    nenv = cells.shape[0]
    w = words_per_plane(m, n)
    out = np.zeros((w, nenv), dtype=np.uint64)
    bits = cell_bit_index(m, n)
    on = cells != 0
    for a, b in enumerate(bits):
        out[b >> 6] |= on[:, a].astype(np.uint64) << np.uint64(b & 63)
    return out


def unpack_cells(words: np.ndarray, m: int, n: int) -> np.ndarray:
    """(W, N) u64 -> (N, m*n) u8 of 0/1"""
    words = np.asarray(words, dtype=np.uint64)
    nenv = words.shape[1]
    out = np.zeros((nenv, m * n), dtype=np.uint8)
    for a, b in enumerate(cell_bit_index(m, n)):
        out[:, a] = ((words[b >> 6] >> np.uint64(b & 63)) & np.uint64(1)).astype(np.uint8)
    return out


def pack_boards(boards, m: int, n: int) -> np.ndarray:
    """(N, 2, m, n) dense planes -> (2, W, N) u64"""
    b = np.asarray(boards)
    nenv = b.shape[0]
    return np.stack([pack_cells(b[:, p].reshape(nenv, m * n), m, n) for p in (0, 1)])


def unpack_boards(planes: np.ndarray, m: int, n: int) -> np.ndarray:
    """(2, W, N) u64 -> (N, 2, m, n) f32 of 0.0/1.0"""
    planes = np.asarray(planes, dtype=np.uint64)
    nenv = planes.shape[2]
    out = np.stack([unpack_cells(planes[p], m, n) for p in (0, 1)], axis=1)
    return out.reshape(nenv, 2, m, n).astype(np.float32)


def valid_cell_words(m: int, n: int) -> np.ndarray:
    """(W,) u64 with a 1 on every real cell and 0 on guard / padding bits"""
    w = words_per_plane(m, n)
    out = np.zeros(w, dtype=np.uint64)
    for b in cell_bit_index(m, n):
        out[b >> 6] |= np.uint64(1) << np.uint64(b & 63)
    return out


def record_words(m: int, n: int) -> int:
    """Rows of one rollout record: ceil(m*(n+1)/32) (include/mnk_hip.h, "Rollout record")."""
    return (m * (n + 1) + 31) // 32


def record_rows(planes: np.ndarray, m: int, n: int, side=None) -> np.ndarray:
    """State-layout planes u64[..., 2, W, N] -> record rows u64[..., R, N]: row w = 32-bit word w of the
    first plane | word w of the second plane << 32.  ``side`` (int array [..., N], 0 black / 1 white to move):
    the rollout records the MOVER's plane first, so white-to-move positions have their planes swapped."""
    planes = np.ascontiguousarray(planes, dtype=np.uint64)
    lead, (two, w, nenv) = planes.shape[:-3], planes.shape[-3:]
    assert two == 2 and w == words_per_plane(m, n)
    if side is not None:
        swap = np.asarray(side).astype(bool)[..., None, None, :]
        planes = np.where(swap, planes[..., ::-1, :, :], planes)
    r = record_words(m, n)
    lo = planes & np.uint64(0xFFFFFFFF)
    hi = planes >> np.uint64(32)
    halves = np.stack([lo, hi], axis=-2).reshape(lead + (2, 2 * w, nenv))[..., :r, :]  # [.., plane, word32, N]
    return halves[..., 0, :, :] | (halves[..., 1, :, :] << np.uint64(32))


def planes_from_record_rows(rows: np.ndarray, m: int, n: int, side=None) -> np.ndarray:
    """Inverse of ``record_rows``: u64[..., R, N] -> absolute planes u64[..., 2, W, N]."""
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    lead, (r, nenv) = rows.shape[:-2], rows.shape[-2:]
    w = words_per_plane(m, n)
    assert r == record_words(m, n)
    halves = np.zeros(lead + (2, 2 * w, nenv), dtype=np.uint64)
    halves[..., 0, :r, :] = rows & np.uint64(0xFFFFFFFF)
    halves[..., 1, :r, :] = rows >> np.uint64(32)
    pairs = halves.reshape(lead + (2, w, 2, nenv))
    planes = pairs[..., 0, :] | (pairs[..., 1, :] << np.uint64(32))
    if side is not None:
        swap = np.asarray(side).astype(bool)[..., None, None, :]
        planes = np.where(swap, planes[..., ::-1, :, :], planes)
    return planes

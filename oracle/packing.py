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

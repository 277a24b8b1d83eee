"""ctypes binding of ``libmnk_hip.so`` (C ABI in ``include/mnk_hip.h``).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There
is no fallback: if the shared object is missing or a call fails, an exception is
raised -- the product path never computes on the CPU.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MNK_HIP_LIB: another build of the same library (A/B experiments with compile-time options); default: the in-tree build
LIB_PATH = os.environ.get("MNK_HIP_LIB") or os.path.join(_HERE, "libmnk_hip.so")
ABI_VERSION = 6

MNK_OK = 0
ERR_NONE, ERR_ACTION_RANGE, ERR_ILLEGAL_MOVE = 0, 1, 2
STEP_STRICT, STEP_AUTORESET = 1, 2
LOGITS_F32, LOGITS_BF16 = 0, 1
OBS_F32, OBS_BF16, OBS_U8 = 0, 1, 2
ACT_U8, ACT_U16, ACT_BITS7, ACT_U8P1 = 1, 2, 3, 4
COMM_ID_BYTES = 128
SP_NEED_OPP, SP_WAS_RESET = 1, 2
STREAM_MOVE, STREAM_OPP, STREAM_SIDE, STREAM_SAMPLE = 0, 1, 2, 3
STATS_REPLICAS, STATS_STRIDE, STATS_COUNTERS = 64, 8, 5
# run-time specialised API kernels (MNK_JIT_API_* of include/mnk_hip.h): bit numbers for jit_prepare()
(JIT_API_STEP, JIT_API_STEP_DRAW, JIT_API_STEP_SUBSET, JIT_API_OBSERVE, JIT_API_SAMPLE_LEGAL, JIT_API_UNPACK_RECORDS,
 JIT_API_GATHER_OBS, JIT_API_SP_PRE, JIT_API_SP_POST, JIT_API_SP_STEP, JIT_API_SP_DRAW) = range(11)
JIT_API_COUNT = 19
REC_ACTION_MASK, REC_REWARD_SHIFT, REC_DONE_BIT, REC_SIDE_BIT = 0xFFFF, 16, 24, 25

_vp, _i, _i64, _u64, _u32, _f = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint32,
                                 ctypes.c_float)

# name -> argtypes; restype is int unless noted.  Kept in step with include/mnk_hip.h
# (tests/test_abi.py parses the header and compares).
SIGNATURES = {
    "mnk_abi_version": [],
    "mnk_reload_config": [],
    "mnk_state_words": [_i, _i],
    "mnk_record_words": [_i, _i],
    "mnk_geometry_supported": [_i, _i, _i],
    "mnk_last_launch_error": [],
    "mnk_reset_all": [_vp, _vp, _i64, _i, _vp],
    "mnk_reset_idx": [_vp, _vp, _i64, _i, _vp, _i64, _vp, _vp],
    "mnk_reset_mask": [_vp, _vp, _i64, _i, _vp, _vp],
    "mnk_step": [_vp, _vp, _i64, _i, _i, _i, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i, _vp, _u32, _vp],
    "mnk_step_random": [_vp, _vp, _i64, _i, _i, _i, _u64, _u64, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _i, _u32, _vp],
    "mnk_observe": [_vp, _vp, _i64, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp],
    "mnk_pack_boards": [_vp, _vp, _i64, _i, _i, _vp],
    "mnk_unpack_boards": [_vp, _vp, _i64, _i, _i, _vp],
    "mnk_sample_legal": [_vp, _i64, _i, _i, _u64, _u64, _vp, _i64, _i, _vp, _vp],
    "mnk_sample_logits": [_vp, _i, _vp, _i64, _i, _u64, _vp, _u64, _vp, _i64, _i, _vp, _vp, _vp],
    "mnk_selfplay_pre": [_vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _u64, _u64, _vp, _i64, _vp, _vp, _vp, _vp,
                         _i, _vp, _vp, _u32, _vp],
    "mnk_selfplay_post": [_vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp,
                          _vp, _u32, _vp],
    "mnk_selfplay_step_random": [_vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _u64, _u64, _vp, _i64, _vp, _vp,
                                 _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    # the sampler block of the *_logits forms: logits, dtype, mask, seed, seed_dev, step, step_dev, env_id0, deterministic,
    # actions (out), logp (out)
    "mnk_selfplay_pre_logits": [_vp, _vp, _i64, _i, _i, _i] + [_vp, _i, _vp, _u64, _vp, _u64, _vp, _i64, _i, _vp, _vp] +
                               [_vp, _vp, _vp, _u64, _u64, _vp, _i64, _vp, _vp, _vp, _vp, _i, _vp, _vp, _u32, _vp],
    "mnk_selfplay_post_logits": [_vp, _vp, _i64, _i, _i, _i] + [_vp, _i, _vp, _u64, _vp, _u64, _vp, _i64, _i, _vp, _vp] +
                                [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    "mnk_selfplay_step_random_logits": [_vp, _vp, _i64, _i, _i, _i] + [_vp, _i, _vp, _u64, _vp, _u64, _vp, _i64, _i, _vp, _vp] +
                                       [_vp, _vp, _vp, _u64, _u64, _vp, _i64, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                        _u32, _vp],
    "mnk_rollout_random": [_vp, _vp, _i64, _i, _i, _i, _i, _u64, _u64, _i64, _vp, _vp, _vp, _vp, _i, _vp],
    "mnk_action_log_words": [_i, _i],
    "mnk_replay_actions": [_vp, _vp, _i64, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp],
    "mnk_unpack_records": [_vp, _vp, _i64, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "mnk_gather_obs": [_vp, _i64, _i64, _i, _i, _vp, _i64, _vp, _i, _vp, _i, _vp, _vp],
    "mnk_gae": [_vp, _vp, _vp, _vp, _i64, _i, _f, _f, _vp, _vp, _vp],
    "mnk_jit_compile_rollout": [_i, _i, _i, _i, _i],
    "mnk_jit_compile_kernel": [_i, _i, _i, _i, _i, _i],
    "mnk_jit_last_error": [],
    "mnk_jit_compile_api": [_i, _i, _i, _i],
    "mnk_jit_prepare": [_i, _i, _i, _i64],
    "mnk_jit_api_ready": [_i, _i, _i, _i],
    "mnk_jit_stats": [_vp],
    "mnk_comm_unique_id": [_vp],
    "mnk_comm_init": [_vp, _vp, _i, _i],
    "mnk_comm_destroy": [_vp],
    "mnk_allgather_records": [_vp, _vp, _vp, _i64, _vp],
    "mnk_allgather_records_direct": [_vp, _vp, _vp, _i64, _vp],
    "mnk_comm_last_error": [],
    "mnk_comm_version": [],
    "mnk_probe_record_writes": [_vp, _i64, _i, _i, _vp],
}

_STATUS = {-1: "invalid argument (null pointer / negative size)", -2: "unsupported board geometry",
           -3: "kernel launch failed", -4: "RCCL call failed"}


class MnkHipError(RuntimeError):
    pass


_lib = None


def load():
    """Returns the loaded library; raises MnkHipError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MnkHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for this path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)  # AttributeError here = header / library mismatch
        except AttributeError:
            if os.environ.get("MNK_HIP_LIB"):  # an older build loaded for an A/B run: entry points added since are absent
                continue
            raise
        fn.argtypes = argtypes
        if name in ("mnk_last_launch_error", "mnk_comm_last_error", "mnk_jit_last_error"):
            fn.restype = ctypes.c_char_p
        elif name in ("mnk_jit_compile_rollout", "mnk_jit_compile_kernel", "mnk_jit_compile_api"):
            fn.restype = ctypes.c_int64
        else:
            fn.restype = ctypes.c_int
    if lib.mnk_abi_version() != ABI_VERSION:
        raise MnkHipError(f"libmnk_hip.so ABI {lib.mnk_abi_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a contiguous tensor (or NULL for None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "mnk_hip: tensors handed to the C ABI must be contiguous"
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device):
    """hipStream_t of torch's current stream on ``device`` -- through the raw accessor where this torch has it (one C
    call, ~0.3 us) instead of building a ``torch.cuda.Stream`` object per launch (~4 us of the ~18 us a step costs on
    the host)"""
    if _raw_stream is not None:
        idx = device.index if isinstance(device, torch.device) else torch.device(device).index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


_bound = {}


def call(name, *args):
    fn = _bound.get(name)
    if fn is None:
        fn = _bound[name] = getattr(load(), name)
    lib = _lib
    rc = fn(*args)
    if rc != MNK_OK:
        detail = _STATUS.get(rc, f"status {rc}")
        if rc == -3:
            detail += ": " + (lib.mnk_last_launch_error() or b"").decode()
        if rc == -4:
            detail += ": " + (lib.mnk_comm_last_error() or b"").decode()
        raise MnkHipError(f"{name}: {detail}")
    return rc


_OBS_DTYPES = {torch.float32: OBS_F32, torch.bfloat16: OBS_BF16, torch.uint8: OBS_U8}


def obs_dtype_code(dtype) -> int:
    """MNK_OBS_* code of a torch dtype an observation may be written in (float32 = the reference's; bfloat16 and
    uint8 are the opt-in narrow forms: cells are exactly 0 / 1, so ``obs.float()`` is the same tensor)."""
    try:
        return _OBS_DTYPES[dtype]
    except KeyError:
        raise TypeError(f"observations can be written as float32, bfloat16 or uint8, not {dtype}") from None


def obs_code(t) -> int:
    """MNK_OBS_* code of an observation tensor (None -> F32: the pointer is NULL and the code is not looked at)"""
    return OBS_F32 if t is None else obs_dtype_code(t.dtype)


def reload_config() -> None:
    """The library reads its developer knobs (MNK_ROLLOUT_PAIR / _FORM / _SADDR, MNK_JIT, MNK_EMIT_ENVS / _THREADS) from
    the environment once; call this after changing ``os.environ`` to make it read them again."""
    call("mnk_reload_config")


def jit_api_draw_kind(which: int, logits_dtype=None) -> int:
    """MNK_JIT_API_* number of a self-play step kernel with the draw folded in: ``which`` 0 pre / 1 post / 2
    step_random; ``logits_dtype`` torch.float32, torch.bfloat16 or None (no logits: uniform over the mask)"""
    lt = 2 if logits_dtype is None else (1 if logits_dtype == torch.bfloat16 else 0)
    return JIT_API_SP_DRAW + 3 * lt + which


def jit_prepare(m: int, n: int, k: int, kinds=None) -> int:
    """Compile and load NOW the board's own variants of the API kernels in ``kinds`` (iterable of JIT_API_* numbers;
    None: of every kernel launched on this board so far) instead of when they get hot -- before a hipGraph capture,
    where nothing can be compiled: warm up, ``jit_prepare(m, n, k)``, capture.  Returns how many are ready; 0 for
    boards with a built-in variant (3x3x3, 9x9x5, 13x13x5, 15x15x5, 19x19x5) or with MNK_JIT_API / MNK_JIT = 0."""
    mask = 0
    for kind in (kinds if kinds is not None else ()):
        mask |= 1 << int(kind)
    if kinds is not None and mask == 0:
        return 0
    lib = load()
    rc = lib.mnk_jit_prepare(m, n, k, mask)
    if rc < 0:
        raise MnkHipError(f"mnk_jit_prepare({m}, {n}, {k}): {_STATUS.get(rc, rc)}: " + (lib.mnk_jit_last_error() or b"").decode())
    return rc


def jit_stats() -> dict:
    """what the run-time compiler did in this process: programs compiled by hiprtc, code objects read from the cache on
    disk instead (``$MNK_JIT_CACHE``, default ``~/.cache/mnk_hip``; "0": off), written to it, failed compilations"""
    out = (ctypes.c_int64 * 4)()
    load().mnk_jit_stats(out)
    return dict(zip(("compiled", "cache_hits", "cache_stores", "failed"), (int(v) for v in out)))


def jit_api_ready(m: int, n: int, k: int, kind: int) -> bool:
    """is the board's own variant of API kernel ``kind`` (JIT_API_*) loaded on the current device?"""
    return bool(load().mnk_jit_api_ready(m, n, k, int(kind)))


def state_words(m, n):
    return load().mnk_state_words(m, n)


def record_words(m: int, n: int) -> int:
    return load().mnk_record_words(m, n)


def geometry_supported(m, n, k):
    return bool(load().mnk_geometry_supported(m, n, k))

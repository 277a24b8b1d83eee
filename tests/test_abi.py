"""CPU: the C-ABI library builds, loads, and exports exactly what include/mnk_hip.h declares.
No compute calls here (no GPU in the build container)."""
import os
import re

import pytest

import __graft_entry__ as entry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mnk_hip.h")


@pytest.fixture(scope="module")
def lib():
    entry.build_hip()
    entry._ensure_path()
    import mnk_hip

    return mnk_hip


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = re.findall(r"\b(?:int|int64_t|const char\*)\s+(mnk_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S)
    out = {}
    for name, args in decls:
        args = args.strip()
        out[name] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return out


def test_header_declares_the_path():
    decl = declared_functions()
    for name in ("mnk_step", "mnk_observe", "mnk_reset_all", "mnk_reset_idx", "mnk_sample_legal",
                 "mnk_rollout_random", "mnk_replay_actions", "mnk_selfplay_pre", "mnk_selfplay_post", "mnk_sample_logits",
                 "mnk_pack_boards", "mnk_unpack_boards", "mnk_unpack_records", "mnk_gather_obs", "mnk_gae",
                 "mnk_step_random", "mnk_selfplay_step_random", "mnk_selfplay_pre_logits", "mnk_selfplay_post_logits",
                 "mnk_selfplay_step_random_logits"):
        assert name in decl


def test_library_exports_every_declared_symbol(lib):
    handle = lib.load()
    decl = declared_functions()
    assert set(decl) == set(lib.SIGNATURES), "binding and header disagree on the function list"
    for name, nargs in decl.items():
        assert hasattr(handle, name), f"{name} declared in mnk_hip.h but not exported"
        assert len(lib.SIGNATURES[name]) == nargs, f"{name}: binding has {len(lib.SIGNATURES[name])} args, header {nargs}"
    assert handle.mnk_abi_version() == lib.ABI_VERSION


def test_geometry_queries_need_no_gpu(lib):
    assert [lib.state_words(*s) for s in [(3, 3), (9, 9), (13, 13), (19, 19)]] == [1, 2, 3, 6]
    assert lib.geometry_supported(9, 9, 5) and lib.geometry_supported(19, 19, 5) and lib.geometry_supported(22, 22, 5)
    assert not lib.geometry_supported(3, 3, 4)      # k larger than the board
    assert lib.geometry_supported(25, 25, 5) and lib.geometry_supported(31, 31, 6) and lib.geometry_supported(16, 61, 5)
    assert lib.state_words(25, 25) == 11 and lib.record_words(25, 25) == 21 and lib.state_words(31, 31) == 16
    assert not lib.geometry_supported(32, 32, 5)    # beyond 1 024 bits per plane
    assert not lib.geometry_supported(9, 62, 5)     # rows of more than 61 cells
    assert lib.state_words(32, 32) == 0


def test_argument_errors_are_reported_not_crashed(lib):
    # null state pointers are rejected on the host before anything is enqueued
    with pytest.raises(lib.MnkHipError):
        lib.call("mnk_reset_all", None, None, 16, 2, None)
    with pytest.raises(lib.MnkHipError):
        lib.call("mnk_step", None, None, 16, 9, 9, 5, None, None, 16, None, None, None, None, 0, None, 0, None)
    with pytest.raises(lib.MnkHipError):
        lib.call("mnk_rollout_random", None, None, 16, 40, 40, 5, 4, 0, 0, 0, None, None, None, None, 0, None)


def test_product_path_has_no_cpu_mode(lib):
    from env.torch_vector_mnk_env import TorchVectorMnkEnv

    with pytest.raises(RuntimeError, match="no CPU mode"):
        TorchVectorMnkEnv(3, 3, 3, 4, device="cpu")
    with pytest.raises(AssertionError):
        TorchVectorMnkEnv(3, 3, 4, 4, device="cpu")  # reference env:9


def test_rollout_kernel_specialises_at_run_time_without_a_gpu(lib):
    """hiprtc compiles the rollout kernel for boards that have no ahead-of-time specialisation (csrc/mnk_jit.hip);
    compiling needs no GPU, so the build container checks that the embedded headers still compile for gfx950."""
    handle = lib.load()
    for (m, n, k, rec, act) in [(12, 12, 5, 1, 0), (7, 9, 7, 0, 1), (22, 22, 10, 1, 2), (11, 11, 5, 1, 3), (22, 22, 10, 1, 4)]:
        size = handle.mnk_jit_compile_rollout(m, n, k, rec, act)
        assert size > 4096, (handle.mnk_jit_last_error() or b"").decode()
    assert handle.mnk_jit_compile_rollout(40, 40, 5, 1, 0) == -2   # MNK_EGEOM: beyond the packed layout
    assert handle.mnk_jit_compile_rollout(13, 13, 5, 1, 3) == -1   # MNK_EINVAL: 169 cells do not fit the 7-bit log
    assert handle.mnk_jit_compile_rollout(9, 9, 5, 1, 4) == -1     # MNK_EINVAL: the byte + bit log is for boards above 256 cells
    assert handle.mnk_jit_compile_rollout(9, 9, 5, 1, 5) == -1     # MNK_EINVAL: no such log format
    assert [handle.mnk_action_log_words(f, 256) for f in (1, 2, 3, 4, 5)] == [64, 128, 56, 72, 0]
    # round 4: the replay kernel and the two-lanes-per-env rollout exist as run-time specialisations too, and boards of
    # more than 512 bits per plane (25x25, 31x31, rows of 61 cells) have no other rollout / replay kernel
    ROLLOUT, REPLAY, PAIR = 0, 1, 2
    for (m, n, k, rec, act, kind) in [(12, 12, 5, 1, 0, PAIR), (7, 9, 7, 1, 1, PAIR), (12, 12, 5, 1, 1, REPLAY),
                                      (25, 25, 5, 1, 2, ROLLOUT), (25, 25, 5, 1, 2, REPLAY), (31, 31, 6, 0, 2, REPLAY),
                                      (22, 23, 10, 1, 4, REPLAY), (16, 61, 5, 1, 0, PAIR)]:
        size = handle.mnk_jit_compile_kernel(m, n, k, rec, act, kind)
        assert size > 4096, (m, n, k, kind, (handle.mnk_jit_last_error() or b"").decode())
    assert handle.mnk_jit_compile_kernel(12, 12, 5, 1, 0, REPLAY) == -1  # a replay reads a log
    assert handle.mnk_jit_compile_kernel(11, 11, 5, 1, 3, PAIR) == -1    # the 7-bit stream exists in the one-lane form only
    assert handle.mnk_jit_compile_kernel(25, 25, 5, 1, 4, ROLLOUT) == -1 # 625 cells: two bytes per action
    assert handle.mnk_jit_compile_kernel(12, 12, 5, 1, 0, 3) == -1


def test_api_kernels_specialise_at_run_time_without_a_gpu(lib):
    """ABI 6: hiprtc instantiates the API-level kernel templates (csrc/mnk_api_kernels.h, mnk_selfplay_kernels.h: the text
    hipcc compiles, embedded in the library) with a board's own geometry -- step, observe, the records / gather write-outs,
    the three self-play step kernels with and without the folded-in draw (any row width).  Compiling needs no GPU."""
    handle = lib.load()
    boards = {(12, 12, 5): range(lib.JIT_API_COUNT),           # every kind once
              (6, 7, 4): (lib.JIT_API_STEP, lib.JIT_API_SP_STEP, lib.jit_api_draw_kind(1, None)),   # 2 words per plane
              (5, 5, 4): (lib.JIT_API_STEP_DRAW, lib.JIT_API_OBSERVE),                              # 1 word per plane
              (10, 33, 5): (lib.JIT_API_SP_POST, lib.JIT_API_UNPACK_RECORDS),    # rows of more than 31 cells: table write-out
              (25, 25, 5): (lib.JIT_API_SP_STEP, lib.jit_api_draw_kind(2, __import__("torch").bfloat16))}  # 21 words, 625 cells
    for (m, n, k), kinds in boards.items():
        for kind in kinds:
            size = handle.mnk_jit_compile_api(m, n, k, kind)
            assert size > 4096, (m, n, k, kind, (handle.mnk_jit_last_error() or b"").decode())
    assert handle.mnk_jit_compile_api(12, 12, 5, lib.JIT_API_COUNT) == -1   # no such kernel
    assert handle.mnk_jit_compile_api(12, 12, 5, -1) == -1
    assert handle.mnk_jit_compile_api(40, 40, 5, 0) == -2                   # beyond the packed layout


def test_compiled_code_objects_are_reused_from_disk(lib, tmp_path, monkeypatch):
    """$MNK_JIT_CACHE: a code object hiprtc produced is written to disk (by rename, checksummed, keyed by the embedded
    sources + hiprtc version + options + kernel) and the next compilation of the same kernel -- in this or a later
    process -- reads it back instead of compiling; a damaged file is ignored and replaced; "0" switches the cache off."""
    import time

    handle = lib.load()
    cache = tmp_path / "jit"
    monkeypatch.setenv("MNK_JIT_CACHE", str(cache))
    s0 = lib.jit_stats()
    t0 = time.time()
    size = handle.mnk_jit_compile_api(9, 10, 4, lib.JIT_API_SP_STEP)
    cold = time.time() - t0
    assert size > 4096
    files = sorted(cache.glob("*.co"))
    s1 = lib.jit_stats()
    assert len(files) == 1 and s1["compiled"] == s0["compiled"] + 1 and s1["cache_stores"] == s0["cache_stores"] + 1
    t0 = time.time()
    assert handle.mnk_jit_compile_api(9, 10, 4, lib.JIT_API_SP_STEP) == size
    warm = time.time() - t0
    s2 = lib.jit_stats()
    assert s2["cache_hits"] == s1["cache_hits"] + 1 and s2["compiled"] == s1["compiled"] and warm < cold / 4, (cold, warm)
    # the rollout kernel's programs go through the same cache; another kernel = another file
    assert handle.mnk_jit_compile_rollout(9, 10, 4, 1, 0) > 4096 and handle.mnk_jit_compile_rollout(9, 10, 4, 1, 0) > 4096
    assert len(list(cache.glob("*.co"))) == 2 and lib.jit_stats()["cache_hits"] == s2["cache_hits"] + 1
    # a damaged file (truncated; a flipped byte) is not trusted
    blob = files[0].read_bytes()
    files[0].write_bytes(blob[: len(blob) // 2])
    assert handle.mnk_jit_compile_api(9, 10, 4, lib.JIT_API_SP_STEP) == size
    assert lib.jit_stats()["compiled"] == s2["compiled"] + 2 and files[0].read_bytes() == blob   # (+1: the rollout above)
    files[0].write_bytes(blob[:-1] + bytes([blob[-1] ^ 1]))
    assert handle.mnk_jit_compile_api(9, 10, 4, lib.JIT_API_SP_STEP) == size and files[0].read_bytes() == blob
    assert not list(cache.glob(".tmp*"))
    monkeypatch.setenv("MNK_JIT_CACHE", "0")
    before = lib.jit_stats()
    assert handle.mnk_jit_compile_api(9, 10, 4, lib.JIT_API_SP_STEP) == size
    after = lib.jit_stats()
    assert after["compiled"] == before["compiled"] + 1 and after["cache_hits"] == before["cache_hits"]


def test_header_is_plain_c(tmp_path):
    """include/mnk_hip.h is a C ABI: it must compile as C99 (a cgo / JNI / plain C host includes it as is) and as C++."""
    import subprocess

    src = tmp_path / "hdr.c"
    src.write_text('#include "mnk_hip.h"\nint main(void) { return MNK_ABI_VERSION == 6 ? 0 : 1; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(src)], check=True)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", "-x", "c++", str(src)], check=True)


def test_a_plain_c_host_links_against_the_library():
    """examples/c_host.c (C99 + the HIP runtime's C API, no Python / torch / C++) builds against include/mnk_hip.h and
    libmnk_hip.so; its ``--abi`` mode calls only the entry points that touch no GPU.  The GPU suite runs the rest
    (tests/test_gpu_c_host.py)."""
    import subprocess

    import __graft_entry__ as entry

    entry.build_hip()
    exe = entry.build_c_host()
    res = subprocess.run([exe, "--abi"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    got = dict(kv.split("=") for kv in res.stdout.split())
    assert got == {"abi": "6", "header_abi": "6", "words_9x9": "2", "record_words_9x9": "3", "words_19x19": "6",
                   "supported_9x9x5": "1", "supported_2x2x3": "0"}

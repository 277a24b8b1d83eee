"""Host-side sanitizer pass over the C ABI (build container only; SURVEY.md section 5 lists ASan / race detection among the
reference-side practices).

    python tools/sanitize_host.py            # builds libmnk_hip_asan.so, runs the CPU tests of the ABI against it

What runs instrumented: the HOST code of every translation unit -- argument checks of all entry points, geometry
construction, the config cache, the hiprtc compile path and its caches, the dlopen'ed RCCL table, error-string
bookkeeping -- under AddressSanitizer + UndefinedBehaviorSanitizer.  The device code is compiled as always
(`-fno-gpu-sanitize`): GPU ASan needs xnack+ code objects, which the GPU pool refuses.  No GPU is touched: the tests are
tests/test_abi.py (symbols, argument errors, the hiprtc compile of the embedded headers, a C host's --abi mode),
tests/test_distributed_cpu.py (gloo ranks, each loading the library) and tests/test_bench_robustness.py.
Python itself is not instrumented, so ASan's runtime is LD_PRELOADed; leak detection is off (CPython never frees its
arenas).  Exit code = pytest's; any ASan / UBSan report aborts the test process (halt_on_error)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def main():
    lib = entry.build_hip(sanitize=True)
    env = dict(os.environ, MNK_HIP_LIB=lib, LD_PRELOAD=entry.sanitizer_runtime(),
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1:detect_stack_use_after_return=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", MNK_SANITIZED="1")
    tests = sys.argv[1:] or ["tests/test_abi.py", "tests/test_distributed_cpu.py", "tests/test_bench_robustness.py"]
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider"] + tests
    print("[sanitize_host]", " ".join(cmd), f"(MNK_HIP_LIB={lib})", flush=True)
    return subprocess.call(cmd, cwd=ROOT, env=env)


if __name__ == "__main__":
    sys.exit(main())

"""CPU: the parts of bench.py that decide how a multi-GPU run ENDS -- the launcher that waits for the ranks, the ranks'
common verdict about a hung exchange form, the watchdog of a phase that never finishes.  The driver takes a run's exit
code at face value: a hang or a dead rank must come back non-zero, and in bounded time."""
import os
import subprocess
import sys
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_CHILD = r"""
import os, sys, time
if os.environ["RANK"] == "0":
    sys.exit(3)            # what a rank does whose C-ABI communicator could not be created (bench.py EXIT_COMM)
time.sleep(120)            # a sibling that sits in a collective its dead peer will never join
"""


def test_a_dead_rank_takes_the_launcher_down_within_the_grace_period():
    import bench

    t0 = time.monotonic()
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(2, argv=[sys.executable, "-c", _CHILD], grace_s=1.0)
    took = time.monotonic() - t0
    assert took < 30, f"the launcher waited {took:.0f} s for a rank that sleeps 120 s"
    text = str(e.value.code)
    assert "rank exit codes" in text and "[3, -15]" in text, text  # rank 0 left with 3, rank 1 was terminated


def test_all_ranks_finishing_is_a_clean_exit_and_a_late_failure_is_still_seen():
    import bench

    bench.spawn_ranks(3, argv=[sys.executable, "-c", "import time; time.sleep(0.2)"], grace_s=1.0)  # no SystemExit
    late = "import os, sys, time; time.sleep(0.5 * int(os.environ['RANK'])); sys.exit(7 if os.environ['RANK'] == '2' else 0)"
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(3, argv=[sys.executable, "-c", late], grace_s=1.0)
    assert "[0, 0, 7]" in str(e.value.code)


def test_a_rank_that_ignores_sigterm_is_killed():
    import bench

    stubborn = ("import os, signal, sys, time\n"
                "if os.environ['RANK'] == '0': sys.exit(EXIT)\n"
                "signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
                "time.sleep(120)\n").replace("EXIT", str(bench.EXIT_HUNG))
    procs = []
    for rank in range(2):
        procs.append(subprocess.Popen([sys.executable, "-c", stubborn], env=dict(os.environ, RANK=str(rank))))
    time.sleep(1.0)  # let rank 1 install its handler
    t0 = time.monotonic()
    codes = bench.wait_ranks(procs, grace_s=0.5)
    assert codes == [bench.EXIT_HUNG, -9] and time.monotonic() - t0 < 30, codes


def test_the_ranks_agree_on_a_hang_without_a_gpu_collective():
    """``agree_any``: every rank posts its verdict in the process group's store; any "hung" makes it "hung" for all, and
    a rank that never posts counts as hung once the wait runs out -- no rank is left waiting in an all-reduce."""
    import torch.distributed as dist

    import bench

    store = dist.HashStore()
    got = {}

    def rank_fn(rank, mine, tag, world=3):
        got[(tag, rank)] = bench.agree_any(store, tag, rank, world, mine, wait_s=20.0)

    for tag, verdicts, want in (("a", (False, False, False), False), ("b", (False, True, False), True)):
        threads = [threading.Thread(target=rank_fn, args=(r, v, tag)) for r, v in enumerate(verdicts)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(60)
        assert [got[(tag, r)] for r in range(3)] == [want] * 3
    # rank 2 never reports: the others stop waiting for it and take the form to have hung
    t0 = time.monotonic()
    threads = [threading.Thread(target=lambda r=r: got.__setitem__(("c", r), bench.agree_any(store, "c", r, 3, False, 1.0)))
               for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert got[("c", 0)] is True and got[("c", 1)] is True and time.monotonic() - t0 < 30


def test_a_phase_that_never_finishes_ends_the_rank_with_its_own_exit_code():
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "seen = []\n"
            "w = bench.PhaseWatchdog(1.0, 3, lambda phase: print('PARTIAL', phase, flush=True))\n"
            "w.enter('warm-up'); time.sleep(0.3); w.enter('timed region'); time.sleep(60)\n") % ROOT
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    import bench

    assert out.returncode == bench.EXIT_PHASE and time.monotonic() - t0 < 60
    assert "phase 'timed region'" in out.stderr and "rank 3" in out.stderr
    assert "PARTIAL timed region" in out.stdout  # rank 0's hook: the line as far as it was assembled


def test_a_stopped_or_disabled_watchdog_does_nothing():
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "w = bench.PhaseWatchdog(0.5, 0); w.enter('x'); w.stop(); time.sleep(2.5)\n"
            "w = bench.PhaseWatchdog(0.0, 0); w.enter('y'); time.sleep(1.5); print('ALIVE')\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ALIVE" in out.stdout, out.stderr[-2000:]


def test_exchange_every_must_divide_the_step_counts():
    for argv, ok in ((["--exchange-every", "2", "--steps", "6", "--warmup", "2", "--settle", "8"], True),
                     (["--exchange-every", "4", "--steps", "6"], False), (["--exchange-every", "0"], False)):
        out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; sys.argv[1:] = %r; "
                              "a = bench.parse(); print('EVERY', a.exchange_every)" % (ROOT, argv)], capture_output=True, text=True,
                             timeout=120)
        assert (out.returncode == 0) == ok, out.stderr[-1500:]
        if ok:
            assert "EVERY 2" in out.stdout

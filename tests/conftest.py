import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rl-selfplay-mnk_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# run-time compiled kernels are not taken from (or left in) the user's cache on disk: the suite is also the check that the
# embedded sources still compile, and must not depend on what an earlier run left behind
os.environ.setdefault("MNK_JIT_CACHE", "0")

# the oracle is torch-eager on small batches: a GPU box's 256 logical CPUs make every tiny op slower, not faster
try:
    import torch

    torch.set_num_threads(min(8, os.cpu_count() or 1))
except ImportError:  # pragma: no cover
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def random_play_stats(board: str) -> dict:
    """Known-answer statistics of uniform random play on ``board`` ("9x9x5"), recorded from the imported reference by
    tests/golden/make_golden_stats.py (first games only, >= 3e5 of them): games, mean_plies, sd_plies, se_mean_plies,
    draws, draw_rate, se_draw_rate, black_wins, white_wins."""
    import numpy as np

    data = np.load(os.path.join(GOLDEN, "random_play_stats.npz"))
    return dict(zip((str(f) for f in data["fields"]), (float(v) for v in data[board])))

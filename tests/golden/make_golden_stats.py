"""Known-answer statistics of uniform random play, recorded from the imported reference
(build container only; ``/root/reference`` is mounted there):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_stats.py [games-per-board]

SURVEY.md section 8c, vector G5: the HIP rollout draws its moves from Philox, the reference from
``torch.multinomial`` on the global generator, so a bitwise replay of an RNG-driven rollout is impossible; what
both must agree on is the DISTRIBUTION of uniform random play.  This script plays FIRST games only -- every env
plays exactly one game from the empty board with the reference's own ``TorchVectorMnkEnv``
(src/env/torch_vector_mnk_env.py:34-119) and ``RandomPolicy`` (src/selfplay/policy.py:13-29, multinomial over the
legal mask) under a fixed ``torch.manual_seed`` -- and stores, per board: games, mean / standard deviation of the
game length in plies, draws, black wins, white wins, and the standard errors of the mean length and the draw rate.
(BASELINE.md section 2's 53.3 plies / 0.26 % draws at 9x9x5 came from a fixed window of steps after a common start,
which under-counts long games; first-game statistics have no such bias.)

Data only: ``random_play_stats.npz`` holds numbers, nothing of the reference's text.  The GPU tests read their
tolerances from it (``tests/test_gpu_env.py``).
"""
import os
import sys
import time

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
from env.torch_vector_mnk_env import TorchVectorMnkEnv  # noqa: E402
from selfplay.policy import RandomPolicy  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
BOARDS = [(3, 3, 3), (9, 9, 5), (12, 12, 5), (19, 19, 5)]
SEED = 20260104
BATCH = 16384


def first_games(m, n, k, games, seed):
    """lengths int32[games], outcome int8[games] (0 draw, 1 black wins, 2 white wins)"""
    torch.manual_seed(seed)
    lengths, outcomes = [], []
    pol = RandomPolicy(m * n)
    left = games
    while left > 0:
        nenv = min(BATCH, left)
        env = TorchVectorMnkEnv(m, n, k, nenv, device="cpu")
        obs = env.reset()
        alive = torch.arange(nenv)
        length = torch.zeros(nenv, dtype=torch.int32)
        outcome = torch.zeros(nenv, dtype=torch.int8)
        while alive.numel():
            sub = {"observation": obs["observation"][alive], "action_mask": obs["action_mask"][alive]}
            acts = pol.act(sub)
            mover = env.current_player[alive].clone()  # 0 black, 1 white (before the ply)
            obs, rew, done = env.step_subset(acts, alive)
            fin = done[alive]
            if bool(fin.any()):
                idx = alive[fin]
                length[idx] = env.move_counts[idx].to(torch.int32)
                won = rew[idx] > 0
                outcome[idx] = torch.where(won, (mover[fin] + 1).to(torch.int8), torch.zeros((), dtype=torch.int8))
                alive = alive[~fin]
        lengths.append(length.numpy())
        outcomes.append(outcome.numpy())
        left -= nenv
    return np.concatenate(lengths), np.concatenate(outcomes)


def main():
    games = int(sys.argv[1]) if len(sys.argv) > 1 else 327680
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    out = {"seed": np.int64(SEED), "boards": np.array(BOARDS, dtype=np.int32)}
    for (m, n, k) in BOARDS:
        t0 = time.time()
        length, outcome = first_games(m, n, k, games, SEED + m * 10000 + n * 100 + k)  # a seed per board, not per position in the list
        g = length.size
        mean, sd = float(length.mean()), float(length.std(ddof=1))
        draws, black, white = (int((outcome == v).sum()) for v in (0, 1, 2))
        p = draws / g
        key = f"{m}x{n}x{k}"
        out[key] = np.array([g, mean, sd, sd / np.sqrt(g), draws, p, np.sqrt(max(p * (1 - p), 1.0 / g) / g), black, white],
                            dtype=np.float64)
        out[key + "_length_histogram"] = np.bincount(length, minlength=m * n + 1).astype(np.int64)
        print(f"{key}: {g} games, mean {mean:.4f} +- {sd / np.sqrt(g):.4f} plies (sd {sd:.3f}), draws {p * 100:.4f} %, "
              f"black {black / g * 100:.3f} % white {white / g * 100:.3f} %  [{time.time() - t0:.0f} s]", flush=True)
    out["fields"] = np.array(["games", "mean_plies", "sd_plies", "se_mean_plies", "draws", "draw_rate", "se_draw_rate",
                              "black_wins", "white_wins"])
    np.savez_compressed(os.path.join(OUT, "random_play_stats.npz"), **out)


if __name__ == "__main__":
    main()

"""Random-policy rollout on the MI355X: T plies per launch, packed records, optional unpack.

    python examples/random_rollout.py --board 9x9x5 --envs 65536 --plies 256
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]

import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", default="9x9x5")
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--plies", type=int, default=256)
    args = ap.parse_args()
    entry.build()
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay.random_rollout import RandomRollout, unpack_records

    m, n, k = (int(v) for v in args.board.split("x"))
    env = TorchVectorMnkEnv(m, n, k, args.envs, device="cuda")
    roll = RandomRollout(env, seed=0)
    rec = roll.run(args.plies)          # warm-up launch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec = roll.run(args.plies, out=rec)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    episodes, black, white, draws, length = roll.stats.tolist()
    print(f"{args.envs} envs x {args.plies} plies in {dt * 1e6:.0f} us = {args.envs * args.plies / dt:.3e} env-steps/s")
    print(f"games finished so far: {episodes} (black {black}, white {white}, draws {draws}), "
          f"mean length {length / max(episodes, 1):.1f} plies")
    print(f"records: planes {tuple(rec.planes.shape)} int64 (u64 bits), meta {tuple(rec.meta.shape)} int32")
    small = type(rec)(planes=rec.planes[:4].contiguous(), meta=rec.meta[:4].contiguous())
    buf = unpack_records(small, env)     # RolloutBuffer layout of the first 4 plies
    print("unpacked:", {key: tuple(val.shape) for key, val in buf.items()})


if __name__ == "__main__":
    main()

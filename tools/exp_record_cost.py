"""Rollout kernel with and without the record stores, over batch sizes (developer tool, GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.random_rollout import RandomRollout

board = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "9x9x5").split("x"))
for N in (65536, 131072, 262144, 524288):
    env = TorchVectorMnkEnv(*board, N, device="cuda:0"); roll = RandomRollout(env, seed=0); buf = roll.alloc(256)
    for record in (True, False):
        for _ in range(300 if N == 65536 and record else 30): roll.run(256, out=buf if record else None, record=record)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(16): roll.run(256, out=buf if record else None, record=record)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 16
        print(board, N, "record" if record else "no record", round(us, 1), "us", f"{N * 256 / us * 1e6:.3e} env-steps/s", flush=True)
    del env, roll, buf

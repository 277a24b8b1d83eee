"""Timing experiments for the fused rollout kernel (developer tool, GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.random_rollout import RandomRollout

def run(nenv, T, record, launches=20, board=(9, 9, 5), log=False):
    env = TorchVectorMnkEnv(*board, nenv, device="cuda:0")
    roll = RandomRollout(env, seed=0)
    buf = roll.alloc(T, log_actions=log) if record else None
    for _ in range(3):
        roll.run(T, out=buf, record=record)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        roll.run(T, out=buf, record=record)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / launches
    print(f"board={board} N={nenv:7d} T={T:4d} record={int(record)} log={int(log)}  {us:9.1f} us/launch  {us/T*1e3:8.1f} ns/ply  "
          f"{nenv*T/us*1e6:.3e} env-steps/s", flush=True)

if __name__ == "__main__":
    run(65536, 64, True)
    run(65536, 256, True, launches=8)
    run(65536, 256, True, launches=8, log=True)
    run(65536, 512, True, launches=4)
    run(65536, 256, False, launches=8)
    run(131072, 256, True, launches=8)
    run(262144, 256, True, launches=8)
    run(32768, 256, True, launches=8, board=(19, 19, 5))
    run(65536, 256, True, launches=8, board=(3, 3, 3))
    run(65536, 256, True, launches=8, board=(13, 13, 5))
    run(65536, 256, True, launches=8, board=(7, 9, 7))

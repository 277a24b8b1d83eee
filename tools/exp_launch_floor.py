import os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.random_rollout import RandomRollout
env = TorchVectorMnkEnv(9, 9, 5, 65536, device="cuda:0")
roll = RandomRollout(env, seed=0)
for T in (4, 16, 64, 128, 256, 1024):
    for rec in (True, False):
        buf = roll.alloc(T) if rec else None
        for _ in range(3): roll.run(T, out=buf, record=rec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): roll.run(T, out=buf, record=rec)
        e1.record(); torch.cuda.synchronize()
        print(f"T={T:5d} record={int(rec)} {e0.elapsed_time(e1)*100:8.1f} us/launch", flush=True)

"""A/B: one lane per env vs two lanes per env in the fused rollout (developer tool, GPU box)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
ROOT = %r
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.random_rollout import RandomRollout
for board, N in (((9, 9, 5), 65536), ((9, 9, 5), 32768), ((9, 9, 5), 131072), ((13, 13, 5), 65536), ((3, 3, 3), 65536)):
    for rec, log in ((True, False), (True, True), (False, False)):
        env = TorchVectorMnkEnv(*board, N, device="cuda:0")
        roll = RandomRollout(env, seed=0)
        buf = roll.alloc(256, log_actions=log) if rec else None
        for _ in range(3): roll.run(256, out=buf, record=rec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): roll.run(256, out=buf, record=rec)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 8
        print(f"  {board} N={N:7d} rec={int(rec)} log={int(log)}: {us:8.1f} us / 256 plies  {N*256/us*1e6:.3e} env-steps/s", flush=True)
''' % ROOT
for pair in ("0", "1"):
    print("MNK_ROLLOUT_PAIR=" + pair, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MNK_ROLLOUT_PAIR=pair), check=False)

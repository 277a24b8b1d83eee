"""Developer tool: cost of writing the action log, by format (us per launch of 256 plies).
usage: python tools/exp_log_formats.py [board] [envs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch  # noqa: E402

from env.torch_vector_mnk_env import TorchVectorMnkEnv  # noqa: E402
from selfplay.random_rollout import ACT_BITS7, ACT_U8, ACT_U8P1, ACT_U16, RandomRollout, action_log_fits  # noqa: E402

board = sys.argv[1] if len(sys.argv) > 1 else "9x9x5"
nenv = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
m, n, k = (int(v) for v in board.split("x"))
env = TorchVectorMnkEnv(m, n, k, nenv, device="cuda:0")
roll = RandomRollout(env, seed=0)
T = 256
for _ in range(200):
    roll.run(T, record=False)
names = {0: "no log", ACT_U8: "u8", ACT_U16: "u16", ACT_BITS7: "7-bit", ACT_U8P1: "u8+1bit"}
for rep in range(2):
    for fmt in (0, ACT_U8, ACT_U16, ACT_BITS7, ACT_U8P1):
        if fmt and not action_log_fits(fmt, m * n):
            continue
        buf = roll.alloc(T, log_actions=fmt, with_state=False) if fmt else roll.alloc(T)
        for _ in range(20):
            roll.run(T, out=buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            roll.run(T, out=buf)
        e1.record()
        torch.cuda.synchronize()
        print(f"{board} x {nenv}  {names[fmt]:7s} {e0.elapsed_time(e1) * 10:8.2f} us/launch", flush=True)

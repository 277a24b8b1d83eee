"""Where does one lane per env overtake two lanes per env?  (developer tool, GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
ROOT = %r
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.random_rollout import RandomRollout
for board in ((9, 9, 5), (19, 19, 5)):
    for N in (24576, 32768, 36864, 40960, 45056, 49152, 57344, 65536):
        env = TorchVectorMnkEnv(*board, N, device="cuda:0"); roll = RandomRollout(env, seed=0); buf = roll.alloc(256)
        for _ in range(40): roll.run(256, out=buf)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(16): roll.run(256, out=buf)
        e1.record(); torch.cuda.synchronize()
        print(os.environ["MNK_ROLLOUT_PAIR"], board, N, round(e0.elapsed_time(e1) * 1e3 / 16, 1), flush=True)
''' % ROOT
for pair in ("0", "1"):
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MNK_ROLLOUT_PAIR=pair), check=False)

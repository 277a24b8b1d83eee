"""Developer tool: same-box A/B of the rollout kernel between two builds of the library.
usage: python tools/exp_ab_rollout.py <other libmnk_hip.so>     (children: MNK_HIP_LIB selects the build)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
ROOT = %r
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.random_rollout import RandomRollout
tag = os.environ.get("MNK_HIP_LIB", "in-tree")[-24:]
CASES = (((9, 9, 5), 65536), ((9, 9, 5), 32768), ((9, 9, 5), 262144), ((13, 13, 5), 65536), ((15, 15, 5), 65536),
         ((19, 19, 5), 65536), ((19, 19, 5), 32768), ((3, 3, 3), 65536), ((12, 12, 5), 65536))
if os.environ.get("AB_CASES"):
    CASES = tuple(((int(a), int(b), int(c)), int(n)) for a, b, c, n in (x.split(",") for x in os.environ["AB_CASES"].split(";")))
for board, N in CASES:
    env = TorchVectorMnkEnv(*board, N, device="cuda:0"); roll = RandomRollout(env, seed=0); buf = roll.alloc(256)
    for _ in range(300): roll.run(256, out=buf)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(64): roll.run(256, out=buf)
    e1.record(); torch.cuda.synchronize()
    print(f"{tag:>24s} {board} x {N}: {e0.elapsed_time(e1) * 1e3 / 64:8.1f} us per 256 plies", flush=True)
''' % ROOT
other = os.path.abspath(sys.argv[1])
for rep in range(int(os.environ.get("AB_REPS", "2"))):
    for lib in (other, None):
        env = dict(os.environ)
        if lib:
            env["MNK_HIP_LIB"] = lib
        else:
            env.pop("MNK_HIP_LIB", None)
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)

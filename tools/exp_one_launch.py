"""Developer tool: BASELINE config 2 as one launch per ply (mnk_step_random) under different workgroup shapes of the
write-out kernels (MNK_EMIT_ENVS / MNK_EMIT_THREADS are read once per process: one child process per shape).
usage: python tools/exp_one_launch.py            (parent: sweeps)    |    ... child <envs>"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch

    import bench
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay.random_rollout import RandomRollout

    nenv = int(sys.argv[2])
    env = TorchVectorMnkEnv(9, 9, 5, nenv, device="cuda:0")
    RandomRollout(env, seed=0).run(150, record=False)
    for _ in range(3):
        rate, us, floor = bench.api_path_one_launch_rate(env, 0)
    print(f"envs/WG={os.environ.get('MNK_EMIT_ENVS', 'auto'):>4s} threads={os.environ.get('MNK_EMIT_THREADS', '256'):>3s} "
          f"N={nenv}: {us:6.2f} us per ply  {rate:.3e} env-steps/s  (floor {floor:.2f} us)", flush=True)
else:
    for nenv in (65536, 262144):
        for envs in ("32", "64", "128"):
            for threads in ("128", "256"):
                e = dict(os.environ, MNK_EMIT_ENVS=envs, MNK_EMIT_THREADS=threads)
                subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(nenv)], env=e, check=False)

"""Sweep envs-per-workgroup x threads-per-workgroup of the write-out kernels (developer tool)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
ROOT = %r
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.random_rollout import RandomRollout
from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper
from selfplay.policy import RandomPolicy
def timeit(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for (m, n, k, N) in ((9, 9, 5, 65536), (19, 19, 5, 32768)):
    env = TorchVectorMnkEnv(m, n, k, N, device="cuda:0")
    RandomRollout(env, seed=0).run(150, record=False)
    obs = torch.empty((N, 2, m, n), device="cuda:0"); mask = torch.empty((N, m * n), dtype=torch.bool, device="cuda:0")
    t_obs = timeit(lambda: env.observe_into(obs, mask))
    wrap = TorchSelfPlayWrapper(env, seed=3); wrap.set_opponent(RandomPolicy(m * n)); wrap.reset()
    acts = torch.zeros(N, dtype=torch.long, device="cuda:0")
    import mnk_hip
    rew = torch.empty(N, device="cuda:0"); term = torch.empty(N, dtype=torch.bool, device="cuda:0")
    def fused():
        mnk_hip.call("mnk_selfplay_step_random", mnk_hip.ptr(env._planes), mnk_hip.ptr(env._meta), N, m, n, k,
                     mnk_hip.ptr(acts), mnk_hip.ptr(wrap.pending_resets), mnk_hip.ptr(wrap.agent_side), None, 1, 5, None, 0,
                     mnk_hip.ptr(rew), mnk_hip.ptr(term), mnk_hip.ptr(obs), mnk_hip.OBS_F32, mnk_hip.ptr(mask), None,
                     mnk_hip.ptr(env._err), None, None, None, 0, env._stream())
    t_f = timeit(fused)
    print(f"  {m}x{n} N={N}: observe {t_obs:6.1f} us   fused step {t_f:6.1f} us", flush=True)
''' % ROOT
for envs in (16, 32, 64):
    for threads in (64, 128, 256, 512):
        if threads < envs: continue
        print(f"envs/wg={envs} threads/wg={threads}", flush=True)
        e = dict(os.environ, MNK_EMIT_ENVS=str(envs), MNK_EMIT_THREADS=str(threads))
        subprocess.run([sys.executable, "-c", CHILD], env=e, check=False)

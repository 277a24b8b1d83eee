"""The fused self-play step far beyond the MALL (developer tool, GPU box): 2^20 ... 2^23 envs, f32 / bf16 / u8 observations --
what the write-out reaches when the launch's fixed cost no longer shows (eager launches, HIP events around 10 of them)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.policy import RandomPolicy
from selfplay.random_rollout import RandomRollout
from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper
DEV = "cuda:0"
for (m, n, k, nenv) in [(9, 9, 5, 1 << 20), (9, 9, 5, 1 << 22), (9, 9, 5, 1 << 23), (19, 19, 5, 1 << 21)]:
    c = m * n
    env = TorchVectorMnkEnv(m, n, k, nenv, device=DEV)
    RandomRollout(env, seed=0).run(60, record=False)
    w = TorchSelfPlayWrapper(env, seed=1)
    w.set_opponent(RandomPolicy(c, seed=2))
    for dt in (torch.float32, torch.bfloat16, torch.uint8):
        out = {"observation": torch.empty((nenv, 2, m, n), dtype=dt, device=DEV),
               "action_mask": torch.empty((nenv, c), dtype=torch.bool, device=DEV),
               "rewards": torch.empty(nenv, dtype=torch.float32, device=DEV),
               "terminated": torch.empty(nenv, dtype=torch.bool, device=DEV)}
        acts = torch.zeros(nenv, dtype=torch.long, device=DEV)
        w.reset(out=out)
        env.sample_legal_into(acts, seed=3, step=0)
        for _ in range(3):
            w.step(acts, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 10
        for _ in range(reps):
            w.step(acts, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        eb = out["observation"].element_size()
        nbytes = nenv * (2 * (16 * env.words + 8) + 2 * c * eb + c + 8 + 4 + 1 + 2 + 16)
        print(f"{m}x{n}x{k} N=2^{nenv.bit_length()-1} obs {str(dt).split('.')[-1]:8s}: fused self-play step {us:9.1f} us  {nbytes/1e6:8.1f} MB  {nbytes/us/1e6:5.2f} TB/s ({nbytes/us/8e6:.2f} of 8)", flush=True)
        del out
    del env, w
    torch.cuda.empty_cache()

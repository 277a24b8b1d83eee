"""cProfile of the host side of wrapper.step / env.step at 384 envs (developer tool): where the Python time goes."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper
from selfplay.policy import RandomPolicy

N = 384
env = TorchVectorMnkEnv(9, 9, 5, N, device="cuda:0")
wrap = TorchSelfPlayWrapper(env, seed=1); wrap.set_opponent(RandomPolicy(81)); wrap.reset()
acts = torch.zeros(N, dtype=torch.long, device="cuda:0")
for _ in range(200): wrap.step(acts)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5000): wrap.step(acts)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
env2 = TorchVectorMnkEnv(9, 9, 5, N, device="cuda:0"); env2.reset()
for _ in range(200): env2.step(acts)
pr = cProfile.Profile(); pr.enable()
for _ in range(5000): env2.step(acts)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)

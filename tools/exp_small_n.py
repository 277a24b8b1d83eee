"""Per-step latency of the API surface at the reference's default scale (384 envs) (developer tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper
from selfplay.policy import RandomPolicy

def wall(fn, reps=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

for N in (384, 4096, 65536):
    env = TorchVectorMnkEnv(9, 9, 5, N, device="cuda:0")
    wrap = TorchSelfPlayWrapper(env, seed=1); wrap.set_opponent(RandomPolicy(81)); obs, _ = wrap.reset()
    acts = torch.zeros(N, dtype=torch.long, device="cuda:0")
    agent = RandomPolicy(81, seed=2)
    state = {"obs": obs}
    def full():
        a = agent.act(state["obs"]); state["obs"], *_ = wrap.step(a)
    class Lowest:
        def act(self, o): return torch.argmax(o["action_mask"].to(torch.uint8), dim=1)
    print(f"N={N:6d}  env.step {wall(lambda: env.step(acts)):7.1f} us   env.observe {wall(env.observe):7.1f} us   "
          f"wrapper.step(fused random opp) {wall(lambda: wrap.step(acts)):7.1f} us   agent.act+wrapper.step {wall(full):7.1f} us", flush=True)
    wrap.set_opponent(Lowest())
    print(f"          wrapper.step(pre + torch argmax policy + post) {wall(lambda: wrap.step(acts)):7.1f} us", flush=True)

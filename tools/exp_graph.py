"""Eager vs hipGraph-replayed agent-step at the reference's training scale (developer tool, GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch, torch.nn as nn
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper
from selfplay.policy import RandomPolicy, FusedNNPolicy
from selfplay.graphed import GraphedAgentStep

DEV = "cuda:0"
m = n = 9; k = 5; c = 81

class ConvNet(nn.Module):  # small conv policy, BatchNorm-free (inference)
    def __init__(self, width=32):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(2, width, 3, padding=1), nn.ReLU(), nn.Conv2d(width, width, 3, padding=1), nn.ReLU(),
                                  nn.Conv2d(width, width, 3, padding=1), nn.ReLU())
        self.pi = nn.Sequential(nn.Conv2d(width, 2, 1), nn.Flatten(), nn.Linear(2 * c, c))
        self.v = nn.Sequential(nn.Conv2d(width, 1, 1), nn.Flatten(), nn.Linear(c, 1), nn.Tanh())
    def forward(self, obs, action_mask=None):
        f = self.body(obs)
        logits = self.pi(f)
        if action_mask is not None:
            logits = torch.where(action_mask.bool(), logits, torch.full_like(logits, -torch.inf))
        return torch.distributions.Categorical(logits=logits, validate_args=False), self.v(f)

def wall(fn, reps):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

for N in (384, 1024, 4096, 65536):
    for opp_kind in ("random", "nn"):
        torch.manual_seed(0)
        net, opp_net = ConvNet().to(DEV).eval(), ConvNet().to(DEV).eval()
        def make():
            w = TorchSelfPlayWrapper(TorchVectorMnkEnv(m, n, k, N, device=DEV), seed=1)
            w.set_opponent(RandomPolicy(c, seed=2) if opp_kind == "random" else FusedNNPolicy(opp_net, seed=2))
            return w
        w = make(); obs, _ = w.reset(); agent = FusedNNPolicy(net, seed=3); state = {"obs": obs}
        def eager():
            a = agent.act(state["obs"]); state["obs"], *_ = w.step(a)
        t_eager = wall(eager, 200)
        g = GraphedAgentStep(make(), net, seed=3)
        t_graph = wall(g.step, 200)
        print(f"N={N:6d} opponent={opp_kind:6s}  eager {t_eager:8.1f} us/step ({N/t_eager*1e6:.3e} agent-steps/s)   "
              f"hipGraph {t_graph:8.1f} us/step ({N/t_graph*1e6:.3e})   x{t_eager/t_graph:.2f}", flush=True)

# ---- the whole rollout (n_steps agent-steps) as ONE hipGraph writing the rollout buffer's rows (GraphedRollout) against
# the reference-shaped eager loop (net -> sample -> wrapper.step -> buffer.add) with the sink attached
from alg.rollout_buffer import RolloutBuffer
from selfplay.graphed import GraphedRollout

T = 64
for N in (384, 1024, 4096):
    for opp_kind in ("random", "nn"):
        torch.manual_seed(0)
        net, opp_net = ConvNet().to(DEV).eval(), ConvNet().to(DEV).eval()
        def make():
            w = TorchSelfPlayWrapper(TorchVectorMnkEnv(m, n, k, N, device=DEV), seed=1)
            w.set_opponent(RandomPolicy(c, seed=2) if opp_kind == "random" else FusedNNPolicy(opp_net, seed=2))
            return w
        w = make(); buf = RolloutBuffer(T, N, (2, m, n), c, device=DEV); w.attach_sink(buf)
        obs, _ = w.reset(); state = {"obs": obs}
        def eager_rollout():
            for _ in range(T):
                o = state["obs"]
                with torch.no_grad():
                    dist, values = net(o["observation"], o["action_mask"])
                    a = dist.sample(); lp = dist.log_prob(a)
                nxt, r, term, trunc, _ = w.step(a)
                buf.add(o["observation"], a, r, values, lp, term | trunc, o["action_mask"])
                state["obs"] = nxt
            buf.reset()
        t_eager = wall(eager_rollout, 5) / T
        buf2 = RolloutBuffer(T, N, (2, m, n), c, device=DEV)
        g = GraphedRollout(make(), buf2, net, seed=3)
        def graphed_rollout():
            g.run(); buf2.reset()
        t_graph = wall(graphed_rollout, 20) / T
        print(f"rollout of {T} steps, N={N:5d} opponent={opp_kind:6s}  eager loop (torch Categorical, sink) {t_eager:8.1f} us/step "
              f"({N/t_eager*1e6:.3e} agent-steps/s)   one hipGraph {t_graph:8.1f} us/step ({N/t_graph*1e6:.3e})   x{t_eager/t_graph:.2f}",
              flush=True)

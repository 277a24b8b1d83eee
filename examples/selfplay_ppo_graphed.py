"""Self-play PPO at the cadence of the reference's trainer (src/train.py:93-123) with the whole rollout as ONE hipGraph.

What train.py does per iteration -- ``set_opponent(NNPolicy(deepcopy(agent.network)))`` (or, with probability 0.15, a
frozen earlier self from the pool), ``agent.learn(env)`` = n_steps x [forward, masked sample, wrapper.step, buffer.add],
GAE, PPO epochs, every 20 iterations a copy of the agent into the pool -- here runs as:

    roll.set_opponent_weights(source)    the captured opponent's weights overwritten in place, its sampler re-keyed
    roll.run()                           n_steps agent-steps replayed as one graph: forward -> [draw + agent ply +
                                         opponent's view] -> opponent forward -> [draw + reply + merge + next observation],
                                         every field written straight into the RolloutBuffer's rows
    buffer.compute_advantages_and_returns(...); PPO epochs; buffer.reset()

The graph is captured once; the optimizer updates the agent's parameters in place, so the replays see the new weights.

    python examples/selfplay_ppo_graphed.py --board 3x3x3 --envs 2048 --iters 40
"""
import argparse
import copy
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import __graft_entry__ as entry  # noqa: E402


class ActorCritic(nn.Module):
    """``net(obs, mask) -> (Categorical, value)``, the forward contract of the reference nets (cnn.py:63-80); no argument
    validation in the Categorical (a host synchronisation, not allowed during graph capture)"""

    def __init__(self, cells):
        super().__init__()
        self.body = nn.Sequential(nn.Flatten(), nn.Linear(2 * cells, 256), nn.Tanh(), nn.Linear(256, 256), nn.Tanh())
        self.pi, self.v = nn.Linear(256, cells), nn.Linear(256, 1)

    def forward(self, obs, action_mask=None):
        h = self.body(obs)
        logits = self.pi(h)
        if action_mask is not None:
            logits = torch.where(action_mask.bool(), logits, torch.full_like(logits, -torch.inf))
        return torch.distributions.Categorical(logits=logits, validate_args=False), torch.tanh(self.v(h))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", default="3x3x3")
    ap.add_argument("--envs", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--iters", type=int, default=40)
    args = ap.parse_args()
    entry.build()
    from alg.rollout_buffer import RolloutBuffer
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay.graphed import GraphedRollout
    from selfplay.opponent_pool import OpponentPool
    from selfplay.policy import FusedNNPolicy, NNPolicy, RandomPolicy
    from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper
    from selfplay.validation import validate_gpu

    dev = "cuda"
    m, n, k = (int(v) for v in args.board.split("x"))
    cells = m * n
    torch.manual_seed(0)
    random.seed(0)
    net = ActorCritic(cells).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=2e-3)
    wrap = TorchSelfPlayWrapper(TorchVectorMnkEnv(m, n, k, args.envs, device=dev), seed=1)
    wrap.track_episodes()
    wrap.set_opponent(FusedNNPolicy(copy.deepcopy(net)))       # the captured opponent: same architecture as the agent
    pool = OpponentPool(max_size=8)
    pool.add_opponent(FusedNNPolicy(copy.deepcopy(net)))        # train.py:99
    buf = RolloutBuffer(args.steps, args.envs, (2, m, n), cells, device=dev)
    roll = GraphedRollout(wrap, buf, net)                       # one capture (its warm-up rollout is discarded)
    buf.reset()
    rollout_s = 0.0
    for it in range(args.iters):
        source = pool.get_random_opponent().model if random.random() < 0.15 else net    # train.py:107-113
        t0 = time.perf_counter()
        roll.set_opponent_weights(source)
        roll.run()
        nxt = roll.next_obs()
        with torch.no_grad():
            _, last = net(nxt["observation"], nxt["action_mask"])
        torch.cuda.synchronize()
        rollout_s += time.perf_counter() - t0
        buf.compute_advantages_and_returns(last.reshape(-1), 0.99, 0.95)
        for _ in range(4):
            for b_obs, b_act, b_logp, b_ret, b_adv, b_mask, _ in buf.get_data_loader(8192):
                dist, value = net(b_obs, b_mask)
                ratio = torch.exp(dist.log_prob(b_act) - b_logp)
                surrogate = torch.min(ratio * b_adv, torch.clamp(ratio, 0.8, 1.2) * b_adv).mean()
                loss = -surrogate + 0.5 * (value.reshape(-1) - b_ret).pow(2).mean() - 0.01 * dist.entropy().mean()
                opt.zero_grad()
                loss.backward()
                opt.step()
        buf.reset()
        if it % 20 == 0:                                          # train.py:122-123
            pool.add_opponent(FusedNNPolicy(copy.deepcopy(net)))
        if it % 10 == 9:
            stats = wrap.pop_episode_stats()                      # (also where a device-side error would surface)
            res = validate_gpu(NNPolicy(net), RandomPolicy(cells), (m, n, k), n_episodes=4096, device=dev)
            net.train()
            print(f"iter {it + 1:3d}: {stats['episodes']} training games, mean reward {stats['mean_reward']:+.3f}; "
                  f"score vs random {res['validation/vs_benchmark/score_rate']:.3f}; rollouts so far "
                  f"{rollout_s / (it + 1) * 1e3:.2f} ms each ({args.envs * args.steps * (it + 1) / rollout_s:.3e} agent-steps/s)")


if __name__ == "__main__":
    main()

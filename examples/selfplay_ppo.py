"""Self-play PPO on the HIP env with the drop-in pieces (wrapper, packed rollout buffer as the SINK of the fused step,
opponent pool, device-side episode statistics, validation) -- the structure of the reference's train.py / alg/ppo.py
loop, with a small MLP policy so it runs in seconds.  ``wrap.attach_sink(buf)`` makes every ``wrap.step`` write the
packed canonical planes of the next observation into row t+1 of the buffer and rewards / terminated into row t, so
``buf.add`` copies only actions, values and log-probabilities.

    python examples/selfplay_ppo.py --board 3x3x3 --envs 2048 --iters 40
"""
import argparse
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import __graft_entry__ as entry  # noqa: E402


class ActorCritic(nn.Module):
    def __init__(self, cells):
        super().__init__()
        self.body = nn.Sequential(nn.Flatten(), nn.Linear(2 * cells, 256), nn.Tanh(), nn.Linear(256, 256), nn.Tanh())
        self.pi, self.v = nn.Linear(256, cells), nn.Linear(256, 1)

    def forward(self, obs, action_mask=None):
        h = self.body(obs)
        logits = self.pi(h)
        if action_mask is not None:
            logits = torch.where(action_mask.bool(), logits, torch.full_like(logits, -torch.inf))
        return torch.distributions.Categorical(logits=logits), torch.tanh(self.v(h))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", default="3x3x3")
    ap.add_argument("--envs", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--iters", type=int, default=40)
    args = ap.parse_args()
    entry.build()
    from alg.packed_rollout_buffer import PackedRolloutBuffer
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay.opponent_pool import OpponentPool
    from selfplay.policy import NNPolicy, RandomPolicy
    from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper
    from selfplay.validation import validate_gpu

    dev = "cuda"
    m, n, k = (int(v) for v in args.board.split("x"))
    cells = m * n
    torch.manual_seed(0)
    net = ActorCritic(cells).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=2e-3)
    wrap = TorchSelfPlayWrapper(TorchVectorMnkEnv(m, n, k, args.envs, device=dev), seed=1)
    wrap.track_episodes()
    wrap.set_opponent(RandomPolicy(cells))
    pool = OpponentPool(max_size=8)
    buf = PackedRolloutBuffer(args.steps, args.envs, m, n, device=dev)
    wrap.attach_sink(buf)  # the step kernels write straight into the buffer's rows
    obs, _ = wrap.reset()
    packed = buf.row(0)["packed"]  # where reset() put the planes of the first observation
    for it in range(args.iters):
        if it % 5 == 4:
            pool.add_opponent(NNPolicy(copy.deepcopy(net)))
            wrap.set_opponent(pool.get_random_opponent())
            net.train()
        for t in range(args.steps):
            with torch.no_grad():
                dist, values = net(obs["observation"], obs["action_mask"])
                actions = dist.sample()
                logp = dist.log_prob(actions)
            obs, rewards, term, trunc, _ = wrap.step(actions)
            buf.add(packed, actions, rewards, values, logp, term | trunc)
            packed = buf.row(t + 1)["packed"]  # the step wrote the next observation's planes here (spill row at the end)
        with torch.no_grad():
            _, last = net(obs["observation"], obs["action_mask"])
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
        if it % 10 == 9:
            stats = wrap.pop_episode_stats()
            res = validate_gpu(NNPolicy(net), RandomPolicy(cells), (m, n, k), n_episodes=4096, device=dev)
            net.train()
            print(f"iter {it + 1:3d}: {stats['episodes']} training games, mean reward {stats['mean_reward']:+.3f}; "
                  f"score vs random {res['validation/vs_benchmark/score_rate']:.3f}")


if __name__ == "__main__":
    main()

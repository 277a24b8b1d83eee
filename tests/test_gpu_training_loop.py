"""GPU: the drop-in pieces working together the way the reference's trainer drives them.

A compact PPO (this file's own, ~60 lines; the reference's ``alg/ppo.py`` does not travel to the GPU box)
issues the call sequence of ``PPOAgent.learn`` (ppo.py:81-124, :131-148) against the HIP classes:
``wrapper.reset`` once, then ``net(obs, mask) -> sample -> wrapper.step -> buffer.add`` per step, GAE through
the drop-in ``RolloutBuffer``, minibatches from ``get_data_loader``, opponents rotated through an
``OpponentPool`` of frozen ``NNPolicy`` copies plus ``RandomPolicy``, and ``validate_gpu`` against a random
benchmark before and after.  The assertion is behavioural: on 3x3x3 the agent's score against random play goes up
markedly -- which only happens if observations, masks, rewards, terminations, autoresets and advantages are
all wired correctly."""
import copy

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class TinyActorCritic(nn.Module):
    """(obs, mask) -> (Categorical, value): the forward contract of the reference nets (cnn.py:63-80)."""

    def __init__(self, cells):
        super().__init__()
        self.body = nn.Sequential(nn.Flatten(), nn.Linear(2 * cells, 128), nn.Tanh(), nn.Linear(128, 128), nn.Tanh())
        self.pi, self.v = nn.Linear(128, cells), nn.Linear(128, 1)

    def forward(self, obs, action_mask=None):
        h = self.body(obs)
        logits = self.pi(h)
        if action_mask is not None:
            logits = torch.where(action_mask.bool(), logits, torch.full_like(logits, -torch.inf))
        return torch.distributions.Categorical(logits=logits), torch.tanh(self.v(h))


def test_ppo_style_training_improves_against_random():
    import __graft_entry__ as entry

    entry.build_hip()
    entry._ensure_path()
    from alg.rollout_buffer import RolloutBuffer
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay.opponent_pool import OpponentPool
    from selfplay.policy import NNPolicy, RandomPolicy
    from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper
    from selfplay.validation import validate_gpu

    torch.manual_seed(0)
    m = n = k = 3
    cells, nenv, n_steps = m * n, 2048, 16
    net = TinyActorCritic(cells).to(DEV)
    opt = torch.optim.Adam(net.parameters(), lr=3e-3)
    env = TorchVectorMnkEnv(m, n, k, nenv, device=DEV)
    wrap = TorchSelfPlayWrapper(env, seed=1)
    wrap.track_episodes()
    pool = OpponentPool(max_size=4)
    buf = RolloutBuffer(n_steps, nenv, (2, m, n), cells, device=DEV)
    benchmark = RandomPolicy(cells, seed=5)

    def score():
        res = validate_gpu(NNPolicy(net), benchmark, (m, n, k), n_episodes=4096, device=DEV)
        net.train()
        return res["validation/vs_benchmark/score_rate"]

    before = score()
    wrap.set_opponent(RandomPolicy(cells, seed=7))
    obs, _ = wrap.reset()
    for it in range(40):
        if it % 5 == 4:  # train.py:106-123: sometimes a frozen earlier self, kept in a bounded pool
            pool.add_opponent(NNPolicy(copy.deepcopy(net)))
            wrap.set_opponent(pool.get_random_opponent())
        for _ in range(n_steps):  # ppo.py:93-122
            observation, mask = obs["observation"], obs["action_mask"]
            with torch.no_grad():
                dist, values = net(observation, mask)
                actions = dist.sample()
                logp = dist.log_prob(actions)
            obs, rewards, term, trunc, _ = wrap.step(actions)
            buf.add(observation, actions, rewards, values, logp, term | trunc, mask)
        with torch.no_grad():
            _, last = net(obs["observation"], obs["action_mask"])
        buf.compute_advantages_and_returns(last.reshape(nenv), 0.99, 0.95)
        for _ in range(4):
            for b_obs, b_act, b_logp, b_ret, b_adv, b_mask, _ in buf.get_data_loader(8192):
                dist, value = net(b_obs, b_mask)
                ratio = torch.exp(dist.log_prob(b_act) - b_logp)
                surrogate = torch.min(ratio * b_adv, torch.clamp(ratio, 0.8, 1.2) * b_adv).mean()
                loss = -surrogate + 0.5 * (value.reshape(-1) - b_ret).pow(2).mean() - 0.01 * dist.entropy().mean()
                assert torch.isfinite(loss)
                opt.zero_grad()
                loss.backward()
                opt.step()
        buf.reset()
    stats = wrap.pop_episode_stats()
    after = score()
    print(f"score vs random: {before:.3f} -> {after:.3f}; episodes {stats}")
    assert stats["episodes"] > 50000 and 1.5 < stats["mean_length"] < 6
    # an untrained net scores ~0.5-0.6 against random play; a few hundred updates reach ~0.8-0.9
    assert after > before + 0.15 and after > 0.7, f"score vs random went {before:.3f} -> {after:.3f}"


def test_graphed_training_at_the_reference_cadence_improves_against_random():
    """The same trainer with the round-4 pieces, at the cadence of the reference's train.py:106-123: a NEW opponent
    before every rollout -- a copy of the current agent, or with probability 0.15 a frozen earlier self from the pool --
    installed into ONE captured ``GraphedRollout`` in place (``set_opponent_weights``: no capture after the first), the
    agent's and the opponent's masked draws folded into the step kernels, every step written into the buffer's rows.
    Behavioural assertion as above, plus: the graph object never changes."""
    import random

    import __graft_entry__ as entry

    entry.build_hip()
    entry._ensure_path()
    from alg.rollout_buffer import RolloutBuffer
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay.graphed import GraphedRollout
    from selfplay.opponent_pool import OpponentPool
    from selfplay.policy import FusedNNPolicy, NNPolicy, RandomPolicy
    from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper
    from selfplay.validation import validate_gpu

    class CaptureSafeActorCritic(TinyActorCritic):
        """the same network; its Categorical skips argument validation (a host synchronisation: not allowed while a
        hipGraph is being captured -- the reference nets build theirs the same way under torch.compile)"""

        def forward(self, obs, action_mask=None):
            h = self.body(obs)
            logits = self.pi(h)
            if action_mask is not None:
                logits = torch.where(action_mask.bool(), logits, torch.full_like(logits, -torch.inf))
            return torch.distributions.Categorical(logits=logits, validate_args=False), torch.tanh(self.v(h))

    torch.manual_seed(0)
    random.seed(0)
    m = n = k = 3
    cells, nenv, n_steps = m * n, 2048, 16
    net = CaptureSafeActorCritic(cells).to(DEV)
    opt = torch.optim.Adam(net.parameters(), lr=3e-3)
    wrap = TorchSelfPlayWrapper(TorchVectorMnkEnv(m, n, k, nenv, device=DEV), seed=1)
    wrap.track_episodes()
    wrap.set_opponent(FusedNNPolicy(copy.deepcopy(net), seed=2))  # train.py:99: the pool starts with a copy of the agent
    pool = OpponentPool(max_size=4)
    pool.add_opponent(FusedNNPolicy(copy.deepcopy(net), seed=3))
    buf = RolloutBuffer(n_steps, nenv, (2, m, n), cells, device=DEV)
    benchmark = RandomPolicy(cells, seed=5)

    def score():
        res = validate_gpu(NNPolicy(net), benchmark, (m, n, k), n_episodes=4096, device=DEV)
        net.train()
        return res["validation/vs_benchmark/score_rate"]

    before = score()
    roll = GraphedRollout(wrap, buf, net, seed=4)   # captures once (its warm-up is a real rollout: discarded here)
    graph = roll.graph
    buf.reset()
    for it in range(40):
        source = pool.get_random_opponent().model if random.random() < 0.15 else net  # train.py:107-113
        roll.set_opponent_weights(source)
        roll.run()                                                                      # ppo.py:93-122 as one hipGraph
        nxt = roll.next_obs()
        with torch.no_grad():
            _, last = net(nxt["observation"], nxt["action_mask"])
        buf.compute_advantages_and_returns(last.reshape(nenv), 0.99, 0.95)
        for _ in range(4):
            for b_obs, b_act, b_logp, b_ret, b_adv, b_mask, _ in buf.get_data_loader(8192):
                dist, value = net(b_obs, b_mask)
                ratio = torch.exp(dist.log_prob(b_act) - b_logp)
                surrogate = torch.min(ratio * b_adv, torch.clamp(ratio, 0.8, 1.2) * b_adv).mean()
                loss = -surrogate + 0.5 * (value.reshape(-1) - b_ret).pow(2).mean() - 0.01 * dist.entropy().mean()
                assert torch.isfinite(loss)
                opt.zero_grad()
                loss.backward()
                opt.step()
        buf.reset()
        if it % 20 == 0:  # train.py:122-123
            pool.add_opponent(FusedNNPolicy(copy.deepcopy(net), seed=100 + it))
    assert roll.graph is graph
    stats = wrap.pop_episode_stats()
    after = score()
    print(f"graphed, new opponent per rollout: score vs random {before:.3f} -> {after:.3f}; episodes {stats}")
    assert stats["episodes"] > 50000 and 1.5 < stats["mean_length"] < 6
    assert after > before + 0.1 and after > 0.65, f"score vs random went {before:.3f} -> {after:.3f}"

"""Golden fixtures for the CALLERS and the SINK of the rollout path, recorded from the imported reference
(build container only; ``/root/reference`` is mounted there):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_callers.py

SURVEY.md section 8 rows a18, a19/f1, a20, f3.  Data only -- inputs and what the reference computed from them:

  gae.npz           ``RolloutBuffer.compute_advantages_and_returns`` (src/alg/rollout_buffer.py:60-80) on random
                    [T, N] rewards / values / dones / last_values, including a partly filled buffer and T = 1
  validate.npz      ``validate_gpu`` (src/selfplay/validation.py:6-44) with deterministic agent / opponent policies
                    (oracle/policies.py) -> the result dict
  tournament.npz    ``MatchRunner._play_batch_games`` (src/model_comparison/match_runner.py:125-218) with
                    deterministic policies, policy 1 as black and as white -> (wins, losses, draws)
  ppo_learn_*.npz   two consecutive ``PPOAgent.learn`` calls (src/alg/ppo.py:81-152) on the reference wrapper with a
                    row-local deterministic opponent: the action / side streams the reference produced (inputs for
                    the replay) and everything its ``RolloutBuffer`` held when ``update_networks`` was entered, plus
                    the episode statistics ``learn`` returned.  The reference's classes are driven unmodified; the
                    recording happens in a proxy around the wrapper and in a hook on ``update_networks``.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.packing import pack_boards, pack_cells  # noqa: E402
from oracle.policies import (HighestLegalPolicy, LowestLegalPolicy, MaskHashPolicy, OpeningByRowPolicy,  # noqa: E402
                             RowSaltedHashPolicy)

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
from alg.ppo import PPOAgent  # noqa: E402
from alg.rollout_buffer import RolloutBuffer  # noqa: E402
from env.torch_vector_mnk_env import TorchVectorMnkEnv  # noqa: E402
from model_comparison.match_runner import GameConfig, MatchRunner  # noqa: E402
from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper  # noqa: E402
from selfplay.validation import validate_gpu  # noqa: E402
from utils.hardware import HardwareConfig  # noqa: E402
from utils.model_export import create_model_from_architecture  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# case name -> (n_steps of the buffer, steps actually added, N, gamma, lambda, done rate)
GAE_CASES = {
    "t64_n96": (64, 64, 96, 0.99, 0.95, 0.05),
    "t256_n33": (256, 256, 33, 0.99, 0.95, 0.02),
    "t1_n7": (1, 1, 7, 0.99, 0.95, 0.3),
    "partial10of16_n40": (16, 10, 40, 0.97, 0.9, 0.1),
    "all_done_t8_n16": (8, 8, 16, 0.99, 0.95, 1.0),
}

# (m, n, k, episodes, agent, opponent)
VALIDATE_CASES = [
    (3, 3, 3, 128, "rowhash0", "hash0"),
    (3, 3, 3, 65, "rowhash1", "lowest"),   # odd count: 32 black + 33 white
    (4, 6, 3, 96, "rowhash2", "hash1"),
    (9, 9, 5, 256, "rowhash0", "hash0"),
    (9, 9, 5, 64, "lowest", "highest"),
    (13, 13, 5, 48, "rowhash3", "hash2"),
]

# (m, n, k, games, p1, p2)
TOURNAMENT_CASES = [
    (3, 3, 3, 96, "open0", "open1"),
    (4, 6, 3, 64, "open2", "hash1"),
    (9, 9, 5, 192, "open0", "open3"),
    (9, 9, 5, 32, "lowest", "highest"),
    (13, 13, 5, 40, "open1", "open2"),
]

# (m, n, k, N, n_steps, architecture, opponent, seed)
PPO_CASES = [
    (3, 3, 3, 64, 16, "cnn_b_s", "hash0", 0),
    (9, 9, 5, 128, 24, "cnn_b_s", "hash1", 1),
]


def policy(name):
    kinds = {"rowhash": RowSaltedHashPolicy, "hash": MaskHashPolicy, "open": OpeningByRowPolicy}
    if name == "lowest":
        return LowestLegalPolicy()
    if name == "highest":
        return HighestLegalPolicy()
    for prefix, cls in kinds.items():
        if name.startswith(prefix):
            return cls(int(name[len(prefix):]))
    raise KeyError(name)


def make_gae():
    out = {}
    g = torch.Generator().manual_seed(7)
    for name, (n_steps, steps, nenv, gamma, lam, done_rate) in GAE_CASES.items():
        buf = RolloutBuffer(n_steps, nenv, (2, 3, 3), 9, device="cpu")
        rewards = torch.randint(-1, 2, (steps, nenv), generator=g).float()
        values = torch.tanh(torch.randn(steps, nenv, generator=g))
        dones = torch.rand(steps, nenv, generator=g) < done_rate
        last_values = torch.tanh(torch.randn(nenv, generator=g))
        buf.rewards[:steps] = rewards
        buf.values[:steps] = values
        buf.dones[:steps] = dones
        buf.ptr = steps
        buf.compute_advantages_and_returns(last_values, gamma, lam)
        out[name + "/rewards"] = rewards.numpy()
        out[name + "/values"] = values.numpy()
        out[name + "/dones"] = dones.numpy()
        out[name + "/last_values"] = last_values.numpy()
        out[name + "/hyper"] = np.array([n_steps, steps, nenv, gamma, lam], dtype=np.float64)
        out[name + "/advantages"] = buf.advantages.numpy().copy()  # all n_steps rows: rows >= steps stay zero
        out[name + "/returns"] = buf.returns.numpy().copy()
    return out


def make_validate():
    out = {}
    for (m, n, k, episodes, agent, opp) in VALIDATE_CASES:
        torch.manual_seed(0)
        res = validate_gpu(policy(agent), policy(opp), (m, n, k), n_episodes=episodes, device="cpu")
        key = f"{m}x{n}x{k}_{episodes}_{agent}_{opp}"
        out[key] = np.array([res["validation/vs_benchmark/win_rate"], res["validation/vs_benchmark/loss_rate"],
                             res["validation/vs_benchmark/draw_rate"], res["validation/vs_benchmark/score_rate"],
                             res["validation/vs_benchmark/games_played"]], dtype=np.float64)
        print("validate", key, out[key])
    return out


def make_tournament():
    out = {}
    for (m, n, k, games, p1, p2) in TOURNAMENT_CASES:
        runner = MatchRunner(GameConfig(m=m, n=n, k=k, device="cpu"))
        rows = []
        for p1_black in (True, False):
            rows.append(runner._play_batch_games(policy(p1), policy(p2), games, p1_is_black=p1_black))
        key = f"{m}x{n}x{k}_{games}_{p1}_{p2}"
        out[key] = np.array(rows, dtype=np.int64)  # [as black, as white] x (wins, losses, draws)
        print("tournament", key, out[key].tolist())
    return out


class RecordingEnv:
    """What ``PPOAgent.learn`` sees as ``vec_env`` (ppo.py:82, :102): forwards to the reference wrapper and
    writes down the actions it was given and the sides the wrapper held after every call."""

    def __init__(self, wrapper):
        self.wrapper = wrapper
        self.actions, self.sides = [], []

    def reset(self):
        out = self.wrapper.reset()
        self.sides.append(self.wrapper.agent_side.numpy().astype(np.uint8))
        return out

    def step(self, actions):
        self.actions.append(actions.numpy().astype(np.int32))
        out = self.wrapper.step(actions)
        self.sides.append(self.wrapper.agent_side.numpy().astype(np.uint8))
        return out


def make_ppo_learn(m, n, k, nenv, n_steps, arch, opp, seed):
    torch.manual_seed(seed)
    c = m * n
    env = TorchVectorMnkEnv(m, n, k, nenv, device="cpu")
    wrapper = TorchSelfPlayWrapper(env)
    wrapper.set_opponent(policy(opp))
    vec_env = RecordingEnv(wrapper)
    net = create_model_from_architecture(arch, obs_shape=(2, m, n), action_dim=c)
    hw = HardwareConfig(device="cpu", dtype=torch.float32, use_scaler=False, compile_mode=None)
    opt = torch.optim.AdamW(net.parameters(), lr=5e-4, eps=1e-5)
    agent = PPOAgent((2, m, n), c, net, hw_config=hw, n_steps=n_steps, optimizer=opt, batch_size=256, num_envs=nenv,
                     ppo_epochs=1)
    snaps = []
    inner = agent.update_networks

    def hooked():  # the buffer as PPO's update sees it (learn() resets it right after, ppo.py:148)
        b = agent.buffer
        snaps.append({
            "obs_planes": np.stack([pack_boards(o.numpy(), m, n) for o in b.observations]),
            "obs_mask": np.stack([pack_cells(a.numpy(), m, n) for a in b.action_masks]),
            "actions": b.actions.numpy().astype(np.int32), "rewards": b.rewards.numpy().astype(np.int8),
            "dones": b.dones.numpy().copy(), "values": b.values.numpy().copy(), "log_probs": b.log_probs.numpy().copy(),
            "advantages": b.advantages.numpy().copy(), "returns": b.returns.numpy().copy(), "ptr": b.ptr,
        })
        return inner()

    agent.update_networks = hooked
    out = {"geom": np.array([m, n, k, nenv, n_steps], dtype=np.int64),
           "hyper": np.array([agent.gamma, agent.gae_lambda], dtype=np.float64)}
    calls = 2
    for call in range(calls):
        metrics = agent.learn(vec_env)
        snap = snaps[call]
        assert snap.pop("ptr") == n_steps
        assert np.array_equal(snap["actions"], np.stack(vec_env.actions[call * n_steps:(call + 1) * n_steps]))
        for key, val in snap.items():
            out[f"call{call}/{key}"] = val
        out[f"call{call}/metrics"] = np.array([metrics.mean_reward, metrics.mean_length], dtype=np.float64)
    out["actions"] = np.stack(vec_env.actions)          # [calls * n_steps, N]
    out["sides"] = np.stack(vec_env.sides)              # [1 + calls * n_steps, N]: after reset, after every step
    return out, agent


def make_ppo_learn_with_last_values(*case):
    """``learn`` computes ``last_values`` with the pre-update weights and drops it.  To hand the replay the exact
    bootstrap values, wrap ``compute_advantages_and_returns`` of the agent's (reference) buffer for the recording:
    the argument it receives is stored as data."""
    captured = []
    orig = RolloutBuffer.compute_advantages_and_returns

    def spy(self, last_values, gamma=0.99, gae_lambda=0.95):
        captured.append(last_values.detach().reshape(-1).numpy().copy())
        return orig(self, last_values, gamma, gae_lambda)

    RolloutBuffer.compute_advantages_and_returns = spy
    try:
        out, _ = make_ppo_learn(*case)
    finally:
        RolloutBuffer.compute_advantages_and_returns = orig
    for call, lv in enumerate(captured):
        out[f"call{call}/last_values"] = lv
    return out


# Round 4 (``--more-boards``): a learn() recording on a board WITHOUT a built-in kernel variant, under a prefix of its own
# (the older tests address ppo_learn_* by index): boards_ppo_learn_6x7x4_n64_t16.npz
MORE_PPO_CASES = [(6, 7, 4, 64, 16, "cnn_b_s", "hash1", 2)]


def main():
    if "--more-boards" in sys.argv:
        torch.set_num_threads(1)
        for case in MORE_PPO_CASES:
            m, n, k, nenv, n_steps, arch, opp, seed = case
            path = os.path.join(OUT, f"boards_ppo_learn_{m}x{n}x{k}_n{nenv}_t{n_steps}.npz")
            np.savez_compressed(path, **make_ppo_learn_with_last_values(*case))
            print("wrote", os.path.basename(path), os.path.getsize(path))
        return
    # one intra-op thread: the PPO update between the two learn() calls reduces gradients over the batch, and the
    # sampled actions of the second call depend on the last bits of those sums -- with one thread the fixture
    # regenerates byte for byte on any machine
    torch.set_num_threads(1)
    for name, maker in (("gae", make_gae), ("validate", make_validate), ("tournament", make_tournament)):
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **maker())
        print("wrote", os.path.basename(path), os.path.getsize(path))
    for case in PPO_CASES:
        m, n, k, nenv, n_steps, arch, opp, seed = case
        path = os.path.join(OUT, f"ppo_learn_{m}x{n}x{k}_n{nenv}_t{n_steps}.npz")
        np.savez_compressed(path, **make_ppo_learn_with_last_values(*case))
        print("wrote", os.path.basename(path), os.path.getsize(path))


if __name__ == "__main__":
    main()

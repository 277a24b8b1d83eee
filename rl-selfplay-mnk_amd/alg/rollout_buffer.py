"""MI355X drop-in for the reference's ``RolloutBuffer`` (``/root/reference/src/alg/rollout_buffer.py:4-113``),
the sink of the rollout path.

Same constructor, fields (``[T, N, ...]`` device tensors, same names and dtypes), ``add``, ``ptr`` /
"Buffer was full." behaviour and minibatch generator, so ``PPOAgent`` (``alg/ppo.py:67-69, 106-108, 134,
148, 178``) uses it unchanged.  ``compute_advantages_and_returns`` -- in the reference a Python loop of T
steps x ~8 eager launches -- is one launch of ``mnk_gae`` (one lane per env, reverse scan over T, coalesced
over the env axis) with the reference's operation order and f32 roundings, so the results are bit-identical.

This directory has no ``__init__.py`` on purpose: ``alg`` is a namespace package in the reference too, so
with ``rl-selfplay-mnk_amd/`` ahead of the reference's ``src/`` on ``sys.path`` this module replaces
``alg.rollout_buffer`` while ``alg.ppo`` etc. keep resolving to the reference.
"""
import torch

import mnk_hip


class RolloutBuffer:
    def __init__(self, n_steps, num_envs, obs_shape, action_dim, device="cpu"):
        self.n_steps = n_steps
        self.num_envs = num_envs
        self.obs_shape = obs_shape
        self.action_dim = action_dim
        self.device = device
        if torch.device(device).type != "cuda":
            raise RuntimeError("RolloutBuffer: this is the MI355X implementation (GAE runs in a HIP kernel); "
                               "it needs a GPU device")
        mnk_hip.load()
        self.reset()

    def reset(self):
        """reference rollout_buffer.py:13-45: all fields zeroed, write pointer at 0"""
        t, n, dev = self.n_steps, self.num_envs, self.device

        def field(*shape, dtype=torch.float32):
            return torch.zeros((t, n) + shape, dtype=dtype, device=dev)

        self.observations = field(*self.obs_shape)
        self.actions = field(dtype=torch.long)
        self.log_probs = field()
        self.rewards = field()
        self.values = field()
        self.returns = field()
        self.advantages = field()
        self.dones = field(dtype=torch.bool)
        self.action_masks = field(self.action_dim, dtype=torch.bool)
        self.ptr = 0

    def add(self, obs, action, reward, value, log_prob, done, action_mask):
        """reference rollout_buffer.py:47-58"""
        if self.ptr >= self.n_steps:
            raise IndexError("Buffer was full.")
        row = self.ptr
        self.observations[row].copy_(obs)
        self.actions[row].copy_(action)
        self.rewards[row].copy_(reward)
        self.values[row].copy_(value.view(-1))
        self.log_probs[row].copy_(log_prob)
        self.dones[row].copy_(done)
        self.action_masks[row].copy_(action_mask)
        self.ptr += 1

    def compute_advantages_and_returns(self, last_values, gamma=0.99, gae_lambda=0.95):
        """reference rollout_buffer.py:60-80, one kernel launch"""
        steps = self.ptr
        if steps == 0 or self.num_envs == 0:
            return
        last_values = last_values.reshape(self.num_envs).to(torch.float32).contiguous()
        dev = self.rewards.device
        mnk_hip.call("mnk_gae", mnk_hip.ptr(self.rewards), mnk_hip.ptr(self.values), mnk_hip.ptr(self.dones),
                     mnk_hip.ptr(last_values), self.num_envs, steps, float(gamma), float(gamma * gae_lambda),
                     mnk_hip.ptr(self.advantages), mnk_hip.ptr(self.returns), mnk_hip.stream_ptr(dev))

    def get_data_loader(self, batch_size, normalize_advantages=True):
        """reference rollout_buffer.py:82-113: one shuffled pass over the first ``ptr`` steps"""
        steps = self.ptr
        total = steps * self.num_envs
        flat = {
            "obs": self.observations[:steps].reshape(total, *self.obs_shape),
            "actions": self.actions[:steps].reshape(total),
            "log_probs": self.log_probs[:steps].reshape(total),
            "returns": self.returns[:steps].reshape(total),
            "advantages": self.advantages[:steps].reshape(total),
            "masks": self.action_masks[:steps].reshape(total, self.action_dim),
            "values": self.values[:steps].reshape(total),
        }
        if normalize_advantages:
            adv = flat["advantages"]
            flat["advantages"] = (adv - adv.mean()) / (adv.std() + 1e-8)
        order = torch.randperm(total, device=self.device)
        for lo in range(0, total, batch_size):
            pick = order[lo:lo + batch_size]
            yield tuple(flat[key][pick] for key in ("obs", "actions", "log_probs", "returns", "advantages",
                                                    "masks", "values"))

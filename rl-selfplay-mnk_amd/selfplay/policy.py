"""Policies with the reference's ``act(obs, deterministic=False) -> (B,) int64`` contract
(``/root/reference/src/selfplay/policy.py:7-54``), drawing through the HIP sampler.

``RandomPolicy``   uniform over the legal cells (policy.py:13-29) -- the masked draw over a zero
                   logit row in ``mnk_sample_logits``; as a wrapper opponent it is recognised
                   (``fused_uniform_random``) and the whole self-play step becomes one launch.
``NNPolicy``       the reference's network policy, unchanged in behaviour: the net applies
                   its own mask and ``Categorical`` (policy.py:32-54).
``FusedNNPolicy``  same distribution, but mask + softmax + draw run in ``mnk_sample_logits``
                   on the raw logits (the epilogue of cnn.py:69-79 fused with the sample).
"""
from abc import ABC, abstractmethod
from typing import Dict

import torch
import torch.nn as nn

import mnk_hip


class Policy(ABC):
    @abstractmethod
    def act(self, obs: Dict[str, torch.Tensor], deterministic: bool = False) -> torch.Tensor:
        pass


class _HipSampler:
    """Philox-keyed draws from masked logits on the mask's device."""

    def __init__(self, seed=None):
        self.seed = int(torch.initial_seed() if seed is None else seed) & 0xFFFFFFFFFFFFFFFF
        self.calls = 0
        self.step_dev = None  # optional device int64[1] added to the step counter (graph replays)

    def draw(self, logits, mask, deterministic, want_logp=False):
        mask = mask.contiguous()
        if mask.device.type != "cuda":
            raise RuntimeError("mnk policies sample on the GPU; got a mask on " + str(mask.device))
        if mask.dim() == 1:
            mask = mask.unsqueeze(0)
        b, c = mask.shape
        logits = logits.to(torch.float32).reshape(b, c).contiguous()
        actions = torch.empty(b, dtype=torch.long, device=mask.device)
        logp = torch.empty(b, dtype=torch.float32, device=mask.device) if want_logp else None
        if b:
            mnk_hip.call("mnk_sample_logits", mnk_hip.ptr(logits), mnk_hip.ptr(mask), b, c, self.seed, self.calls,
                         mnk_hip.ptr(self.step_dev), 0, 1 if deterministic else 0, mnk_hip.ptr(actions),
                         mnk_hip.ptr(logp), mnk_hip.stream_ptr(mask.device))
        if self.step_dev is None:
            self.calls += 1
        return (actions, logp) if want_logp else actions


class RandomPolicy(Policy):
    fused_uniform_random = True  # lets TorchSelfPlayWrapper fold the opponent into its step kernel

    def __init__(self, action_dim: int, seed=None):
        self.action_dim = action_dim
        self._sampler = _HipSampler(seed)
        self._zeros = None

    def act(self, obs: Dict[str, torch.Tensor], deterministic: bool = False) -> torch.Tensor:
        mask = obs["action_mask"]
        if mask.dim() == 1:
            mask = mask.unsqueeze(0)
        if self._zeros is None or self._zeros.shape != mask.shape or self._zeros.device != mask.device:
            self._zeros = torch.zeros(mask.shape, dtype=torch.float32, device=mask.device)
        # deterministic: argmax of the 0/1 weights = first legal cell (policy.py:26-27)
        return self._sampler.draw(self._zeros, mask, deterministic)


class NNPolicy(Policy):
    def __init__(self, model: nn.Module):
        self.model = model
        self.model.eval()  # policy.py:35

    def act(self, obs: Dict[str, torch.Tensor], deterministic: bool = False) -> torch.Tensor:
        observation, action_mask = obs["observation"], obs["action_mask"]
        if observation.dim() == 3:  # policy.py:41-44
            observation = observation.unsqueeze(0)
        if action_mask.dim() == 1:
            action_mask = action_mask.unsqueeze(0)
        with torch.no_grad():
            dist, _ = self.model(observation, action_mask)
            return torch.argmax(dist.logits, dim=1) if deterministic else dist.sample()


class FusedNNPolicy(Policy):
    """``model(obs, None)`` must return ``(dist, value)`` with ``dist.logits`` the unmasked
    (possibly normalised) logits -- true for every reference architecture."""

    def __init__(self, model: nn.Module, seed=None):
        self.model = model
        self.model.eval()
        self._sampler = _HipSampler(seed)

    def act(self, obs: Dict[str, torch.Tensor], deterministic: bool = False) -> torch.Tensor:
        observation, action_mask = obs["observation"], obs["action_mask"]
        if observation.dim() == 3:
            observation = observation.unsqueeze(0)
        if action_mask.dim() == 1:
            action_mask = action_mask.unsqueeze(0)
        with torch.no_grad():
            dist, _ = self.model(observation, None)
            return self._sampler.draw(dist.logits, action_mask, deterministic)

"""Policies with the reference's ``act(obs, deterministic=False) -> (B,) int64`` contract
(``/root/reference/src/selfplay/policy.py:7-54``), drawing through the HIP sampler.

``RandomPolicy``   uniform over the legal cells (policy.py:13-29) -- the masked draw over a zero
                   logit row in ``mnk_sample_logits``; as a wrapper opponent it is recognised
                   (``fused_uniform_random``) and the whole self-play step becomes one launch.
``NNPolicy``       the reference's network policy, unchanged in behaviour: the net applies
                   its own mask and ``Categorical`` (policy.py:32-54).
``FusedNNPolicy``  same distribution, but mask + softmax + draw run in ``mnk_sample_logits``
                   on the raw logits (the epilogue of cnn.py:69-79 fused with the sample).  As a wrapper opponent it is
                   recognised (``fused_logits``): the wrapper asks it for its logits only and the draw happens INSIDE
                   ``mnk_selfplay_post_logits`` -- same stream of random numbers, one launch fewer per step.
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


_GOLDEN = 0x9E3779B97F4A7C15
_instances = [0]  # samplers created so far in this process


def _mix64(x: int) -> int:
    """splitmix64 finaliser: a bijection on 64-bit integers with good avalanche"""
    x &= 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


def default_key(seed=None) -> int:
    """Philox key of a sampler.  With an explicit ``seed`` the key IS the seed (reproducible, what the tests use).
    Without one -- the reference's constructors take no seed (policy.py:14, :33) -- every sampler created in the
    process gets its own key, derived from ``torch.initial_seed()`` and a per-process instance counter: two
    unseeded policies (agent and opponent of the same net, two RandomPolicy instances) must not draw the same
    uniform for row i on call c.  ``torch.manual_seed`` before constructing the policies makes a run repeatable."""
    if seed is not None:
        return int(seed) & 0xFFFFFFFFFFFFFFFF
    _instances[0] += 1
    return _mix64(int(torch.initial_seed()) + _instances[0] * _GOLDEN)


class _HipSampler:
    """Philox-keyed draws from masked logits on the mask's device."""

    def __init__(self, seed=None):
        self.seed = default_key(seed)
        self.calls = 0
        self.step_dev = None  # optional device int64[1] added to the step counter (graph replays)
        self.seed_dev = None  # optional device int64[1] that REPLACES the key (a captured sampler that can be re-keyed)
        self.env_id0 = 0      # Philox row id of row 0

    def block(self, deterministic=False):
        """the sampler's part of the argument list of mnk_sample_logits / mnk_selfplay_*_logits (after logits, dtype,
        mask): seed, seed_dev, step, step_dev, env_id0, deterministic"""
        return (self.seed, mnk_hip.ptr(self.seed_dev), self.calls, mnk_hip.ptr(self.step_dev), self.env_id0,
                1 if deterministic else 0)

    def advance(self):
        """one draw has been made with ``block()``"""
        if self.step_dev is None:
            self.calls += 1

    @staticmethod
    def prepare(logits, mask):
        """(logits contiguous f32 / bf16 [B, C] or None, MNK_LOGITS_* code, mask contiguous bool / u8 [B, C])"""
        mask = mask.contiguous()
        if mask.device.type != "cuda":
            raise RuntimeError("mnk policies sample on the GPU; got a mask on " + str(mask.device))
        if mask.dim() == 1:
            mask = mask.unsqueeze(0)
        if mask.dtype != torch.bool and mask.dtype != torch.uint8:
            mask = mask != 0
        b, c = mask.shape
        dtype = mnk_hip.LOGITS_F32
        if logits is not None:
            if logits.dtype == torch.bfloat16:
                dtype = mnk_hip.LOGITS_BF16
            elif logits.dtype != torch.float32:
                logits = logits.to(torch.float32)
            logits = logits.reshape(b, c).contiguous()
        return logits, dtype, mask

    def draw(self, logits, mask, deterministic, want_logp=False):
        """logits: f32 / bf16 [B, C] (bf16 is read as is -- what a network emits under autocast, alg/ppo.py:194),
        or None for all-zero logits (a uniform draw over the legal cells that reads only the mask)."""
        logits, dtype, mask = self.prepare(logits, mask)
        b, c = mask.shape
        actions = torch.empty(b, dtype=torch.long, device=mask.device)
        logp = torch.empty(b, dtype=torch.float32, device=mask.device) if want_logp else None
        if b:
            mnk_hip.call("mnk_sample_logits", mnk_hip.ptr(logits), dtype, mnk_hip.ptr(mask), b, c, *self.block(deterministic),
                         mnk_hip.ptr(actions), mnk_hip.ptr(logp), mnk_hip.stream_ptr(mask.device))
        self.advance()
        return (actions, logp) if want_logp else actions


HipSampler = _HipSampler


class RandomPolicy(Policy):
    fused_uniform_random = True  # lets TorchSelfPlayWrapper fold the opponent into its step kernel

    def __init__(self, action_dim: int, seed=None):
        self.action_dim = action_dim
        self._sampler = _HipSampler(seed)

    def act(self, obs: Dict[str, torch.Tensor], deterministic: bool = False) -> torch.Tensor:
        # all logits zero: uniform over the legal cells; deterministic: argmax of the 0/1 weights = first legal
        # cell (policy.py:26-27).  No logits tensor is materialised -- the kernel reads the mask only.
        return self._sampler.draw(None, obs["action_mask"], deterministic)


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

    fused_logits = True  # lets TorchSelfPlayWrapper fold this opponent's draw into its post kernel

    def __init__(self, model: nn.Module, seed=None):
        self.model = model
        self.model.eval()
        self._sampler = _HipSampler(seed)

    def logits(self, obs: Dict[str, torch.Tensor]) -> torch.Tensor:
        """the raw logits of the policy head on ``obs`` (no mask, no draw): what the step kernels with a folded-in
        draw take (``mnk_selfplay_*_logits``)"""
        observation = obs["observation"]
        if observation.dim() == 3:
            observation = observation.unsqueeze(0)
        with torch.no_grad():
            dist, _ = self.model(observation, None)
            return dist.logits

    def act(self, obs: Dict[str, torch.Tensor], deterministic: bool = False) -> torch.Tensor:
        action_mask = obs["action_mask"]
        if action_mask.dim() == 1:
            action_mask = action_mask.unsqueeze(0)
        return self._sampler.draw(self.logits(obs), action_mask, deterministic)

"""Policies used by the oracle and by the parity tests (test infrastructure).

``OracleRandomPolicy`` restates ``RandomPolicy`` of the reference
(``/root/reference/src/selfplay/policy.py:13-29``): multinomial over the legal
mask with the ``+1e-8`` guard for rows that have no legal cell.

The other policies are *row-local and deterministic*: the action of a row depends
only on that row's mask, never on the batch it arrives in.  That makes them give
the same move whether the caller hands over a ``nonzero`` subset (reference
wrapper) or the full fixed-shape batch (HIP wrapper), which is what the golden
wrapper traces rely on.  They accept ``act(obs)`` and ``act(obs, deterministic)``
because the wrapper calls the first form (wrapper:92-94) and validation the second.
"""
import torch


class OracleRandomPolicy:
    """reference policy.py:13-29"""

    def __init__(self, action_dim: int):
        self.action_dim = action_dim

    def act(self, obs, deterministic: bool = False):
        weights = obs["action_mask"].float()
        empty = weights.sum(dim=1, keepdim=True) == 0
        if bool(empty.any()):
            weights = weights + empty.float() * 1e-8
        if deterministic:
            return torch.argmax(weights, dim=1)
        return torch.multinomial(weights, num_samples=1).squeeze(1)


class LowestLegalPolicy:
    """First legal cell in action order; 0 when the row has none."""

    def act(self, obs, deterministic: bool = False):
        return torch.argmax(obs["action_mask"].to(torch.int64), dim=1)


class HighestLegalPolicy:
    """Last legal cell in action order; 0 when the row has none."""

    def act(self, obs, deterministic: bool = False):
        mask = obs["action_mask"].to(torch.int64)
        c = mask.shape[1]
        last = c - 1 - torch.argmax(mask.flip(1), dim=1)
        return torch.where(mask.sum(dim=1) > 0, last, torch.zeros_like(last))


class MaskHashPolicy:
    """The r-th legal cell, r = hash(mask) mod (number of legal cells); 0 when none.

    The hash is a weighted sum of the legal cell indices in int64, so it is exact
    on CPU and GPU alike.
    """

    def __init__(self, salt: int = 0):
        self.salt = int(salt)

    def act(self, obs, deterministic: bool = False):
        mask = obs["action_mask"].to(torch.int64)
        c = mask.shape[1]
        a = torch.arange(c, device=mask.device, dtype=torch.int64)
        w = a * a * 31 + a * 7 + 3 + self.salt
        h = (mask * w).sum(dim=1)
        nl = mask.sum(dim=1)
        r = torch.remainder(h, torch.clamp(nl, min=1))
        rank = torch.cumsum(mask, dim=1) - 1  # rank of each legal cell among the legal ones
        pick = (mask == 1) & (rank == r.unsqueeze(1))
        act = torch.argmax(pick.to(torch.int64), dim=1)
        return torch.where(nl > 0, act, torch.zeros_like(act))


class FixedCellPolicy:
    """Always the same cell -- the reference tests' ``ScriptedPolicy``
    (``src/tests/test_mnk_integration.py:11-24``); note: no ``deterministic`` argument."""

    def __init__(self, cell: int):
        self.cell = int(cell)

    def act(self, obs):
        mask = obs["action_mask"]
        return torch.full((mask.shape[0],), self.cell, dtype=torch.int64, device=mask.device)


class PhiloxOpponent:
    """Uniform legal reply keyed by (seed, global env id, step): what the fused HIP step
    ``mnk_selfplay_step_random`` draws for its built-in random opponent (stream OPP).
    ``OracleSelfPlay`` hands the env indices of the rows through ``act_indexed``; the test sets
    ``step`` before every wrapper call."""

    def __init__(self, seed: int, env_id0: int = 0):
        self.seed, self.env_id0, self.step = int(seed), int(env_id0), 0

    def act_indexed(self, obs, idx):
        import numpy as np

        from . import philox

        ids = (idx.numpy() + self.env_id0).astype(np.uint64)
        x = philox.rand_u32(self.seed, ids, self.step, philox.STREAM_OPP)
        return torch.from_numpy(philox.pick_legal(obs["action_mask"].numpy(), x))

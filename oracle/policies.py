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


def _rth_legal(mask, r):
    """index of the r-th legal cell of every row (r already reduced mod the row's legal count); 0 when none"""
    nl = mask.sum(dim=1)
    rank = torch.cumsum(mask, dim=1) - 1
    pick = (mask == 1) & (rank == r.unsqueeze(1))
    act = torch.argmax(pick.to(torch.int64), dim=1)
    return torch.where(nl > 0, act, torch.zeros_like(act))


def _mask_hash(mask, salt=0):
    c = mask.shape[1]
    a = torch.arange(c, device=mask.device, dtype=torch.int64)
    return (mask * (a * a * 31 + a * 7 + 3 + salt)).sum(dim=1)


class RowSaltedHashPolicy:
    """MaskHashPolicy whose hash also mixes in the row's position in the batch it is called on.  For callers
    that hand the policy the FULL env batch on every call (the agent side of ``validate_gpu``,
    validation.py:23-24): every env then plays its own game although all start from the same position."""

    def __init__(self, salt: int = 0):
        self.salt = int(salt)

    def act(self, obs, deterministic: bool = False):
        mask = obs["action_mask"].to(torch.int64)
        row = torch.arange(mask.shape[0], device=mask.device, dtype=torch.int64)
        h = _mask_hash(mask, self.salt) + row * 7919 + (row * row) % 1013
        return _rth_legal(mask, torch.remainder(h, torch.clamp(mask.sum(dim=1), min=1)))


class OpeningByRowPolicy:
    """Row-position-dependent while fewer than ``open_plies`` stones are on the board, MaskHashPolicy after that.
    For the tournament loop (match_runner.py:149-196): the reference hands policies ``obs[is_turn]`` subsets, the
    HIP loop the full batch -- during the first two plies no game can be over (k >= 2), both pass all rows in env
    order, so the opening may depend on the row; afterwards the action depends on the row's mask alone."""

    def __init__(self, salt: int = 0, open_plies: int = 2):
        self.salt, self.open_plies = int(salt), int(open_plies)

    def act(self, obs, deterministic: bool = False):
        mask = obs["action_mask"].to(torch.int64)
        c = mask.shape[1]
        nl = mask.sum(dim=1)
        row = torch.arange(mask.shape[0], device=mask.device, dtype=torch.int64)
        opening = (c - nl) < self.open_plies
        h = torch.where(opening, row * 40503 + self.salt * 17 + 11, _mask_hash(mask, self.salt))
        return _rth_legal(mask, torch.remainder(h, torch.clamp(nl, min=1)))

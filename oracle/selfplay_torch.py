"""Oracle restatement of the reference's self-play wrapper (test infrastructure).

Restates ``/root/reference/src/selfplay/torch_self_play_wrapper.py`` (class
``TorchSelfPlayWrapper``, lines 6-115): the single-agent view of the two-player
env with NEXT_STEP autoreset, an opponent that replies inside ``step`` and
zero-sum rewards.  Subset semantics (``nonzero`` index lists) are kept exactly as
the reference has them; the HIP path produces the same full-size results from
fixed-shape masked kernels and is compared with this class.

One addition for replayable parity: ``side_source(count) -> LongTensor`` supplies
the fresh sides drawn at (auto)reset.  The default draws them like the reference
does (``torch.randint(0, 2, (count,))``, wrapper:26 and :43-45).
"""
from typing import Callable, Optional

import torch

from .env_torch import WHITE


class OracleSelfPlay:
    """CPU oracle for ``TorchSelfPlayWrapper`` (reference wrapper:6-115)."""

    def __init__(self, env, side_source: Optional[Callable[[int], torch.Tensor]] = None):
        self.env = env
        self.device = env.device
        self.num_envs = env.num_envs
        self.opponent_policy = None
        self.agent_side = torch.zeros(self.num_envs, dtype=torch.int64, device=self.device)
        self.pending_resets = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        self._side_source = side_source or (
            lambda count: torch.randint(0, 2, (count,), device=self.device)
        )
        self.last_opponent_actions = None  # (indices, actions) of the latest opponent reply, for tests

    def set_opponent(self, policy):  # wrapper:16-17
        self.opponent_policy = policy

    # -- wrapper:19-30 -------------------------------------------------------
    def reset(self, seed=None, options=None):
        self.env.reset()
        self.pending_resets.fill_(False)
        if options and "agent_side" in options:
            self.agent_side[:] = torch.as_tensor(options["agent_side"], device=self.device)
        else:
            self.agent_side = self._side_source(self.num_envs).to(torch.int64)
        self._opponent_reply(torch.arange(self.num_envs, device=self.device))
        return self.canonical_obs(), {}

    # -- wrapper:32-67 -------------------------------------------------------
    def step(self, actions: torch.Tensor):
        to_reset = self.pending_resets.clone()
        to_play = ~to_reset
        rewards = torch.zeros(self.num_envs, device=self.device)
        terminated = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)

        if bool(to_reset.any()):
            ridx = torch.nonzero(to_reset).squeeze(1)
            self.env.reset(ridx)
            self.agent_side[ridx] = self._side_source(len(ridx)).to(torch.int64)
            self._opponent_reply(ridx)  # result ignored, as in the reference (:46)

        if bool(to_play.any()):
            pidx = torch.nonzero(to_play).squeeze(1)
            _, r_agent, d_agent = self.env.step_subset(actions[pidx], pidx)
            rewards[pidx] = r_agent[pidx]
            terminated[pidx] = d_agent[pidx]
            alive = pidx[~terminated[pidx]]
            if len(alive) > 0:
                r_opp, d_opp = self._opponent_reply(alive)
                if r_opp is not None:
                    rewards[alive] -= r_opp[alive]
                    terminated[alive] = d_opp[alive]

        self.pending_resets = terminated.clone()
        return self.canonical_obs(), rewards, terminated, torch.zeros_like(terminated), {}

    # -- wrapper:69-97 -------------------------------------------------------
    def _opponent_reply(self, env_idxs: torch.Tensor):
        if len(env_idxs) == 0:
            return None, None
        theirs = self.env.current_player[env_idxs] != self.agent_side[env_idxs]
        if not bool(theirs.any()):
            return None, None
        idx = env_idxs[theirs]
        full = self.env.observe()
        obs = full["observation"][idx]
        mask = full["action_mask"][idx]
        as_white = self.env.current_player[idx] == WHITE
        if bool(as_white.any()):
            obs[as_white] = torch.flip(obs[as_white], dims=(1,))
        with torch.no_grad():
            view = {"observation": obs, "action_mask": mask}
            if hasattr(self.opponent_policy, "act_indexed"):  # oracle-only hook: policies keyed by env id
                acts = self.opponent_policy.act_indexed(view, idx)
            else:
                acts = self.opponent_policy.act(view)
        self.last_opponent_actions = (idx.clone(), acts.clone())
        _, r, d = self.env.step_subset(acts, idx)
        return r, d

    # -- wrapper:99-115 ------------------------------------------------------
    def canonical_obs(self):
        raw = self.env.observe()
        obs = raw["observation"].clone()
        mask = raw["action_mask"]
        white_agent = self.agent_side == WHITE
        if bool(white_agent.any()):
            obs[white_agent] = torch.flip(obs[white_agent], dims=(1,))
        stuck = mask.sum(dim=1) == 0
        if bool(stuck.any()):
            mask[stuck, 0] = True
        return {"observation": obs, "action_mask": mask}

    get_agent_obs = canonical_obs  # wrapper:114-115

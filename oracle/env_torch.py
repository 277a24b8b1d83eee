"""Oracle restatement of the reference's batched MNK environment (test infrastructure).

Restates ``/root/reference/src/env/torch_vector_mnk_env.py`` (class
``TorchVectorMnkEnv``, lines 7-119) with the same arithmetic -- dense f32
``(N, 2, m, n)`` planes, a three-convolution K-in-a-row scan over the mover's
whole plane -- so that timing it on host cores is a fair "port" CPU baseline and
so that the HIP kernels can be compared with it bit for bit.

Behaviour that is deliberately kept (all observed on the reference):
  * no legality check: a move onto an occupied cell sets the mover's bit anyway
    and still counts as a move (reference lines 67-69; its validators at 86-104
    are never called);
  * the scan looks at the mover's plane only, anywhere on the board, so lines that
    existed before the move count and the other side's lines do not (106-119);
  * a win beats a draw on the last cell (72);
  * rewards / dones are full-size even for a subset step (75-80);
  * the side to move toggles for every stepped env, finished or not (82);
  * negative actions wrap the way torch advanced indexing wraps them.
"""
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

BLACK = 0  # reference src/env/constants.py:1
WHITE = 1  # reference src/env/constants.py:2


class OracleVectorEnv:
    """CPU oracle for ``TorchVectorMnkEnv`` (reference env:7-119)."""

    def __init__(self, m: int, n: int, k: int, num_envs: int, device: str = "cpu"):
        # reference env:9 -- boards smaller than k are rejected with an AssertionError
        assert m >= k and n >= k, f"Board ({m}x{n}) is too small for k={k}"
        self.m, self.n, self.k = int(m), int(n), int(k)
        self.num_envs = int(num_envs)
        self.device = device
        self.max_moves = self.m * self.n  # env:21
        # env:17-19 -- state containers
        self.boards = torch.zeros((self.num_envs, 2, self.m, self.n), dtype=torch.float32, device=device)
        self.current_player = torch.zeros(self.num_envs, dtype=torch.int64, device=device)
        self.move_counts = torch.zeros(self.num_envs, dtype=torch.int64, device=device)
        self.env_indices = torch.arange(self.num_envs, device=device)  # env:23
        # env:26-32 -- scan stencils: a row of ones, a column of ones, eye and mirrored eye
        eye = torch.eye(self.k, device=device)
        self._row = torch.ones((1, 1, 1, self.k), device=device)
        self._col = torch.ones((1, 1, self.k, 1), device=device)
        self._diag = torch.stack([eye, eye.flip(1)]).unsqueeze(1)

    # -- env:34-44 ---------------------------------------------------------
    def reset(self, env_indices: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        if env_indices is None:
            self.boards.zero_()
            self.current_player.zero_()
            self.move_counts.zero_()
        else:
            self.boards[env_indices] = 0
            self.current_player[env_indices] = BLACK
            self.move_counts[env_indices] = 0
        return self.observe()

    # -- env:46-53 ---------------------------------------------------------
    def observe(self) -> Dict[str, torch.Tensor]:
        taken = (self.boards != 0.0).any(dim=1)
        return {
            "observation": self.boards.clone(),
            "action_mask": (~taken).flatten(1),
        }

    # -- env:55-58 ---------------------------------------------------------
    def step(self, actions: torch.Tensor):
        return self.step_subset(actions, self.env_indices)

    # -- env:60-84 ---------------------------------------------------------
    def step_subset(
        self, actions: torch.Tensor, active_indices: torch.Tensor
    ) -> Tuple[Dict[str, torch.Tensor], torch.Tensor, torch.Tensor]:
        r = torch.div(actions, self.n, rounding_mode="floor")
        c = torch.remainder(actions, self.n)
        mover = self.current_player[active_indices]
        self.boards[active_indices, mover, r, c] = 1.0
        self.move_counts[active_indices] += 1

        won = self.scan_wins(active_indices, mover)
        drawn = (self.move_counts[active_indices] >= self.max_moves) & ~won

        rewards = torch.zeros(self.num_envs, device=self.device)
        if bool(won.any()):
            rewards[active_indices[won]] = 1.0
        dones = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        dones[active_indices] = won | drawn

        self.current_player[active_indices] ^= 1
        return self.observe(), rewards, dones

    # -- env:106-119 -------------------------------------------------------
    def scan_wins(self, active_indices: torch.Tensor, mover: torch.Tensor) -> torch.Tensor:
        plane = self.boards[active_indices, mover].unsqueeze(1)
        need = self.k - 0.1
        b = plane.shape[0]
        hit = (F.conv2d(plane, self._row) > need).reshape(b, -1).any(dim=1)
        hit |= (F.conv2d(plane, self._col) > need).reshape(b, -1).any(dim=1)
        hit |= (F.conv2d(plane, self._diag) > need).reshape(b, -1).any(dim=1)
        return hit

"""The HIP env under the reference's CALLERS' access patterns (who calls what, and how).

The reference cannot travel to the GPU box, so its own files never run there.  What can: the oracle's restatement
of the reference *wrapper* (oracle/selfplay_torch.py, pinned op for op against the imported reference in the build
container) and test-side restatements of the loops of match_runner.py / play.py -- all of which talk to an env only
through the reference's public surface (``reset(idx)``, ``observe()``, ``step_subset``, ``current_player[idx]``,
``boards[0].cpu()`` ...).  Here they drive the HIP ``TorchVectorMnkEnv`` instead of the oracle env, and every value
they see must equal what the oracle env (or the fixture recorded from the reference) gives, bit for bit.
"""
import numpy as np
import pytest
import torch

from oracle.env_torch import OracleVectorEnv
from oracle.policies import MaskHashPolicy, RowSaltedHashPolicy
from oracle.selfplay_torch import OracleSelfPlay
from replay import golden_files, replay_ppo_learn, replay_selfplay_trace
from test_gpu_callers import hip  # noqa: F401  (fixture)
from test_oracle_golden import OPP

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _ReplayWrapperLogic(OracleSelfPlay):
    """The reference wrapper's op sequence (wrapper:19-115) with the fresh sides taken from a fixture."""

    def step(self, actions):
        self._resetting = self.pending_resets.clone()
        return super().step(actions)


def _set_sides(wrapper, sides):
    sides_t = torch.from_numpy(sides.astype(np.int64)).to(wrapper.device)

    def source(count):
        if count == wrapper.num_envs:
            return sides_t.clone()
        return sides_t[torch.nonzero(wrapper._resetting).squeeze(1)]

    wrapper._side_source = source


@pytest.mark.parametrize("idx", range(7))
def test_reference_wrapper_logic_over_the_hip_env(hip, golden_dir, idx):  # noqa: F811
    """wrapper:32-112's calls -- index-subset resets, up to two ``step_subset`` per step, ``observe()`` whose result is
    indexed and mutated in place, ``current_player[idx] != agent_side[idx]`` -- on the HIP env reproduce the traces the
    reference recorded (tests/golden/selfplay_*.npz)."""
    path = golden_files(golden_dir, "selfplay_")[idx]
    log = np.load(path)
    m, n, k, nenv, _ = (int(v) for v in log["geom"])
    wrap = _ReplayWrapperLogic(hip.Env(m, n, k, nenv, device=DEV))
    wrap.set_opponent(OPP[path.split("_")[-2]]())
    replay_selfplay_trace(wrap, log, _set_sides)


def test_reference_learn_loop_over_wrapper_logic_and_hip_env(hip, golden_dir):  # noqa: F811
    """ppo.py:81-136's rollout with the reference wrapper's logic on the HIP env and the drop-in RolloutBuffer."""
    log = np.load(golden_files(golden_dir, "ppo_learn_")[0])
    m, n, k, nenv, _ = (int(v) for v in log["geom"])
    wrap = _ReplayWrapperLogic(hip.Env(m, n, k, nenv, device=DEV))
    wrap.set_opponent(MaskHashPolicy(0 if m == 3 else 1))

    def make_buffer(n_steps, num_envs, obs_shape, action_dim):
        return hip.Buffer(n_steps, num_envs, obs_shape, action_dim, device=DEV)

    replay_ppo_learn(wrap, make_buffer, log, _set_sides)


def arena_loop(env, p1, p2, p1_is_black, log):
    """The env-facing expressions of ``MatchRunner._play_batch_games`` (match_runner.py:140-215): turn masks from
    ``env.current_player``, boolean-indexed + cloned + flipped observations, ``step_subset`` on ``nonzero`` indices."""
    n_games, dev = env.num_envs, env.device
    obs = env.reset()
    dones = torch.zeros(n_games, dtype=torch.bool, device=dev)
    tally = [0, 0, 0]
    p1_side = 0 if p1_is_black else 1
    while not dones.all():
        current_player = env.current_player
        active = ~dones
        turn = [(current_player == p1_side) & active, (current_player != p1_side) & active]
        if not turn[0].any() and not turn[1].any():
            break
        actions = torch.full((n_games,), 0, dtype=torch.long, device=dev)
        for who, policy, side in ((0, p1, p1_side), (1, p2, 1 - p1_side)):
            if turn[who].any():
                seen = {"observation": obs["observation"][turn[who]].clone(), "action_mask": obs["action_mask"][turn[who]]}
                if side == 1:
                    seen["observation"] = torch.flip(seen["observation"], dims=(1,))
                actions[turn[who]] = policy.act(seen, deterministic=False)
        moving = torch.nonzero(turn[0] | turn[1]).squeeze(1)
        obs, rewards, step_dones = env.step_subset(actions[moving], moving)
        fresh = step_dones & ~dones
        won = (rewards == 1.0) & fresh
        tally[0] += int((won & turn[0]).sum().item())
        tally[1] += int((won & ~turn[0]).sum().item())
        tally[2] += int(((rewards == 0.0) & fresh).sum().item())
        dones[fresh] = True
        log.append((actions.cpu().numpy(), rewards.cpu().numpy(), step_dones.cpu().numpy(),
                    obs["observation"].cpu().numpy(), obs["action_mask"].cpu().numpy(),
                    torch.as_tensor(env.current_player[...]).cpu().numpy().copy()))  # (the oracle's is the live tensor)
    return tuple(tally)


@pytest.mark.parametrize("m,n,k,games,p1_is_black", [(3, 3, 3, 96, True), (9, 9, 5, 160, False), (4, 6, 3, 33, True)])
def test_arena_loop_sees_the_same_env(hip, m, n, k, games, p1_is_black):  # noqa: F811
    # row-salted policies: every game goes its own way, so the turn masks and the moving subset get ragged
    got_log, want_log = [], []
    want = arena_loop(OracleVectorEnv(m, n, k, games), RowSaltedHashPolicy(3), RowSaltedHashPolicy(11), p1_is_black, want_log)
    got = arena_loop(hip.Env(m, n, k, games, device=DEV), RowSaltedHashPolicy(3), RowSaltedHashPolicy(11), p1_is_black,
                     got_log)
    assert got == want and sum(got) == games and len(got_log) == len(want_log)
    assert min(want) > 0 or (m, n) != (3, 3)  # on 3x3 all three outcomes occur
    for t, (g, w) in enumerate(zip(got_log, want_log)):
        for name, a, b in zip(("actions", "rewards", "dones", "observation", "action_mask", "current_player"), g, w):
            assert a.dtype == b.dtype and np.array_equal(a, b), f"iteration {t}: {name}"


def console_game(env, moves):
    """The env-facing expressions of play.py's game loop (play.py:37-65, :95-109, :133): one env, ``.item()`` on
    ``current_player[0]``, the mask row of env 0, a 1-element action tensor on ``env.device``, ``boards[0].cpu().numpy()``."""
    seen = []
    obs = env.reset()
    for move in moves:
        side = env.current_player[0].item()
        legal = obs["action_mask"][0]
        assert bool(legal[move])
        obs, reward, done = env.step(torch.tensor([move], device=env.device))
        board = env.boards[0].cpu().numpy()
        seen.append((side, legal.cpu().numpy(), board.copy(), float(reward[0].item()), bool(done[0].item())))
        if done[0].item():
            break
    return seen


@pytest.mark.parametrize("m,n,k,moves", [
    (3, 3, 3, [4, 0, 2, 6, 3, 5, 1, 7, 8]),                       # a drawn game
    (9, 9, 5, [40, 0, 41, 9, 42, 18, 43, 27, 44]),                # black completes five in a row
    (19, 19, 5, [0, 360, 20, 340, 40, 320, 60, 300, 80]),         # a diagonal on the big board
])
def test_console_game_sees_the_same_env(hip, m, n, k, moves):  # noqa: F811
    want = console_game(OracleVectorEnv(m, n, k, 1), moves)
    got = console_game(hip.Env(m, n, k, 1, device=DEV), moves)
    assert len(got) == len(want)
    for t, (g, w) in enumerate(zip(got, want)):
        assert g[0] == w[0] and g[3] == w[3] and g[4] == w[4], f"ply {t}"
        assert np.array_equal(g[1], w[1]) and g[2].dtype == w[2].dtype and np.array_equal(g[2], w[2]), f"ply {t}"
    assert got[-1][4]  # every scripted game ends

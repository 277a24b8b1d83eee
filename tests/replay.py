"""Replay golden traces on anything that has the reference's env / wrapper surface.

The same functions drive the CPU oracle (``-m "not gpu"``) and the HIP classes
(``-m gpu``), so a parity test reads: load fixture, replay, compare bit for bit.
"""
import glob
import os

import numpy as np
import torch

from oracle.packing import pack_boards, pack_cells


def golden_files(golden_dir, prefix):
    return sorted(glob.glob(os.path.join(golden_dir, prefix + "*.npz")))


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def dense_boards(env):
    """env.boards as a plain numpy (N,2,m,n) array -- works for tensors and for the packed-state view."""
    b = env.boards
    b = b[...] if not isinstance(b, torch.Tensor) else b
    return _np(b)


def check_env_state(env, planes, cp, mc, where):
    m, n = env.m, env.n
    got = pack_boards(dense_boards(env), m, n)
    assert np.array_equal(got, planes), f"{where}: boards differ"
    assert np.array_equal(_np(env.current_player[...] if not isinstance(env.current_player, torch.Tensor)
                              else env.current_player).astype(np.uint8), cp), f"{where}: current_player differs"
    assert np.array_equal(_np(env.move_counts[...] if not isinstance(env.move_counts, torch.Tensor)
                              else env.move_counts).astype(np.int32), mc), f"{where}: move_counts differ"


def replay_env_log(env, log):
    """Replays a G1/G2 op-log (tests/golden/env_*.npz) on ``env``; asserts bit-equality after every op."""
    m, n, k, nenv, steps = (int(v) for v in log["geom"])
    assert (env.m, env.n, env.k, env.num_envs) == (m, n, k, nenv)
    dev = env.device
    obs = env.reset()
    assert bool(obs["action_mask"].all())
    for t in range(steps):
        acts = torch.from_numpy(log["actions"][t].astype(np.int64)).to(dev)
        active = log["active"][t]
        if active.all():
            obs, rew, done = env.step(acts)
        else:
            idx = torch.from_numpy(np.nonzero(active)[0]).to(dev)
            obs, rew, done = env.step_subset(acts[idx], idx)
        where = f"op {t}"
        assert rew.dtype == torch.float32 and done.dtype == torch.bool
        assert obs["observation"].dtype == torch.float32 and obs["action_mask"].dtype == torch.bool
        assert np.array_equal(_np(rew), log["rewards"][t].astype(np.float32)), f"{where}: rewards differ"
        assert np.array_equal(_np(done), log["dones"][t]), f"{where}: dones differ"
        assert np.array_equal(pack_cells(_np(obs["action_mask"]), m, n), log["mask"][t]), f"{where}: mask differs"
        assert np.array_equal(pack_boards(_np(obs["observation"]), m, n), log["planes"][t]), f"{where}: obs differs"
        check_env_state(env, log["planes"][t], log["cp"][t], log["mc"][t], where)
        ridx = torch.from_numpy(np.nonzero(log["reset"][t])[0]).to(dev)
        env.reset(ridx)


def replay_selfplay_trace(wrapper, log, set_sides):
    """Replays a G3 trace (tests/golden/selfplay_*.npz).

    ``set_sides(wrapper, sides_u8[N])`` tells the wrapper which side each env gets
    *if* it is (auto)reset in the next call -- the reference drew them from torch's
    global generator, the fixture stores what it drew.
    """
    m, n, k, nenv, steps = (int(v) for v in log["geom"])
    env = wrapper.env
    dev = wrapper.device

    def check(obs, t):
        where = f"selfplay {t}"
        assert np.array_equal(pack_boards(_np(obs["observation"]), m, n), log["obs_planes"][t]), f"{where}: obs"
        assert np.array_equal(pack_cells(_np(obs["action_mask"]), m, n), log["obs_mask"][t]), f"{where}: mask"
        assert np.array_equal(_np(wrapper.agent_side).astype(np.uint8), log["sides"][t]), f"{where}: sides"
        check_env_state(env, log["planes"][t], log["cp"][t], log["mc"][t], where)

    set_sides(wrapper, log["sides"][0])
    obs, info = wrapper.reset()
    assert info == {}
    check(obs, 0)
    for t in range(steps):
        set_sides(wrapper, log["sides"][t + 1])
        acts = torch.from_numpy(log["agent_actions"][t].astype(np.int64)).to(dev)
        obs, rew, term, trunc, info = wrapper.step(acts)
        where = f"selfplay step {t}"
        assert info == {} and rew.dtype == torch.float32 and term.dtype == torch.bool
        assert not bool(trunc.any()) and trunc.dtype == torch.bool
        assert np.array_equal(_np(rew), log["rewards"][t].astype(np.float32)), f"{where}: rewards"
        assert np.array_equal(_np(term), log["terminated"][t]), f"{where}: terminated"
        assert np.array_equal(_np(wrapper.pending_resets), log["pending"][t]), f"{where}: pending"
        check(obs, t + 1)


def play_scenario(env, sc):
    """Pokes a tests/scenarios.py position into a 1-env ``env`` and plays its plies.
    Returns the int8 rows in the layout of tests/golden/edges.npz."""
    env.reset()
    for (r, c) in sc["black"]:
        env.boards[0, 0, r, c] = 1
    for (r, c) in sc["white"]:
        env.boards[0, 1, r, c] = 1
    env.current_player[0] = sc["side"]
    env.move_counts[0] = sc["moves_made"]
    rows = []
    for a in sc["plies"]:
        obs, rew, done = env.step(torch.tensor([a], device=env.device))
        rows.append(np.concatenate([
            dense_boards(env).reshape(-1).astype(np.int8),
            _np(obs["action_mask"]).reshape(-1).astype(np.int8),
            np.array([rew[0].item(), float(done[0].item()), int(env.current_player[0]),
                      int(env.move_counts[0])]).astype(np.int8),
        ]))
    return np.stack(rows)

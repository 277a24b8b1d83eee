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
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu()
        return (t.float() if t.dtype == torch.bfloat16 else t).numpy()  # (numpy has no bfloat16; cells are 0 / 1)
    return np.asarray(t)


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
        # float32 like the reference's, unless the env was built with one of the opt-in narrow observation dtypes
        assert obs["observation"].dtype == getattr(env, "obs_dtype", torch.float32) and obs["action_mask"].dtype == torch.bool
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


def replay_ppo_learn(wrapper, make_buffer, log, set_sides, episode_stats=None):
    """Replays tests/golden/ppo_learn_*.npz: the call sequence of the reference's ``PPOAgent.learn`` rollout
    (alg/ppo.py:81-136 -- reset once, then per step: take obs/mask, step with the action the reference's network
    sampled, ``buffer.add(obs, action, reward, value, log_prob, done, mask)``, keep the new obs across ``learn``
    calls; after n_steps: ``compute_advantages_and_returns(last_values)``) on ``wrapper`` + ``make_buffer(...)``,
    and compares every buffer field with what the reference's buffer held, bit for bit.

    ``episode_stats()`` (optional) returns {"mean_reward", "mean_length"} of the episodes finished since its last
    call -- compared with the ``TrainingMetrics`` the reference's ``learn`` returned (ppo.py:110-120, :150-151).
    """
    m, n, k, nenv, n_steps = (int(v) for v in log["geom"])
    gamma, lam = (float(v) for v in log["hyper"])
    dev = wrapper.device
    calls = log["actions"].shape[0] // n_steps
    set_sides(wrapper, log["sides"][0])
    obs, info = wrapper.reset()
    assert info == {}
    for call in range(calls):
        buf = make_buffer(n_steps, nenv, (2, m, n), m * n)
        pre = f"call{call}/"
        values = torch.from_numpy(log[pre + "values"]).to(dev)
        log_probs = torch.from_numpy(log[pre + "log_probs"]).to(dev)
        for t in range(n_steps):
            g = call * n_steps + t
            observation, action_mask = obs["observation"], obs["action_mask"]
            actions = torch.from_numpy(log["actions"][g].astype(np.int64)).to(dev)
            set_sides(wrapper, log["sides"][g + 1])
            next_obs, rewards, terminateds, truncateds, _ = wrapper.step(actions)
            dones = terminateds | truncateds
            buf.add(observation, actions, rewards, values[t].view(-1, 1), log_probs[t], dones, action_mask)
            obs = next_obs
        assert buf.ptr == n_steps
        buf.compute_advantages_and_returns(torch.from_numpy(log[pre + "last_values"]).to(dev), gamma, lam)
        where = f"learn call {call}"
        got_planes = np.stack([pack_boards(_np(o), m, n) for o in buf.observations])
        got_mask = np.stack([pack_cells(_np(a), m, n) for a in buf.action_masks])
        assert np.array_equal(got_planes, log[pre + "obs_planes"]), f"{where}: observations"
        assert np.array_equal(got_mask, log[pre + "obs_mask"]), f"{where}: action_masks"
        assert np.array_equal(_np(buf.actions).astype(np.int32), log[pre + "actions"]), f"{where}: actions"
        assert np.array_equal(_np(buf.rewards), log[pre + "rewards"].astype(np.float32)), f"{where}: rewards"
        assert np.array_equal(_np(buf.dones), log[pre + "dones"]), f"{where}: dones"
        assert np.array_equal(_np(buf.values), log[pre + "values"]), f"{where}: values"
        assert np.array_equal(_np(buf.log_probs), log[pre + "log_probs"]), f"{where}: log_probs"
        assert np.array_equal(_np(buf.advantages), log[pre + "advantages"]), f"{where}: advantages"
        assert np.array_equal(_np(buf.returns), log[pre + "returns"]), f"{where}: returns"
        if episode_stats is not None:
            st = episode_stats()
            want_reward, want_length = (float(v) for v in log[pre + "metrics"])
            assert abs(st["mean_reward"] - want_reward) < 1e-9, f"{where}: mean_reward"
            assert abs(st["mean_length"] - want_length) < 1e-9, f"{where}: mean_length"

"""GPU parity for the CALLERS and the SINK of the rollout path, against fixtures recorded from the imported
reference (tests/golden/make_golden_callers.py): SURVEY.md section 8 rows a18 (PPOAgent.learn rollout), a19/f1
(RolloutBuffer + GAE -> mnk_gae), a20 (validate_gpu), f3 (tournament loop), f4 (minibatch gather from packed
observations -> mnk_gather_obs).  Everything goes through the C ABI of libmnk_hip.so; bit-exact."""
import numpy as np
import pytest
import torch

from oracle.packing import unpack_boards, unpack_cells
from oracle.policies import MaskHashPolicy
from replay import golden_files, replay_ppo_learn
from test_oracle_callers import fixture_cases, gae_cases, named_policy

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    import __graft_entry__ as entry

    entry.build_hip()
    entry._ensure_path()
    import mnk_hip
    from alg.packed_rollout_buffer import PackedRolloutBuffer
    from alg.rollout_buffer import RolloutBuffer
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay import random_rollout, tournament, validation
    from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper

    mnk_hip.load()
    assert torch.cuda.is_available()

    class NS:
        pass

    ns = NS()
    ns.Env, ns.Wrapper, ns.Buffer, ns.PackedBuffer = TorchVectorMnkEnv, TorchSelfPlayWrapper, RolloutBuffer, PackedRolloutBuffer
    ns.rollout, ns.tournament, ns.validation = random_rollout, tournament, validation
    return ns


def test_gae_kernel_equals_the_reference_buffer(hip, golden_dir):
    """mnk_gae (through the drop-in RolloutBuffer and through selfplay.random_rollout.gae) ==
    RolloutBuffer.compute_advantages_and_returns of the reference (rollout_buffer.py:60-80), bit for bit; a partly
    filled buffer leaves the rows past ``ptr`` zero like the reference does."""
    data, names = gae_cases(golden_dir)
    for name in names:
        n_steps, steps, nenv, gamma, lam = data[name + "/hyper"]
        n_steps, steps, nenv = int(n_steps), int(steps), int(nenv)
        buf = hip.Buffer(n_steps, nenv, (2, 3, 3), 9, device=DEV)
        buf.rewards[:steps] = torch.from_numpy(data[name + "/rewards"]).to(DEV)
        buf.values[:steps] = torch.from_numpy(data[name + "/values"]).to(DEV)
        buf.dones[:steps] = torch.from_numpy(data[name + "/dones"]).to(DEV)
        buf.ptr = steps
        last = torch.from_numpy(data[name + "/last_values"]).to(DEV)
        buf.compute_advantages_and_returns(last, float(gamma), float(lam))
        assert np.array_equal(buf.advantages.cpu().numpy(), data[name + "/advantages"]), name
        assert np.array_equal(buf.returns.cpu().numpy(), data[name + "/returns"]), name
        adv, ret = hip.rollout.gae(buf.rewards[:steps], buf.values[:steps], buf.dones[:steps], last, float(gamma),
                                   float(lam))
        assert np.array_equal(adv.cpu().numpy(), data[name + "/advantages"][:steps]), name
        assert np.array_equal(ret.cpu().numpy(), data[name + "/returns"][:steps]), name


def test_gae_accepts_strided_inputs(hip, golden_dir):
    """transposed views in, same numbers out (the temporaries made contiguous must outlive the launch)"""
    data, _ = gae_cases(golden_dir)
    name = "t64_n96"
    r, v, d = (torch.from_numpy(data[f"{name}/{f}"]).to(DEV) for f in ("rewards", "values", "dones"))
    last = torch.from_numpy(data[name + "/last_values"]).to(DEV)
    rt, vt, dt = (x.t().contiguous().t() for x in (r, v, d))  # same values, column-major storage
    assert not rt.is_contiguous()
    adv, ret = hip.rollout.gae(rt, vt, dt, last.view(-1, 1).expand(-1, 2)[:, 1], 0.99, 0.95)
    assert np.array_equal(adv.cpu().numpy(), data[name + "/advantages"])
    assert np.array_equal(ret.cpu().numpy(), data[name + "/returns"])


def test_validate_gpu_equals_the_reference(hip, golden_dir):
    """selfplay.validation.validate_gpu on the HIP wrapper == the reference's validate_gpu (validation.py:6-44)
    for deterministic agent / opponent policies: the same result dict, exactly."""
    for key, (m, n, k, episodes, agent, opp, want) in fixture_cases(golden_dir, "validate.npz").items():
        res = hip.validation.validate_gpu(named_policy(agent), named_policy(opp), (m, n, k), n_episodes=episodes,
                                          device=DEV)
        got = [res[f"validation/vs_benchmark/{f}"] for f in ("win_rate", "loss_rate", "draw_rate", "score_rate",
                                                             "games_played")]
        assert got == want.tolist(), key


def test_tournament_loop_equals_the_reference(hip, golden_dir):
    """selfplay.tournament.play_batch_games == MatchRunner._play_batch_games of the reference
    (match_runner.py:125-218), policy 1 as black and as white: the same (wins, losses, draws)."""
    for key, (m, n, k, games, p1, p2, want) in fixture_cases(golden_dir, "tournament.npz").items():
        for row, p1_black in enumerate((True, False)):
            got = hip.tournament.play_batch_games(named_policy(p1), named_policy(p2), (m, n, k), games, p1_black,
                                                  device=DEV)
            assert list(got) == want[row].tolist(), (key, p1_black)
        match = hip.tournament.play_match(named_policy(p1), named_policy(p2), (m, n, k), 2 * games, device=DEV)
        assert (match["wins"], match["losses"], match["draws"]) == tuple(int(v) for v in want.sum(axis=0))


def _set_sides(wrapper, sides):
    wrapper.force_sides(torch.from_numpy(sides.astype(np.int64)))


@pytest.mark.parametrize("idx", range(2))
def test_ppo_learn_rollout_equals_the_reference(hip, golden_dir, idx):
    """Two consecutive PPOAgent.learn rollouts of the reference (ppo.py:81-136) replayed on the HIP wrapper + the
    drop-in RolloutBuffer: observations, masks, rewards, dones, advantages and returns in the buffer equal the
    reference buffer's, and the device-side episode counters (track_episodes) reproduce the mean_reward /
    mean_length that learn() returned."""
    log = np.load(golden_files(golden_dir, "ppo_learn_")[idx])
    m, n, k, nenv, _ = (int(v) for v in log["geom"])
    wrap = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV))
    wrap.set_opponent(MaskHashPolicy(0 if m == 3 else 1))
    wrap.track_episodes()

    def make_buffer(n_steps, num_envs, obs_shape, action_dim):
        return hip.Buffer(n_steps, num_envs, obs_shape, action_dim, device=DEV)

    replay_ppo_learn(wrap, make_buffer, log, _set_sides, episode_stats=wrap.pop_episode_stats)


@pytest.mark.parametrize("idx", range(2))
def test_packed_buffer_gather_equals_the_reference_observations(hip, golden_dir, idx):
    """f4: the same replay into PackedRolloutBuffer (packed canonical planes in); mnk_gather_obs then hands back
    the observations and masks the REFERENCE's buffer held for the drawn samples (every sample, in a shuffled
    order and in minibatches), and the packed buffer's GAE equals the reference's too."""
    log = np.load(golden_files(golden_dir, "ppo_learn_")[idx])
    m, n, k, nenv, n_steps = (int(v) for v in log["geom"])
    gamma, lam = (float(v) for v in log["hyper"])
    c = m * n
    wrap = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV))
    wrap.set_opponent(MaskHashPolicy(0 if m == 3 else 1))
    _set_sides(wrap, log["sides"][0])
    obs, _ = wrap.reset()
    g = torch.Generator(device="cpu").manual_seed(idx)
    for call in range(2):
        pre = f"call{call}/"
        buf = hip.PackedBuffer(n_steps, nenv, m, n, device=DEV)
        values = torch.from_numpy(log[pre + "values"]).to(DEV)
        log_probs = torch.from_numpy(log[pre + "log_probs"]).to(DEV)
        for t in range(n_steps):
            step = call * n_steps + t
            packed = wrap.packed_obs()
            actions = torch.from_numpy(log["actions"][step].astype(np.int64)).to(DEV)
            _set_sides(wrap, log["sides"][step + 1])
            obs, rew, term, trunc, _ = wrap.step(actions)
            buf.add(packed, actions, rew, values[t], log_probs[t], term | trunc)
        buf.compute_advantages_and_returns(torch.from_numpy(log[pre + "last_values"]).to(DEV), gamma, lam)
        assert np.array_equal(buf.advantages.cpu().numpy(), log[pre + "advantages"])
        assert np.array_equal(buf.returns.cpu().numpy(), log[pre + "returns"])
        # the reference buffer's dense fields for all T*N samples
        ref_obs = np.concatenate([unpack_boards(p, m, n) for p in log[pre + "obs_planes"]]).astype(np.float32)
        ref_mask = np.concatenate([unpack_cells(q, m, n) for q in log[pre + "obs_mask"]]).astype(bool)
        order = torch.randperm(n_steps * nenv, generator=g)
        for lo in range(0, n_steps * nenv, 333):
            pick = order[lo:lo + 333]
            o, msk = buf.gather(pick.to(DEV))
            assert np.array_equal(o.cpu().numpy(), ref_obs[pick.numpy()].reshape(-1, 2, m, n)), (call, lo)
            assert np.array_equal(msk.cpu().numpy(), ref_mask[pick.numpy()].reshape(-1, c)), (call, lo)

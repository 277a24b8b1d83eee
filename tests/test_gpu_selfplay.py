"""GPU parity: the HIP self-play wrapper, samplers, record unpack and GAE (through the C ABI)
against the golden vectors of the reference and against the oracle."""
import numpy as np
import pytest
import torch

from oracle import philox
from oracle.env_torch import OracleVectorEnv
from oracle.packing import pack_boards, pack_cells, planes_from_record_rows, unpack_boards
from oracle.policies import (FixedCellPolicy, HighestLegalPolicy, LowestLegalPolicy, MaskHashPolicy,
                             PhiloxOpponent, RowSaltedHashPolicy)
from oracle.rollout import gae as oracle_gae
from oracle.selfplay_torch import OracleSelfPlay
from conftest import random_play_stats
from replay import golden_files, replay_selfplay_trace

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
OPP = {"lowest": LowestLegalPolicy, "highest": HighestLegalPolicy, "hash": MaskHashPolicy}


@pytest.fixture(scope="module")
def hip():
    import __graft_entry__ as entry

    entry.build_hip()
    entry._ensure_path()
    import mnk_hip
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay import policy, random_rollout, validation
    from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper

    mnk_hip.load()
    assert torch.cuda.is_available()

    class NS:
        pass

    ns = NS()
    ns.lib, ns.Env, ns.Wrapper = mnk_hip, TorchVectorMnkEnv, TorchSelfPlayWrapper
    ns.policy, ns.rollout, ns.validation = policy, random_rollout, validation
    return ns


# ----------------------------------------------------------------------------- G3 traces
@pytest.mark.parametrize("idx", range(7))
def test_selfplay_trace_matches_reference(hip, golden_dir, idx):
    """TorchSelfPlayWrapper traces recorded from the reference (row-local deterministic opponents,
    the sides the reference drew are replayed through force_sides)."""
    path = golden_files(golden_dir, "selfplay_")[idx]
    log = np.load(path)
    m, n, k, nenv, _ = (int(v) for v in log["geom"])
    wrap = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV))
    wrap.set_opponent(OPP[path.split("_")[-2]]())
    replay_selfplay_trace(wrap, log, lambda w, sides: w.force_sides(torch.from_numpy(sides.astype(np.int64))))


@pytest.mark.parametrize("obs_dtype", [torch.bfloat16, torch.uint8])
@pytest.mark.parametrize("idx", [0, 2, 4, 6])
def test_selfplay_trace_matches_reference_with_narrow_observations(hip, golden_dir, idx, obs_dtype):
    """The reference's wrapper traces on an env that hands out bf16 / u8 observations (opt-in, ABI v4): the opponent
    sees narrow observations, the agent gets narrow canonical observations, and every cell, mask bit, reward, flag
    and side equals what the reference recorded."""
    path = golden_files(golden_dir, "selfplay_")[idx]
    log = np.load(path)
    m, n, k, nenv, _ = (int(v) for v in log["geom"])
    wrap = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV, obs_dtype=obs_dtype))
    seen = []

    class Spy:
        def __init__(self, inner):
            self.inner = inner

        def act(self, obs):
            seen.append(obs["observation"].dtype)
            return self.inner.act(obs)

    wrap.set_opponent(Spy(OPP[path.split("_")[-2]]()))
    replay_selfplay_trace(wrap, log, lambda w, sides: w.force_sides(torch.from_numpy(sides.astype(np.int64))))
    assert seen and all(d == obs_dtype for d in seen)


# ----------------------------------------------------------------------------- the reference's own tests
@pytest.fixture
def wrapper_factory(hip):
    """src/tests/test_mnk_integration.py:27-42"""

    def _create(opponent_action_idx=0):
        env = hip.Env(m=3, n=3, k=3, num_envs=1, device=DEV)
        wrapper = hip.Wrapper(env)
        wrapper.set_opponent(FixedCellPolicy(opponent_action_idx))
        return wrapper

    return _create


def test_reference_canonical_view(wrapper_factory):
    """src/tests/test_mnk_integration.py:89-114"""
    wrapper = wrapper_factory()
    wrapper.reset(options={"agent_side": 0})
    wrapper.env.boards[0, 0, 0, 0] = 1.0
    obs = wrapper.get_agent_obs()
    assert obs["observation"][0, 0, 0, 0] == 1.0
    wrapper.set_opponent(FixedCellPolicy(8))
    wrapper.reset(options={"agent_side": 1})
    wrapper.env.boards[0, 1, 0, 0] = 1.0
    obs = wrapper.get_agent_obs()
    assert obs["observation"][0, 0, 0, 0] == 1.0
    assert obs["observation"][0, 1, 2, 2] == 1.0


def test_reference_agent_win_reward(wrapper_factory):
    """src/tests/test_mnk_integration.py:117-132"""
    wrapper = wrapper_factory()
    wrapper.reset(options={"agent_side": 0})
    wrapper.env.boards[0, 0, 0, 0] = 1
    wrapper.env.boards[0, 0, 0, 1] = 1
    obs, rewards, terms, trunc, _ = wrapper.step(torch.tensor([2], device=wrapper.device))
    assert rewards[0].item() == 1.0
    assert terms[0].item() is True
    assert obs["observation"][0, 0].sum() == 3.0


def test_reference_opponent_win_penalty(wrapper_factory):
    """src/tests/test_mnk_integration.py:135-161"""
    wrapper = wrapper_factory(opponent_action_idx=5)
    wrapper.reset(options={"agent_side": 0})
    wrapper.env.boards[0, 0, 0, 0] = 1
    wrapper.env.boards[0, 0, 0, 1] = 1
    wrapper.env.boards[0, 1, 1, 0] = 1
    wrapper.env.boards[0, 1, 1, 1] = 1
    obs, rewards, terms, truncs, _ = wrapper.step(torch.tensor([6], device=wrapper.device))
    assert terms[0].item() is True
    assert rewards[0].item() == -1.0
    assert obs["observation"][0, 1, 1, :].sum() == 3.0


def test_reference_autoreset_next_step(wrapper_factory):
    """src/tests/test_mnk_integration.py:164-189"""
    wrapper = wrapper_factory()
    wrapper.reset(options={"agent_side": 0})
    wrapper.env.boards[0, 0, 0, 0] = 1
    wrapper.env.boards[0, 0, 0, 1] = 1
    obs, rewards, terms, _, _ = wrapper.step(torch.tensor([2], device=wrapper.device))
    assert terms[0].item() is True
    assert rewards[0].item() == 1.0
    assert obs["observation"][0, 0].sum() == 3.0
    wrapper.force_sides(0)
    obs_new, rewards_new, terms_new, _, _ = wrapper.step(torch.tensor([0], device=wrapper.device))
    assert terms_new[0].item() is False
    assert rewards_new[0].item() == 0.0
    assert obs_new["observation"][0, 0].sum() == 0.0


def test_reference_opponent_starts_after_reset(wrapper_factory):
    """src/tests/test_mnk_integration.py:192-207"""
    wrapper = wrapper_factory(opponent_action_idx=4)
    obs, _ = wrapper.reset(options={"agent_side": 1})
    assert obs["observation"][0, 0].sum() == 0.0
    assert obs["observation"][0, 1, 1, 1] == 1.0


# ----------------------------------------------------------------------------- fused random-opponent step
@pytest.mark.parametrize("m,n,k,nenv,steps", [(3, 3, 3, 200, 40), (9, 9, 5, 130, 150), (19, 19, 5, 64, 60),
                                              (4, 6, 3, 65, 50)])
def test_fused_random_opponent_step_matches_oracle(hip, m, n, k, nenv, steps):
    """mnk_selfplay_step_random (one launch per agent-step) == OracleSelfPlay with the Philox
    opponent and Philox sides; bit-exact state, obs, mask, rewards, terminated, sides."""
    seed, id0 = 77, 4096
    env = hip.Env(m, n, k, nenv, device=DEV)
    wrap = hip.Wrapper(env, seed=seed)
    wrap.env_id0 = id0
    wrap.set_opponent(hip.policy.RandomPolicy(m * n))

    ids = np.arange(id0, id0 + nenv, dtype=np.uint64)
    state = {"step": 0, "resetting": None}

    def sides(count):
        x = philox.rand_u32(seed, ids, state["step"], philox.STREAM_SIDE)
        s = torch.from_numpy(philox.draw_side(x))
        return s if count == nenv else s[torch.nonzero(state["resetting"]).squeeze(1)]

    ora = OracleSelfPlay(OracleVectorEnv(m, n, k, nenv), side_source=sides)
    opp = PhiloxOpponent(seed, id0)
    ora.set_opponent(opp)

    def same(o_hip, o_ora, t):
        assert torch.equal(o_hip["observation"].cpu(), o_ora["observation"]), f"obs {t}"
        assert torch.equal(o_hip["action_mask"].cpu(), o_ora["action_mask"]), f"mask {t}"
        assert torch.equal(wrap.agent_side.cpu(), ora.agent_side), f"sides {t}"
        assert np.array_equal(pack_boards(env.boards.cpu().numpy(), m, n), pack_boards(ora.env.boards.numpy(), m, n))
        assert torch.equal(env.current_player.cpu(), ora.env.current_player)
        assert torch.equal(env.move_counts.cpu(), ora.env.move_counts)

    o1, _ = wrap.reset()
    o2, _ = ora.reset()
    same(o1, o2, "reset")
    rng = np.random.default_rng(0)
    for t in range(steps):
        state["step"] = opp.step = t + 1
        state["resetting"] = ora.pending_resets.clone()
        mask = o2["action_mask"].numpy()
        acts = np.array([rng.choice(np.nonzero(row)[0]) for row in mask], dtype=np.int64)
        o1, r1, t1, tr1, _ = wrap.step(torch.from_numpy(acts).to(DEV))
        o2, r2, t2, tr2, _ = ora.step(torch.from_numpy(acts))
        assert torch.equal(r1.cpu(), r2) and torch.equal(t1.cpu(), t2), f"rewards / terminated {t}"
        assert torch.equal(wrap.pending_resets.cpu(), ora.pending_resets)
        same(o1, o2, t)


@pytest.mark.parametrize("fused", [True, False])
def test_full_size_selfplay_step_equals_the_oracle(hip, fused):
    """Bit-exact parity of the wrapper AT the BASELINE.json size (9x9x5, 65 536 envs): reset + 28 agent-steps against
    OracleSelfPlay -- the one-launch step with the built-in Philox opponent, and the two-launch step (k_selfplay_pre ->
    policy -> k_selfplay_post) with a row-local deterministic opponent: observation, mask, rewards, terminated, sides,
    pending resets and the env state of every env after every step.  The agent's move depends on the row (every env
    plays its own game); games end from the fifth step on, so the later steps cover terminations and the resets after
    them."""
    m, n, k, nenv, seed, steps = 9, 9, 5, 65536, 41, 28
    env = hip.Env(m, n, k, nenv, device=DEV)
    wrap = hip.Wrapper(env, seed=seed)
    ids = np.arange(nenv, dtype=np.uint64)
    state = {"step": 0, "resetting": None}

    def sides(count):
        s_ = torch.from_numpy(philox.draw_side(philox.rand_u32(seed, ids, state["step"], philox.STREAM_SIDE)))
        return s_ if count == nenv else s_[torch.nonzero(state["resetting"]).squeeze(1)]

    ora = OracleSelfPlay(OracleVectorEnv(m, n, k, nenv), side_source=sides)
    if fused:
        wrap.set_opponent(hip.policy.RandomPolicy(m * n))
        opp = PhiloxOpponent(seed, 0)
        ora.set_opponent(opp)
    else:
        opp = None
        wrap.set_opponent(MaskHashPolicy(5))
        ora.set_opponent(MaskHashPolicy(5))
    o1, _ = wrap.reset()
    o2, _ = ora.reset()
    agent = RowSaltedHashPolicy(11)  # deterministic, different in every row: the same agent moves on both sides
    finished = 0
    for t in range(steps + 1):
        assert torch.equal(o1["observation"].cpu(), o2["observation"]) and torch.equal(o1["action_mask"].cpu(), o2["action_mask"]), t
        assert torch.equal(wrap.agent_side.cpu(), ora.agent_side) and torch.equal(wrap.pending_resets.cpu(), ora.pending_resets), t
        assert torch.equal(env._meta.cpu() & 1, ora.env.current_player.to(torch.int32)), t
        if t == steps:
            break
        state["step"] = t + 1
        if opp is not None:
            opp.step = t + 1
        state["resetting"] = ora.pending_resets.clone()
        acts = agent.act(o2)
        o1, r1, t1, _, _ = wrap.step(acts.to(DEV))
        o2, r2, t2, _, _ = ora.step(acts)
        assert torch.equal(r1.cpu(), r2) and torch.equal(t1.cpu(), t2), t
        finished += int(t2.sum())
    assert np.array_equal(pack_boards(env.boards.cpu().numpy(), m, n), pack_boards(ora.env.boards.numpy(), m, n))
    assert torch.equal(env.move_counts.cpu(), ora.env.move_counts)
    assert finished > 100, f"only {finished} games ended in the compared steps: terminations and resets went untested"


def test_two_launch_path_equals_fused_path(hip):
    """pre + policy + post with a policy that reproduces the Philox picks == the one-launch path."""
    m, n, k, nenv, seed = 9, 9, 5, 257, 5
    a_env, b_env = hip.Env(m, n, k, nenv, device=DEV), hip.Env(m, n, k, nenv, device=DEV)
    fused, split = hip.Wrapper(a_env, seed=seed), hip.Wrapper(b_env, seed=seed)
    fused.set_opponent(hip.policy.RandomPolicy(m * n))

    class SamePicks:
        def act(self, obs):
            acts = torch.empty(nenv, dtype=torch.long, device=DEV)
            b_env.sample_legal_into(acts, seed=seed, step=split.step_count - 1, env_id0=0,
                                    stream_id=hip.lib.STREAM_OPP)
            return acts

    split.set_opponent(SamePicks())
    o1, _ = fused.reset()
    o2, _ = split.reset()
    rng = torch.Generator(device="cpu").manual_seed(0)
    for t in range(120):
        assert torch.equal(o1["observation"], o2["observation"]) and torch.equal(o1["action_mask"], o2["action_mask"])
        acts = torch.multinomial(o1["action_mask"].float().cpu(), 1, generator=rng).squeeze(1).to(DEV)
        o1, r1, t1, _, _ = fused.step(acts)
        o2, r2, t2, _, _ = split.step(acts)
        assert torch.equal(r1, r2) and torch.equal(t1, t2)
        assert torch.equal(fused.agent_side, split.agent_side)
    assert set(r1.unique().tolist()) <= {-1.0, 0.0, 1.0}


# ----------------------------------------------------------------------------- masked-logits sampler
def test_masked_logits_head_matches_reference(hip, golden_dir):
    """G6 (epilogue): raw logits + mask -> argmax and log-prob of the reference's masked Categorical
    (cnn.py:69-79).  Tolerance 1e-5 on log-probabilities (f32 logsumexp, different summation order)."""
    g = np.load(f"{golden_dir}/masked_logits.npz")
    sampler = hip.policy._HipSampler(seed=1)
    for arch in ("cnn_b_s", "resnet_b_s"):
        raw = torch.from_numpy(g[arch + "_raw_logits"]).to(DEV)
        mask = torch.from_numpy(g[arch + "_mask"]).to(DEV)
        want = g[arch + "_masked_logits"]
        act, logp = sampler.draw(raw, mask, deterministic=True, want_logp=True)
        act, logp = act.cpu().numpy(), logp.cpu().numpy()
        assert np.array_equal(act, want.argmax(axis=1))
        assert np.allclose(logp, want[np.arange(len(act)), act], atol=1e-5, rtol=0)
        # sampled actions are always legal (or anything on the all-masked row), with the right log-prob
        act, logp = sampler.draw(raw, mask, deterministic=False, want_logp=True)
        act, logp = act.cpu().numpy(), logp.cpu().numpy()
        rows = np.nonzero(g[arch + "_mask"].any(axis=1))[0]
        assert g[arch + "_mask"][rows, act[rows]].all()
        assert np.allclose(logp, want[np.arange(len(act)), act], atol=1e-5, rtol=0)


@pytest.mark.parametrize("arch,dtype", [("cnn_b_s", torch.float32), ("resnet_b_s", torch.float32), ("cnn_b_s", torch.bfloat16)])
def test_sampler_distribution_chi_square(hip, golden_dir, arch, dtype):
    """Distributional parity of the inverse-CDF draw with the reference's probabilities (chi-square
    on >= 1e6 draws): 8 rows of masked logits x 250 000 draws each = 2e6 draws, every row against the probabilities the
    reference's masked Categorical holds for it; threshold = the 99.9 % quantile per row, Bonferroni-corrected over
    the rows.  bf16 logits are tested against the softmax of the bf16-rounded logits."""
    from scipy.stats import chi2

    g = np.load(f"{golden_dir}/masked_logits.npz")
    masks = g[arch + "_mask"]
    rows = [r for r in range(masks.shape[0]) if masks[r].sum() >= 2][3::max(1, masks.shape[0] // 9)][:8]
    assert len(rows) == 8
    draws = 250000
    sampler = hip.policy._HipSampler(seed=9)
    for j, row in enumerate(rows):
        raw = torch.from_numpy(g[arch + "_raw_logits"][row]).to(DEV).to(dtype)
        mask = torch.from_numpy(masks[row]).to(DEV)
        if dtype == torch.float32:
            probs = g[arch + "_probs"][row].astype(np.float64)
        else:
            lg = raw.float().cpu().numpy().astype(np.float64)
            w = np.where(masks[row], np.exp(lg - lg[masks[row]].max()), 0.0)
            probs = w / w.sum()
        # independent draws of the same row: Philox is keyed by the row's env id
        acts = sampler.draw(raw.expand(draws, -1).contiguous(), mask.expand(draws, -1), deterministic=False).cpu().numpy()
        counts = np.bincount(acts, minlength=len(probs)).astype(np.float64)
        assert counts[probs == 0].sum() == 0
        keep = probs * draws >= 5  # the usual validity rule of the test; the rest is pooled into one cell
        rest_p, rest_c = probs[~keep].sum(), counts[~keep].sum()
        stat = (((counts - draws * probs) ** 2)[keep] / (draws * probs[keep])).sum()
        dof = int(keep.sum()) - 1
        if rest_p * draws >= 5:
            stat += (rest_c - draws * rest_p) ** 2 / (draws * rest_p)
            dof += 1
        assert stat < chi2.ppf(1 - 0.001 / len(rows), dof), (arch, row, stat, dof)


def test_random_policy_is_uniform_over_legal(hip):
    pol = hip.policy.RandomPolicy(9, seed=3)
    mask = torch.zeros((90000, 9), dtype=torch.bool, device=DEV)
    mask[:, [0, 4, 5]] = True
    acts = pol.act({"action_mask": mask}).cpu().numpy()
    counts = np.bincount(acts, minlength=9)
    assert counts[[0, 4, 5]].sum() == 90000 and np.all(np.abs(counts[[0, 4, 5]] - 30000) < 700)
    assert pol.act({"action_mask": mask}, deterministic=True).unique().tolist() == [0]
    empty = torch.zeros((4000, 9), dtype=torch.bool, device=DEV)
    acts = pol.act({"action_mask": empty}).cpu().numpy()  # policy.py:21-24: uniform over all cells
    assert set(acts.tolist()) == set(range(9))


# ----------------------------------------------------------------------------- records and GAE
@pytest.mark.parametrize("m,n,k,nenv,steps", [(3, 3, 3, 70, 12), (9, 9, 5, 200, 30), (19, 19, 5, 33, 20)])
def test_unpack_records_matches_numpy(hip, m, n, k, nenv, steps):
    env = hip.Env(m, n, k, nenv, device=DEV)
    rec = hip.rollout.RandomRollout(env, seed=2).run(steps)
    buf = hip.rollout.unpack_records(rec, env)
    planes = rec.planes.cpu().numpy().view(np.uint64)
    meta = rec.meta.cpu().numpy().view(np.uint32)
    side = (meta >> 25) & 1
    for t in range(steps):
        dense = unpack_boards(planes_from_record_rows(planes[t], m, n, side[t]), m, n)  # absolute planes
        flip = side[t] == 1
        dense[flip] = dense[flip][:, ::-1]
        assert np.array_equal(buf["observations"][t].cpu().numpy(), dense)
        legal = ~(dense != 0).any(axis=1).reshape(nenv, m * n)
        assert np.array_equal(buf["action_masks"][t].cpu().numpy(), legal)
    assert np.array_equal(buf["actions"].cpu().numpy(), (meta & 0xFFFF).astype(np.int64))
    assert np.array_equal(buf["rewards"].cpu().numpy(), ((meta >> 16) & 0xFF).astype(np.int8).astype(np.float32))
    assert np.array_equal(buf["dones"].cpu().numpy(), ((meta >> 24) & 1).astype(bool))


def test_gae_matches_reference_arithmetic(hip):
    """mnk_gae == rollout_buffer.py:60-80 restated in f32 numpy, bit for bit."""
    rng = np.random.default_rng(0)
    t, n = 64, 1000
    rewards = rng.choice([-1.0, 0.0, 1.0], size=(t, n)).astype(np.float32)
    values = rng.standard_normal((t, n)).astype(np.float32)
    dones = rng.random((t, n)) < 0.05
    last = rng.standard_normal(n).astype(np.float32)
    adv, ret = hip.rollout.gae(torch.from_numpy(rewards).to(DEV), torch.from_numpy(values).to(DEV),
                               torch.from_numpy(dones).to(DEV), torch.from_numpy(last).to(DEV), 0.99, 0.95)
    want_adv, want_ret = oracle_gae(rewards, values, dones, last, 0.99, 0.95)
    assert np.array_equal(adv.cpu().numpy(), want_adv)
    assert np.array_equal(ret.cpu().numpy(), want_ret)


# ----------------------------------------------------------------------------- callers
def test_validate_gpu_runs_and_counts(hip):
    """selfplay/validation.py:6-44 on the HIP wrapper: random vs random on 3x3x3."""
    res = hip.validation.validate_gpu(hip.policy.RandomPolicy(9, seed=1), hip.policy.RandomPolicy(9, seed=2),
                                      (3, 3, 3), n_episodes=4096, device=DEV)
    w, l, d = (res[f"validation/vs_benchmark/{k}_rate"] for k in ("win", "loss", "draw"))
    assert abs(w + l + d - 1.0) < 1e-9 and res["validation/vs_benchmark/games_played"] == 4096
    ref = random_play_stats("3x3x3")["draw_rate"]  # the imported reference's random play: tests/golden/make_golden_stats.py
    assert abs(d - ref) < 4 * (ref * (1 - ref) / 4096) ** 0.5
    assert 0.3 < w < 0.6 and 0.3 < l < 0.6


def test_ppo_rollout_call_pattern(hip):
    """The call sequence of alg/ppo.py:81-124 -- reset once, then net(obs, mask) -> sample -> step ->
    keep obs for the next iteration -- runs on the HIP wrapper with a torch network in the loop."""
    m, n, k, nenv = 9, 9, 5, 512
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(2 * m * n, m * n)).to(DEV)
    wrap = hip.Wrapper(hip.Env(m, n, k, nenv, device="cuda"))
    wrap.set_opponent(hip.policy.RandomPolicy(m * n))
    obs, _ = wrap.reset()
    ep_reward = torch.zeros(nenv, device=DEV)
    finished = 0
    for _ in range(100):
        observation, action_mask = obs["observation"], obs["action_mask"]
        assert observation.shape == (nenv, 2, m, n) and action_mask.shape == (nenv, m * n)
        assert bool(action_mask.any(dim=1).all())  # Categorical stays valid (wrapper:108-110)
        with torch.no_grad():
            logits = torch.where(action_mask, net(observation), torch.tensor(-torch.inf, device=DEV))
            dist = torch.distributions.Categorical(logits=logits)
            actions = dist.sample()
        next_obs, rewards, terminateds, truncateds, _ = wrap.step(actions)
        dones = terminateds | truncateds
        ep_reward += rewards
        finished += int(dones.sum())
        # the observation handed out at t is still intact after the step (ppo.py:106 stores it afterwards)
        assert torch.equal(observation, obs["observation"])
        obs = next_obs
    assert finished > nenv  # games end and restart
    assert set(rewards.unique().tolist()) <= {-1.0, 0.0, 1.0}


def test_rollout_buffer_dropin(hip):
    """alg/rollout_buffer.py surface: add / full-buffer error / GAE (bit-exact vs the f32 restatement of
    rollout_buffer.py:60-80) / minibatch generator, including a partially filled buffer."""
    from alg.rollout_buffer import RolloutBuffer

    t, n, c = 16, 300, 9
    buf = RolloutBuffer(t, n, (2, 3, 3), c, device=DEV)
    rng = np.random.default_rng(1)
    rows = []
    for step in range(11):  # partially filled: ptr = 11 < n_steps
        obs = torch.from_numpy(rng.integers(0, 2, (n, 2, 3, 3)).astype(np.float32)).to(DEV)
        act = torch.from_numpy(rng.integers(0, c, n)).to(DEV)
        rew = torch.from_numpy(rng.choice([-1.0, 0.0, 1.0], n).astype(np.float32)).to(DEV)
        val = torch.from_numpy(rng.standard_normal((n, 1)).astype(np.float32)).to(DEV)
        lp = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(DEV)
        done = torch.from_numpy(rng.random(n) < 0.1).to(DEV)
        mask = torch.from_numpy(rng.random((n, c)) < 0.7).to(DEV)
        buf.add(obs, act, rew, val, lp, done, mask)
        rows.append((rew.cpu().numpy(), val.cpu().numpy().reshape(-1), done.cpu().numpy()))
    assert buf.ptr == 11
    last = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(DEV)
    buf.compute_advantages_and_returns(last, 0.99, 0.95)
    rewards, values, dones = (np.stack(x) for x in zip(*rows))
    want_adv, want_ret = oracle_gae(rewards, values, dones, last.cpu().numpy(), 0.99, 0.95)
    assert np.array_equal(buf.advantages[:11].cpu().numpy(), want_adv)
    assert np.array_equal(buf.returns[:11].cpu().numpy(), want_ret)
    assert float(buf.advantages[11:].abs().sum()) == 0.0
    seen = 0
    for obs_b, act_b, lp_b, ret_b, adv_b, mask_b, val_b in buf.get_data_loader(1000):
        assert obs_b.shape[1:] == (2, 3, 3) and mask_b.shape[1] == c and act_b.dtype == torch.long
        seen += obs_b.shape[0]
    assert seen == 11 * n
    for _ in range(5):
        buf.add(obs, act, rew, val, lp, done, mask)
    with pytest.raises(IndexError, match="Buffer was full."):
        buf.add(obs, act, rew, val, lp, done, mask)
    buf.reset()
    assert buf.ptr == 0 and float(buf.rewards.abs().sum()) == 0.0


def test_full_size_selfplay_properties(hip):
    """BASELINE.json size (9x9x5, 65 536 envs), fused self-play step with the random opponent: properties
    that need no oracle run.  Every step: rewards in {-1, 0, +1} and non-zero only on terminated envs; the
    returned view is the env state seen from the agent's side with mask == free cells; a terminated env is
    reset by the next step (reward 0, not terminated, at most one opponent stone on the board); stone counts
    of the two sides never differ by more than one; and the long-run outcome of random-vs-random play is
    symmetric between agent and opponent."""
    m, n, k, nenv = 9, 9, 5, 65536
    env = hip.Env(m, n, k, nenv, device=DEV)
    wrap = hip.Wrapper(env, seed=4)
    wrap.set_opponent(hip.policy.RandomPolicy(m * n))
    agent = hip.policy.RandomPolicy(m * n, seed=8)
    obs, _ = wrap.reset()
    wins = losses = draws = 0
    prev_term = torch.zeros(nenv, dtype=torch.bool, device=DEV)
    for t in range(120):
        acts = agent.act(obs)
        assert bool(torch.gather(obs["action_mask"], 1, acts.unsqueeze(1)).all())
        obs, rew, term, trunc, _ = wrap.step(acts)
        assert not bool(trunc.any())
        assert bool(((rew == 0) | (rew == 1) | (rew == -1)).all()) and not bool((rew != 0)[~term].any())
        assert torch.equal(wrap.pending_resets, term)
        # freshly reset envs: no reward, not terminated, board holds 0 or 1 stones (the opponent's opening)
        fresh = prev_term
        assert not bool(term[fresh].any()) and not bool((rew[fresh] != 0).any())
        stones = obs["observation"].sum(dim=(2, 3))
        assert bool((stones[fresh, 0] == 0).all()) and bool((stones[fresh, 1] <= 1).all())
        # the view is the state from the agent's side
        dense = env.boards[...]
        flip = wrap.agent_side == 1
        want = torch.where(flip.view(-1, 1, 1, 1), dense.flip(1), dense)
        assert torch.equal(obs["observation"], want)
        free = (dense.sum(dim=1) == 0).reshape(nenv, -1)
        free[free.sum(dim=1) == 0, 0] = True
        assert torch.equal(obs["action_mask"], free)
        assert bool((stones[:, 0] - stones[:, 1]).abs().max() <= 1)
        wins += int((rew == 1).sum()); losses += int((rew == -1).sum()); draws += int((term & (rew == 0)).sum())
        prev_term = term
    games = wins + losses + draws
    assert games > 100000
    assert abs(wins - losses) / games < 0.02      # same policy on both sides, sides drawn uniformly
    assert draws / games < 0.01


def test_full_size_two_launch_path_with_a_rotating_opponent_pool(hip):
    """BASELINE.json config 3's actual path at full size (9x9x5, 65 536 envs): ``k_selfplay_pre`` -> opponent network
    forward + fused masked draw -> ``k_selfplay_post``, opponents = ``FusedNNPolicy`` nets from an ``OpponentPool(4)``
    swapped every 16 steps (src/train.py:106-114, src/selfplay/opponent_pool.py:5-19), every step written into a
    RolloutBuffer through the sink.  Checks the property set of ``test_full_size_selfplay_properties`` on this path;
    then, with a policy that replays the Philox picks of the built-in random opponent, that
    pre + policy + post == the one-launch ``k_selfplay_step_random`` at the same size (the 257-env test at 65 536)."""
    import random

    import torch.nn as nn

    from alg.rollout_buffer import RolloutBuffer
    from selfplay.opponent_pool import OpponentPool

    m, n, k, nenv, c, steps = 9, 9, 5, 65536, 81, 64

    class Net(nn.Module):  # a small conv policy of the reference's shape: conv trunk, 1x1-conv policy head, value head
        def __init__(self):
            super().__init__()
            self.body = nn.Sequential(nn.Conv2d(2, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.ReLU())
            self.pi, self.v = nn.Conv2d(8, 1, 1), nn.Linear(8 * c, 1)

        def forward(self, obs, action_mask=None):
            h = self.body(obs)
            logits = self.pi(h).flatten(1)
            if action_mask is not None:
                logits = torch.where(action_mask, logits, torch.full_like(logits, -torch.inf))
            return torch.distributions.Categorical(logits=logits, validate_args=False), torch.tanh(self.v(h.flatten(1)))

    torch.manual_seed(3)
    random.seed(3)
    pool = OpponentPool(max_size=4)
    for j in range(5):  # five additions into a pool of four: the oldest falls out (opponent_pool.py:8)
        pool.add_opponent(hip.policy.FusedNNPolicy(Net().to(DEV), seed=100 + j))
    assert pool.size() == 4
    env = hip.Env(m, n, k, nenv, device=DEV)
    wrap = hip.Wrapper(env, seed=4)
    wrap.set_opponent(pool.get_random_opponent())
    buf = RolloutBuffer(steps, nenv, (2, m, n), c, device=DEV)
    wrap.attach_sink(buf)
    agent = hip.policy.RandomPolicy(c, seed=8)
    obs, _ = wrap.reset()
    zeros = torch.zeros(nenv, device=DEV)
    wins = losses = draws = 0
    used = set()
    prev_term = torch.zeros(nenv, dtype=torch.bool, device=DEV)
    for t in range(steps):
        if t % 16 == 0:
            wrap.set_opponent(pool.get_random_opponent())  # train.py:112-114
        used.add(id(wrap.opponent_policy))
        acts = agent.act(obs)
        assert bool(torch.gather(obs["action_mask"], 1, acts.unsqueeze(1)).all())
        nxt, rew, term, trunc, _ = wrap.step(acts)
        buf.add(obs["observation"], acts, rew, zeros.view(-1, 1), zeros, term | trunc, obs["action_mask"])
        obs = nxt
        assert not bool(trunc.any())
        assert bool(((rew == 0) | (rew == 1) | (rew == -1)).all()) and not bool((rew != 0)[~term].any())
        assert torch.equal(wrap.pending_resets, term)
        fresh = prev_term
        assert not bool(term[fresh].any()) and not bool((rew[fresh] != 0).any())
        stones = obs["observation"].sum(dim=(2, 3))
        assert bool((stones[fresh, 0] == 0).all()) and bool((stones[fresh, 1] <= 1).all())
        dense = env.boards[...]
        flip = wrap.agent_side == 1
        assert torch.equal(obs["observation"], torch.where(flip.view(-1, 1, 1, 1), dense.flip(1), dense))
        free = (dense.sum(dim=1) == 0).reshape(nenv, -1)
        free[free.sum(dim=1) == 0, 0] = True
        assert torch.equal(obs["action_mask"], free)
        assert bool((stones[:, 0] - stones[:, 1]).abs().max() <= 1)
        wins += int((rew == 1).sum()); losses += int((rew == -1).sum()); draws += int((term & (rew == 0)).sum())
        prev_term = term
    assert len(used) >= 2 and wins + losses + draws > 20000
    assert buf.ptr == steps and buf.copied_bytes == steps * nenv * (8 + 4 + 4 + 1)  # observations / masks / rewards in place
    # the buffer holds the trajectory: row t+1's observation differs from row t's by the stones of one step (or a reset)
    # (a step after a terminated one is the reset: a fresh board with at most the opponent's opening stone)
    stones = buf.observations.sum(dim=(2, 3, 4))
    grown = stones[1:] - stones[:-1]
    was_reset = torch.zeros_like(buf.dones[:-1])
    was_reset[1:] = buf.dones[:-2]
    assert bool((((grown >= 1) & (grown <= 2) & ~was_reset) | (was_reset & (stones[1:] <= 1))).all())
    del buf, stones, grown

    # pre + policy + post == the one-launch step when the policy replays the built-in opponent's Philox picks
    seed = 5
    a_env, b_env = hip.Env(m, n, k, nenv, device=DEV), hip.Env(m, n, k, nenv, device=DEV)
    fused, split = hip.Wrapper(a_env, seed=seed), hip.Wrapper(b_env, seed=seed)
    fused.set_opponent(hip.policy.RandomPolicy(c))

    class SamePicks:
        def act(self, obs):
            acts = torch.empty(nenv, dtype=torch.long, device=DEV)
            b_env.sample_legal_into(acts, seed=seed, step=split.step_count - 1, env_id0=0, stream_id=hip.lib.STREAM_OPP)
            return acts

    split.set_opponent(SamePicks())
    o1, _ = fused.reset()
    o2, _ = split.reset()
    for t in range(100):
        assert torch.equal(o1["observation"], o2["observation"]) and torch.equal(o1["action_mask"], o2["action_mask"]), t
        acts = agent.act(o1)
        o1, r1, t1, _, _ = fused.step(acts)
        o2, r2, t2, _, _ = split.step(acts)
        assert torch.equal(r1, r2) and torch.equal(t1, t2) and torch.equal(fused.agent_side, split.agent_side), t
    assert torch.equal(a_env._planes, b_env._planes) and torch.equal(a_env._meta, b_env._meta)


@pytest.mark.parametrize("fused", [True, False])
def test_device_episode_stats_equal_the_ppo_host_loop(hip, fused):
    """track_episodes(): the device-side counters == the bookkeeping of alg/ppo.py:104-120 done on the host
    (current_ep_reward += rewards; current_ep_len += 1; on done: record and zero)."""
    m, n, k, nenv = 3, 3, 3, 1000
    env = hip.Env(m, n, k, nenv, device=DEV)
    wrap = hip.Wrapper(env, seed=6)
    opp = hip.policy.RandomPolicy(m * n, seed=1)
    if not fused:
        opp.fused_uniform_random = False  # instance attribute: take the pre / policy / post path
    wrap.set_opponent(opp)
    wrap.track_episodes()
    agent = hip.policy.RandomPolicy(m * n, seed=2)
    obs, _ = wrap.reset()
    ep_reward = torch.zeros(nenv, device=DEV)
    ep_len = torch.zeros(nenv, device=DEV)
    fin_rewards, fin_lengths = [], []
    for _ in range(60):
        obs, rew, term, trunc, _ = wrap.step(agent.act(obs))
        done = term | trunc
        ep_reward += rew
        ep_len += 1
        idx = torch.nonzero(done).squeeze(1)
        fin_rewards.extend(ep_reward[idx].tolist())
        fin_lengths.extend(ep_len[idx].tolist())
        ep_reward[idx] = 0
        ep_len[idx] = 0
    stats = wrap.pop_episode_stats()
    assert stats["episodes"] == len(fin_rewards) > 5000
    assert stats["wins"] == sum(r == 1.0 for r in fin_rewards)
    assert stats["losses"] == sum(r == -1.0 for r in fin_rewards)
    assert stats["draws"] == sum(r == 0.0 for r in fin_rewards)
    assert abs(stats["mean_length"] - np.mean(fin_lengths)) < 1e-9
    assert abs(stats["mean_reward"] - np.mean(fin_rewards)) < 1e-9
    assert wrap.pop_episode_stats()["episodes"] == 0  # popped
    assert torch.equal(wrap._ep_length.to(torch.float32), ep_len) and torch.equal(wrap._ep_return, ep_reward)


@pytest.mark.parametrize("m,n,k,p1_is_black", [(3, 3, 3, True), (3, 3, 3, False), (9, 9, 5, True), (4, 6, 3, False)])
def test_tournament_loop_equals_reference_bookkeeping(hip, m, n, k, p1_is_black):
    """selfplay/tournament.py (lockstep, one sync) against the reference's MatchRunner._play_batch_games
    algorithm (match_runner.py:149-215: active subsets, step_subset, first-finish counting) run on the oracle
    env with the very actions the HIP run drew."""
    from selfplay.tournament import play_batch_games

    games = 700
    log = []

    class Logged:
        def __init__(self, inner):
            self.inner = inner

        def act(self, obs, deterministic=False):
            a = self.inner.act(obs, deterministic)
            log.append((obs["observation"].cpu(), obs["action_mask"].cpu(), a.cpu()))
            return a

    p1, p2 = Logged(hip.policy.RandomPolicy(m * n, seed=1)), Logged(hip.policy.RandomPolicy(m * n, seed=2))
    got = play_batch_games(p1, p2, (m, n, k), games, p1_is_black, device=DEV)

    ora = OracleVectorEnv(m, n, k, games)
    obs = ora.reset()
    over = torch.zeros(games, dtype=torch.bool)
    agent_side = 0 if p1_is_black else 1
    wins = losses = draws = 0
    for ply, (seen_obs, seen_mask, acts) in enumerate(log):
        if bool(over.all()):
            break
        active = ~over
        is_p1 = (ora.current_player == agent_side) & active
        # what the running games' mover saw on the GPU is what the reference would have shown it
        view = obs["observation"].clone()
        white = ora.current_player == 1
        view[white] = view[white].flip(1)
        assert torch.equal(seen_obs[active], view[active]) and torch.equal(seen_mask[active], obs["action_mask"][active])
        idx = torch.nonzero(active).squeeze(1)
        obs, rew, done = ora.step_subset(acts[idx], idx)
        fresh = done & ~over
        won = (rew == 1.0) & fresh
        wins += int((won & is_p1).sum())
        losses += int((won & ~is_p1).sum())
        draws += int(((rew == 0.0) & fresh).sum())
        over |= fresh
    assert bool(over.all())
    assert got == (wins, losses, draws)
    assert sum(got) == games


@pytest.mark.parametrize("m,n,k", [(3, 3, 3), (9, 9, 5), (19, 19, 5)])
def test_packed_rollout_buffer_equals_the_dense_one(hip, m, n, k):
    """alg/packed_rollout_buffer.py: packed planes in, and the minibatch gather (mnk_gather_obs) hands back
    exactly the observations and masks the dense drop-in buffer stored; GAE identical."""
    from alg.packed_rollout_buffer import PackedRolloutBuffer
    from alg.rollout_buffer import RolloutBuffer

    nenv, steps, c = 300, 12, m * n
    env = hip.Env(m, n, k, nenv, device=DEV)
    wrap = hip.Wrapper(env, seed=9)
    wrap.set_opponent(hip.policy.RandomPolicy(c, seed=1))
    agent = hip.policy.RandomPolicy(c, seed=2)
    dense = RolloutBuffer(steps, nenv, (2, m, n), c, device=DEV)
    packed = PackedRolloutBuffer(steps, nenv, m, n, device=DEV)
    obs, _ = wrap.reset()
    for _ in range(steps - 2):  # a partly filled buffer
        pobs = wrap.packed_obs()
        acts = agent.act(obs)
        nobs, rew, term, trunc, _ = wrap.step(acts)
        val = torch.randn(nenv, 1, device=DEV)
        lp = torch.randn(nenv, device=DEV)
        dense.add(obs["observation"], acts, rew, val, lp, term | trunc, obs["action_mask"])
        packed.add(pobs, acts, rew, val, lp, term | trunc)
        obs = nobs
    last = torch.randn(nenv, device=DEV)
    dense.compute_advantages_and_returns(last)
    packed.compute_advantages_and_returns(last)
    assert torch.equal(dense.advantages, packed.advantages) and torch.equal(dense.returns, packed.returns)
    filled = (steps - 2) * nenv
    idx = torch.randperm(filled, device=DEV)[:1000]
    o, msk = packed.gather(idx)
    assert torch.equal(o, dense.observations.reshape(-1, 2, m, n)[idx])
    assert torch.equal(msk, dense.action_masks.reshape(-1, c)[idx])
    o, msk = packed.gather(torch.tensor([-1, 0, filled - 1], device=DEV))   # negative ids wrap
    assert torch.equal(o[1], dense.observations[0, 0]) and torch.equal(o[2], dense.observations[steps - 3, nenv - 1])
    seen = 0
    for b in packed.get_data_loader(777):
        assert b[0].shape[1:] == (2, m, n) and b[5].shape[1] == c and b[0].shape[0] == b[1].shape[0] == b[5].shape[0]
        assert bool(torch.gather(b[5], 1, b[1].unsqueeze(1)).all())  # the stored action was legal under the rebuilt mask
        seen += b[0].shape[0]
    assert seen == filled
    with pytest.raises(IndexError, match="Buffer was full."):
        for _ in range(3):
            packed.add(pobs, acts, rew, val, lp, term)


def test_reference_network_end_to_end(hip, golden_dir):
    """G6: a reference net (cnn_b_s, seed-0 weights stored as data) fed by the HIP path.  The canonical
    observation and mask built by the kernels are bit-identical to the reference's; the same weights on the GPU
    give the reference's masked log-probabilities and values within 1e-5 (fp32, eval mode; the only difference is
    MIOpen vs MKL-DNN summation order); NNPolicy / FusedNNPolicy pick the reference's argmax wherever its top-2
    gap exceeds 1e-4."""
    import torch.nn as nn

    g = np.load(f"{golden_dir}/policy_cnn_b_s.npz")
    m, n, k, nenv, c = 9, 9, 5, 64, 81

    class CnnBS(nn.Module):  # the layer list of the reference's cnn_b_s (alg/architectures/cnn.py:8-60, configs.py:49-56)
        def __init__(self):
            super().__init__()
            body, cin = [], 2
            for _ in range(4):
                body += [nn.Conv2d(cin, 56, 3, padding=1), nn.BatchNorm2d(56), nn.ReLU()]
                cin = 56
            self.shared_body = nn.Sequential(*body)
            self.actor = nn.Sequential(nn.Conv2d(56, 2, 1), nn.Flatten(), nn.LayerNorm(2 * c), nn.ReLU(),
                                       nn.Linear(2 * c, 128), nn.LayerNorm(128), nn.ReLU(), nn.Linear(128, c))
            self.critic = nn.Sequential(nn.Conv2d(56, 1, 1), nn.Flatten(), nn.LayerNorm(c), nn.ReLU(),
                                        nn.Linear(c, 128), nn.LayerNorm(128), nn.ReLU(), nn.Linear(128, 1), nn.Tanh())

        def forward(self, obs, action_mask=None):
            f = self.shared_body(obs)
            logits, value = self.actor(f), self.critic(f)
            if action_mask is not None:  # cnn.py:69-79
                logits = torch.where(action_mask.bool(), logits, torch.full_like(logits, -torch.inf))
                dead = logits.max(dim=1, keepdim=True)[0] == -torch.inf
                logits = torch.where(dead, torch.zeros_like(logits), logits)
            return torch.distributions.Categorical(logits=logits), value

    net = CnnBS()
    net.load_state_dict({key[len("param/"):]: torch.from_numpy(g[key]) for key in g.files if key.startswith("param/")})
    net = net.to(DEV).eval()

    env = hip.Env(m, n, k, nenv, device=DEV)
    env.boards = torch.from_numpy(unpack_boards(g["planes"], m, n))
    env.current_player = torch.from_numpy(g["meta_side"].astype(np.int64))
    env.move_counts = torch.from_numpy(g["meta_moves"].astype(np.int64))
    wrap = hip.Wrapper(env)
    wrap.agent_side.copy_(torch.from_numpy(g["agent_side"].astype(np.int64)))
    obs = wrap.get_agent_obs()
    assert np.array_equal(pack_boards(obs["observation"].cpu().numpy(), m, n), g["obs_planes"])
    assert np.array_equal(pack_cells(obs["action_mask"].cpu().numpy(), m, n), g["obs_mask"])

    prev = torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32
    torch.backends.cudnn.allow_tf32 = torch.backends.cuda.matmul.allow_tf32 = False
    try:
        with torch.no_grad():
            dist, value = net(obs["observation"], obs["action_mask"])
        logp, want = dist.logits.cpu().numpy(), g["logp"]
        legal = np.isfinite(want)
        assert np.array_equal(np.isfinite(logp), legal)
        assert np.abs(logp[legal] - want[legal]).max() < 1e-5
        assert np.abs(value.cpu().numpy() - g["value"]).max() < 1e-5
        top2 = np.sort(np.where(legal, want, -np.inf), axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 1e-4
        assert clear.sum() > 32
        for policy in (hip.policy.NNPolicy(net), hip.policy.FusedNNPolicy(net, seed=0)):
            act = policy.act(obs, deterministic=True).cpu().numpy()
            assert np.array_equal(act[clear], g["argmax"][clear])
            act = policy.act(obs).cpu().numpy()  # sampled actions are legal
            assert legal[np.arange(nenv), act].all()
    finally:
        torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32 = prev


@pytest.mark.parametrize("opponent", ["random", "nn"])
def test_graphed_agent_step_equals_eager(hip, opponent):
    """selfplay/graphed.py: the captured hipGraph of (net -> fused draw -> wrapper.step) replays to exactly
    what the same sequence does eagerly, step after step (the Philox step counter advances on the device)."""
    graphed_agent_step_against_eager(hip, opponent, (3, 3, 3))


def graphed_agent_step_against_eager(hip, opponent, board):
    """(tests/test_gpu_jit_api.py runs the same comparison on a board whose kernels are compiled at run time)"""
    import copy

    import torch.nn as nn

    from selfplay.graphed import GraphedAgentStep

    m, n, k = board
    nenv, c = 384, m * n

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.body = nn.Sequential(nn.Flatten(), nn.Linear(2 * c, 64), nn.Tanh())
            self.pi, self.v = nn.Linear(64, c), nn.Linear(64, 1)

        def forward(self, obs, action_mask=None):
            h = self.body(obs)
            logits = self.pi(h)
            if action_mask is not None:
                logits = torch.where(action_mask.bool(), logits, torch.full_like(logits, -torch.inf))
            return torch.distributions.Categorical(logits=logits, validate_args=False), torch.tanh(self.v(h))

    torch.manual_seed(0)
    net = Net().to(DEV).eval()
    opp_net = Net().to(DEV).eval()

    def make():
        w = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=5)
        w.set_opponent(hip.policy.RandomPolicy(c, seed=3) if opponent == "random"
                       else hip.policy.FusedNNPolicy(copy.deepcopy(opp_net), seed=3))
        return w

    graphed = GraphedAgentStep(make(), net, seed=11)
    # eager twin: same seeds, the step counters passed by value
    w = make()
    obs, _ = w.reset()
    sampler = hip.policy._HipSampler(seed=11)
    warm = int(graphed.step_dev.item())   # steps the collector already played while warming up / capturing
    for t in range(warm + 40):
        with torch.no_grad():
            dist, values = net(obs["observation"], None)
        if opponent == "nn":
            w.opponent_policy._sampler.calls = t + 1  # the reset's reply used draw 0
        sampler.calls = t
        actions, logp = sampler.draw(dist.logits, obs["action_mask"], False, want_logp=True)
        prev = obs
        obs, rew, term, trunc, _ = w.step(actions)
        if t >= warm:
            out = graphed.step()
            assert torch.equal(out["obs"], prev["observation"]) and torch.equal(out["mask"], prev["action_mask"]), t
            assert torch.equal(out["actions"], actions) and torch.equal(out["rewards"], rew), t
            assert torch.equal(out["terminated"], term) and torch.allclose(out["log_probs"], logp), t
    assert torch.equal(graphed.current_obs()["observation"], obs["observation"])


# ----------------------------------------------------------------------------- round-2 additions
def test_unseeded_policies_draw_independently(hip):
    """Policies built with the reference's signatures (no seed: policy.py:14, :33) must not share a random
    stream: two RandomPolicy instances on the same mask disagree, and so do an unseeded wrapper's side draws
    from another's."""
    c, rows = 81, 20000
    mask = torch.ones(rows, c, dtype=torch.bool, device=DEV)
    a, b = hip.policy.RandomPolicy(c), hip.policy.RandomPolicy(c)
    assert a._sampler.seed != b._sampler.seed
    xa, xb = a.act({"action_mask": mask}), b.act({"action_mask": mask})
    same = float((xa == xb).float().mean())
    assert abs(same - 1.0 / c) < 0.01, same          # independent uniform draws agree with probability 1/81
    corr = float(torch.corrcoef(torch.stack([xa.float(), xb.float()]))[0, 1])
    assert abs(corr) < 0.03, corr
    # a second call of the same policy moves on as well
    assert float((a.act({"action_mask": mask}) == xa).float().mean()) < 0.05
    w1, w2 = hip.Wrapper(hip.Env(3, 3, 3, 4096, device=DEV)), hip.Wrapper(hip.Env(3, 3, 3, 4096, device=DEV))
    for w in (w1, w2):
        w.set_opponent(hip.policy.RandomPolicy(9))
        w.reset()
    agree = float((w1.agent_side == w2.agent_side).float().mean())
    assert 0.4 < agree < 0.6, agree
    # explicit seeds stay reproducible
    p, q = hip.policy.RandomPolicy(c, seed=5), hip.policy.RandomPolicy(c, seed=5)
    assert torch.equal(p.act({"action_mask": mask}), q.act({"action_mask": mask}))


@pytest.mark.parametrize("fused", [True, False])
def test_strict_env_is_strict_through_the_wrapper(hip, fused):
    """strict=True refuses occupied cells on the agent's ply inside wrapper.step as well (message of the
    reference's _validate_moves, env/torch_vector_mnk_env.py:94-104); the offending env is left untouched."""
    env = hip.Env(3, 3, 3, 4, device=DEV, strict=True)
    wrap = hip.Wrapper(env, seed=1)
    opp = hip.policy.RandomPolicy(9, seed=2) if fused else LowestLegalPolicy()
    wrap.set_opponent(opp)
    wrap.reset(options={"agent_side": 0})
    obs, *_ = wrap.step(torch.tensor([4, 4, 4, 4], device=DEV))           # agent takes the centre, opponent replies
    before = env.boards[...].clone()
    with pytest.raises(ValueError, match="Illegal Move: Env 2 tried to play in occupied cell."):
        acts = torch.argmax(obs["action_mask"].to(torch.uint8), dim=1)
        acts[2] = 4                                                           # occupied
        wrap.step(acts)
    after = env.boards[...]
    assert torch.equal(after[2], before[2])                                  # untouched
    assert int(after[0].sum()) == int(before[0].sum()) + 2                   # the others played on
    # a non-strict env accepts the same move like the reference does (both planes may end up set, env:67-69)
    env2 = hip.Env(3, 3, 3, 4, device=DEV)
    wrap2 = hip.Wrapper(env2, seed=1)
    wrap2.set_opponent(hip.policy.RandomPolicy(9, seed=2) if fused else LowestLegalPolicy())
    wrap2.reset(options={"agent_side": 0})
    wrap2.step(torch.tensor([4, 4, 4, 4], device=DEV))
    wrap2.step(torch.tensor([4, 4, 4, 4], device=DEV))
    env2.check_errors()


def test_forced_sides_must_cover_every_env(hip):
    wrap = hip.Wrapper(hip.Env(3, 3, 3, 8, device=DEV), seed=1)
    wrap.set_opponent(LowestLegalPolicy())
    with pytest.raises(IndexError, match="3 sides for 8 envs"):
        wrap.force_sides(torch.tensor([0, 1, 0]))
    with pytest.raises(IndexError, match="sides for 8 envs"):
        wrap.reset(options={"agent_side": torch.zeros(9, dtype=torch.long)})
    wrap.reset(options={"agent_side": 1})
    assert bool((wrap.agent_side == 1).all())


def test_out_of_range_action_surfaces_at_pop_episode_stats(hip):
    """non-strict env: an action outside [-C, C) leaves the env untouched and sets the sticky device word; the
    periodic pop_episode_stats() (one sync anyway) raises it -- the reference raises an IndexError at once."""
    wrap = hip.Wrapper(hip.Env(3, 3, 3, 4, device=DEV), seed=1)
    wrap.set_opponent(hip.policy.RandomPolicy(9, seed=2))
    wrap.track_episodes()
    wrap.reset(options={"agent_side": 0})
    wrap.step(torch.tensor([0, 1, 99, 2], device=DEV))
    with pytest.raises(IndexError, match="index 2 is out of bounds"):
        wrap.pop_episode_stats()


def test_bf16_logits_are_sampled_without_an_f32_copy(hip, golden_dir):
    """mnk_sample_logits(MNK_LOGITS_BF16): argmax and log-prob equal torch's masked Categorical on the same
    bf16-rounded logits (f32 arithmetic on both sides, 1e-5), for the reference nets' raw logits
    (cnn.py:69-80 under the autocast of ppo.py:194)."""
    data = np.load(f"{golden_dir}/masked_logits.npz")
    sampler = hip.policy._HipSampler(seed=4)
    for arch in ("cnn_b_s", "resnet_b_s"):
        raw = torch.from_numpy(data[arch + "_raw_logits"]).to(DEV).to(torch.bfloat16)
        mask = torch.from_numpy(data[arch + "_mask"]).to(DEV)
        act, logp = sampler.draw(raw, mask, True, want_logp=True)
        ref_logits = torch.where(mask, raw.float(), torch.full_like(raw.float(), -torch.inf))
        dead = ~mask.any(dim=1)
        ref_logits[dead] = 0.0                                    # cnn.py:76-77
        ref = torch.distributions.Categorical(logits=ref_logits)
        assert torch.equal(act, torch.argmax(ref.logits, dim=1))
        assert torch.allclose(logp, ref.log_prob(act), atol=1e-5, rtol=0)
        # draws: legal, and their log-prob is the reference's
        act, logp = sampler.draw(raw, mask, False, want_logp=True)
        assert bool(mask[~dead].gather(1, act[~dead].unsqueeze(1)).all())
        assert torch.allclose(logp, ref.log_prob(act), atol=1e-5, rtol=0)
    # a strided / odd-offset view still works (the unaligned scalar path of the kernel)
    big = torch.randn(1001, 83, device=DEV)
    view, m = big[1:, 1:82], torch.rand(1000, 81, device=DEV) > 0.3
    m[:, 0] = True
    a1, l1 = sampler.draw(view, m, True, want_logp=True)
    ref = torch.distributions.Categorical(logits=torch.where(m, view, torch.full_like(view, -torch.inf)))
    assert torch.equal(a1, torch.argmax(ref.logits, dim=1)) and torch.allclose(l1, ref.log_prob(a1), atol=1e-5, rtol=0)


@pytest.mark.parametrize("m,n", [(3, 3), (4, 6), (7, 9), (9, 9), (10, 10), (13, 13), (15, 15), (19, 19), (22, 22), (25, 25), (31, 31)])
def test_sampler_all_row_widths(hip, m, n):
    """every (lanes per row, cells per lane) variant of k_sample_logits, ragged last workgroup included: argmax,
    log-prob and legality of the draws against torch's masked Categorical"""
    c, rows = m * n, 1237
    g = torch.Generator(device="cpu").manual_seed(c)
    logits = (torch.randn(rows, c, generator=g) * 3).to(DEV)
    mask = (torch.rand(rows, c, generator=g) > 0.4).to(DEV)
    mask[5] = False                                              # an all-masked row
    mask[7] = False
    mask[7, c - 1] = True                                        # a single legal cell, the last one
    sampler = hip.policy._HipSampler(seed=c)
    ref_logits = torch.where(mask, logits, torch.full_like(logits, -torch.inf))
    ref_logits[5] = 0.0
    ref = torch.distributions.Categorical(logits=ref_logits)
    act, logp = sampler.draw(logits, mask, True, want_logp=True)
    assert torch.equal(act, torch.argmax(ref.logits, dim=1))
    assert torch.allclose(logp, ref.log_prob(act), atol=1e-5, rtol=0)
    act, logp = sampler.draw(logits, mask, False, want_logp=True)
    legal = mask.clone()
    legal[5] = True
    assert bool(legal.gather(1, act.unsqueeze(1)).all()) and int(act[7]) == c - 1
    assert torch.allclose(logp, ref.log_prob(act), atol=1e-5, rtol=0)
    # uniform form (logits == NULL): legal, and the first legal cell when deterministic
    act = sampler.draw(None, mask, False)
    assert bool(legal.gather(1, act.unsqueeze(1)).all())
    first = sampler.draw(None, mask, True)
    want = torch.argmax(legal.to(torch.uint8), dim=1)
    assert torch.equal(first, want)


@pytest.mark.parametrize("c", [9, 81, 169])
def test_drawn_cells_always_carry_weight(hip, c):
    """Round 4 regression: the lane that holds the target was found with the scan's inclusive sum while the cell inside it
    was found with a sequentially accumulated one -- the same number in two associations.  When they differed in the last
    bit and the target fell between them the pick ran through to the lane's last slot, which stands for a cell PAST the
    end of the row (action C .. C+LPR-2: out of range) -- about once in 10^7 draws on 3x3, found by
    examples/selfplay_ppo_graphed.py.  Tens of millions of draws from peaked, half-masked rows: every action is a legal
    cell of its row, f32 and bf16 logits."""
    rows = 1 << 21 if c <= 81 else 1 << 20
    g = torch.Generator(device="cpu").manual_seed(c)
    logits = (torch.randn(rows, c, generator=g) * 6).to(DEV)
    mask = (torch.rand(rows, c, generator=g) > 0.5)
    mask[torch.arange(rows), torch.randint(0, c, (rows,), generator=g)] = True
    mask = mask.to(DEV)
    sampler = hip.policy.HipSampler(seed=c)
    for t in range(24):
        lg = logits if t % 3 else logits.to(torch.bfloat16)
        acts = sampler.draw(lg, mask, False)
        assert int(acts.max()) < c and int(acts.min()) >= 0, t
        assert bool(mask.gather(1, acts.unsqueeze(1)).all()), t
    acts = sampler.draw(None, mask, False)  # the uniform form
    assert bool(mask.gather(1, acts.unsqueeze(1)).all())


def test_checkpoint_resume_is_bit_exact(hip, tmp_path):
    """state_dict / load_state_dict of the env, the wrapper (with episode accounting) and the rollout driver:
    a run restored from a checkpoint written to disk continues exactly like the run that wrote it."""
    m, n, k, nenv = 9, 9, 5, 300
    # wrapper + fused random opponent
    w1 = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=4)
    w1.set_opponent(hip.policy.RandomPolicy(m * n, seed=8))
    w1.track_episodes()
    agent = hip.policy.RandomPolicy(m * n, seed=9)
    obs, _ = w1.reset()
    for _ in range(40):
        obs, *_ = w1.step(agent.act(obs))
    torch.save({"wrapper": w1.state_dict(), "agent_calls": agent._sampler.calls}, tmp_path / "ckpt.pt")
    tail1 = []
    for _ in range(30):
        obs, r, t, _, _ = w1.step(agent.act(obs))
        tail1.append((obs["observation"].clone(), r.clone(), t.clone()))
    ck = torch.load(tmp_path / "ckpt.pt", weights_only=False)  # our own file
    w2 = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=123)
    w2.set_opponent(hip.policy.RandomPolicy(m * n, seed=8))
    w2.load_state_dict(ck["wrapper"])
    agent2 = hip.policy.RandomPolicy(m * n, seed=9)
    agent2._sampler.calls = ck["agent_calls"]
    obs2 = w2.get_agent_obs()
    for (o, r, t) in tail1:
        obs2, r2, t2, _, _ = w2.step(agent2.act(obs2))
        assert torch.equal(obs2["observation"], o) and torch.equal(r2, r) and torch.equal(t2, t)
    assert w1.pop_episode_stats() == w2.pop_episode_stats()
    # rollout driver
    a = hip.rollout.RandomRollout(hip.Env(m, n, k, nenv, device=DEV), seed=5)
    a.run(64, record=False)
    state = a.state_dict()
    want = a.run(32)
    b = hip.rollout.RandomRollout(hip.Env(m, n, k, nenv, device=DEV), seed=0)
    b.load_state_dict(state)
    got = b.run(32)
    assert torch.equal(got.planes, want.planes) and torch.equal(got.meta, want.meta) and torch.equal(a.stats, b.stats)
    with pytest.raises(ValueError, match="state is for"):
        hip.Env(3, 3, 3, nenv, device=DEV).load_state_dict(state["env"])


def test_observations_beyond_4_GiB_are_addressed_with_64_bits(hip):
    """Sized for the device (288 GB): 2^23 envs of 9x9x5 -- an f32 observation of 5.4 GB, past any 32-bit byte offset.
    The last 4 096 envs of the big batch play, see and are rewarded exactly like a 4 096-env wrapper keyed with their
    global env ids (``env_id0``), and so do the first 4 096: the one-launch self-play step, then ``observe``."""
    m, n, k, nenv, part = 9, 9, 5, 1 << 23, 4096
    c = m * n

    def make(count, id0):
        w = hip.Wrapper(hip.Env(m, n, k, count, device=DEV), seed=31)
        w.env_id0 = id0
        w.set_opponent(hip.policy.RandomPolicy(c, seed=9))
        out = {"observation": torch.empty((count, 2, m, n), dtype=torch.float32, device=DEV),
               "action_mask": torch.empty((count, c), dtype=torch.bool, device=DEV),
               "rewards": torch.empty(count, dtype=torch.float32, device=DEV),
               "terminated": torch.empty(count, dtype=torch.bool, device=DEV)}
        return w, out

    big, big_out = make(nenv, 0)
    assert big_out["observation"].numel() * 4 > (1 << 32)
    parts = [(make(part, 0), slice(0, part)), (make(part, nenv - part), slice(nenv - part, nenv))]
    big.reset(out=big_out)
    for (w, out), rows in parts:
        w.reset(out=out)
        assert torch.equal(out["observation"], big_out["observation"][rows]) and torch.equal(out["action_mask"], big_out["action_mask"][rows])
    acts = torch.empty(nenv, dtype=torch.long, device=DEV)
    for t in range(8):
        big.env.sample_legal_into(acts, seed=77, step=t)    # a legal move per env, keyed by global env id
        big.step(acts, out=big_out)
        for (w, out), rows in parts:
            w.step(acts[rows].clone(), out=out)
            for key in out:
                assert torch.equal(out[key], big_out[key][rows]), (t, key, rows)
            assert torch.equal(w.agent_side, big.agent_side[rows]) and torch.equal(w.env._planes, big.env._planes[..., rows])
    got = big.get_agent_obs()  # (k_observe: another 5.4 GB)
    assert torch.equal(got["observation"], big_out["observation"]) and torch.equal(got["action_mask"], big_out["action_mask"])
    big.env.check_errors()

"""GPU parity for the round-3 additions to the C ABI (version 4): caller-owned step outputs and the rollout sink
(SURVEY.md section 8f rank 1), narrow observation dtypes, packed canonical planes written by the step kernels, and
BASELINE.json config 2 as one launch per ply (``mnk_step_random``).  Bit-exact everywhere."""
import numpy as np
import pytest
import torch

from oracle.env_torch import OracleVectorEnv
from oracle.packing import pack_boards
from oracle.policies import MaskHashPolicy
from oracle.rollout import random_rollout
from replay import golden_files, replay_ppo_learn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NARROW = [torch.bfloat16, torch.uint8]
# (m, n, k, envs): packed write-out with vector stores, the same board at a ragged batch (rows of a [T, N] buffer are
# then not 16-byte aligned: scalar tail path), a board on the generic table path, a larger packed board, 3x3
BOARDS = [(9, 9, 5, 1000), (9, 9, 5, 257), (4, 6, 3, 131), (13, 13, 5, 96), (3, 3, 3, 77), (7, 9, 7, 64)]


@pytest.fixture(scope="module")
def hip():
    import __graft_entry__ as entry

    entry.build_hip()
    entry._ensure_path()
    import mnk_hip
    from alg.packed_rollout_buffer import PackedRolloutBuffer
    from alg.rollout_buffer import RolloutBuffer
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay import policy, random_rollout as rr
    from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper

    mnk_hip.load()
    assert torch.cuda.is_available()

    class NS:
        pass

    ns = NS()
    ns.lib, ns.Env, ns.Wrapper, ns.Buffer, ns.PackedBuffer = mnk_hip, TorchVectorMnkEnv, TorchSelfPlayWrapper, RolloutBuffer, PackedRolloutBuffer
    ns.policy, ns.rollout = policy, rr
    return ns


def _midgame(hip, m, n, k, nenv, plies, seed=3, **kw):
    env = hip.Env(m, n, k, nenv, device=DEV, **kw)
    hip.rollout.RandomRollout(env, seed=seed).run(plies, record=False)
    return env


def _same(a, b):
    """a narrow observation equals the f32 one after .float(), and holds only 0 / 1"""
    return torch.equal(a.float(), b.float())


# ----------------------------------------------------------------------------- narrow observations
@pytest.mark.parametrize("m,n,k,nenv", BOARDS)
@pytest.mark.parametrize("dtype", NARROW)
def test_narrow_observations_equal_the_f32_ones(hip, m, n, k, nenv, dtype):
    """MNK_OBS_BF16 / MNK_OBS_U8 in every kernel that writes observations: observe (absolute and canonical), step,
    step with autoreset, the self-play kernels (one launch and two launches), unpack_records, gather_obs."""
    c = m * n
    plies = max(4, c // 3)
    env = _midgame(hip, m, n, k, nenv, plies)
    side = torch.from_numpy(np.random.default_rng(1).integers(0, 2, nenv)).to(DEV)

    def observe(dt, flip):
        obs = torch.empty((nenv, 2, m, n), dtype=dt, device=DEV)
        mask = torch.empty((nenv, c), dtype=torch.bool, device=DEV)
        env.observe_into(obs, mask, flip_side=flip, fix_empty_mask=flip is not None)
        return obs, mask

    for flip in (None, side):
        want, wmask = observe(torch.float32, flip)
        got, gmask = observe(dtype, flip)
        assert got.dtype == dtype and _same(got, want) and torch.equal(gmask, wmask)
        assert torch.equal(want, env.boards[...] if flip is None else
                           torch.where((flip == 1).view(-1, 1, 1, 1), env.boards[...].flip(1), env.boards[...]))

    # step / step + autoreset from the same position
    acts = hip.policy.RandomPolicy(c, seed=5).act({"action_mask": env.legal_mask()})
    for autoreset in (False, True):
        outs = []
        for dt in (torch.float32, dtype):
            e = hip.Env(m, n, k, nenv, device=DEV)
            e.load_state_dict(env.state_dict())
            obs = torch.empty((nenv, 2, m, n), dtype=dt, device=DEV)
            mask = torch.empty((nenv, c), dtype=torch.bool, device=DEV)
            rew = torch.empty(nenv, dtype=torch.float32, device=DEV)
            done = torch.empty(nenv, dtype=torch.bool, device=DEV)
            e.step_into(acts, rew, done, mask, obs, autoreset=autoreset)
            outs.append((obs, mask, rew, done, e._planes.clone()))
        for a, b in zip(outs[0][1:], outs[1][1:]):
            assert torch.equal(a, b)
        assert _same(outs[1][0], outs[0][0])

    # the env's own obs_dtype: every fresh observation comes out narrow, through the wrapper too (both step forms)
    for opp in ("fused", "two-launch"):
        runs = []
        for dt in (torch.float32, dtype):
            e = hip.Env(m, n, k, nenv, device=DEV, obs_dtype=dt)
            w = hip.Wrapper(e, seed=9)
            seen = []

            class Spy:  # a row-local deterministic opponent that also records the dtype it is shown
                def __init__(self):
                    self.inner = MaskHashPolicy(2)

                def act(self, obs, deterministic=False):
                    seen.append(obs["observation"].dtype)
                    return self.inner.act(obs)

            w.set_opponent(hip.policy.RandomPolicy(c, seed=4) if opp == "fused" else Spy())
            agent = hip.policy.RandomPolicy(c, seed=6)
            obs, _ = w.reset()
            trace = [obs["observation"].float()]
            for _ in range(plies):
                obs, rew, term, _, _ = w.step(agent.act(obs))
                assert obs["observation"].dtype == dt
                trace += [obs["observation"].float(), obs["action_mask"].float(), rew, term.float()]
            assert all(s == dt for s in seen)
            runs.append(trace)
        assert all(torch.equal(a, b) for a, b in zip(*runs))

    # records -> RolloutBuffer layout, and the minibatch gather from packed observations
    env2 = hip.Env(m, n, k, nenv, device=DEV)
    rec = hip.rollout.RandomRollout(env2, seed=8).run(12)
    want = hip.rollout.unpack_records(rec, env2)
    got = hip.rollout.unpack_records(rec, env2, obs_dtype=dtype)
    assert got["observations"].dtype == dtype and _same(got["observations"], want["observations"])
    for key in ("action_masks", "actions", "rewards", "dones"):
        assert torch.equal(got[key], want[key])
    buf = hip.PackedBuffer(3, nenv, m, n, device=DEV)
    w = hip.Wrapper(env, seed=1)
    w.agent_side.copy_(side)
    for t in range(3):
        buf.planes[t].copy_(w.packed_obs())
        hip.rollout.RandomRollout(env, seed=20 + t).run(2, record=False)
    pick = torch.randperm(3 * nenv, generator=torch.Generator().manual_seed(0))[:200].to(DEV)
    o32, m32 = buf.gather(pick)
    onw, mnw = buf.gather(pick, obs_dtype=dtype)
    assert onw.dtype == dtype and _same(onw, o32) and torch.equal(mnw, m32)


def test_unsupported_observation_dtype_is_refused(hip):
    env = hip.Env(3, 3, 3, 8, device=DEV)
    with pytest.raises(TypeError):
        env.observe_into(torch.empty((8, 2, 3, 3), dtype=torch.float16, device=DEV))
    with pytest.raises(TypeError):
        hip.Env(3, 3, 3, 8, device=DEV, obs_dtype=torch.int32)
    with pytest.raises(hip.lib.MnkHipError):  # the C ABI checks the code too
        hip.lib.call("mnk_observe", hip.lib.ptr(env._planes), hip.lib.ptr(env._meta), 8, 3, 3, None,
                     hip.lib.ptr(torch.empty((8, 2, 3, 3), device=DEV)), 7, None, 0, None, env._stream())


# ----------------------------------------------------------------------------- packed canonical planes
@pytest.mark.parametrize("m,n,k,nenv", BOARDS)
@pytest.mark.parametrize("fused", [True, False])
def test_step_kernels_emit_the_packed_canonical_planes(hip, m, n, k, nenv, fused):
    """``out["packed"]`` of wrapper.reset / wrapper.step == the canonical observation of the same step, packed
    (channel 0 = the agent's stones), == wrapper.packed_obs() taken afterwards."""
    c = m * n
    w = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=2)
    w.set_opponent(hip.policy.RandomPolicy(c, seed=1) if fused else MaskHashPolicy(1))
    agent = hip.policy.RandomPolicy(c, seed=7)
    packed = torch.full((2, w.env.words, nenv), -1, dtype=torch.int64, device=DEV)
    obs, _ = w.reset(out={"packed": packed})
    for t in range(max(6, c // 2)):
        want = pack_boards(obs["observation"].cpu().numpy(), m, n)
        assert np.array_equal(packed.cpu().numpy().view(np.uint64), want), t
        assert torch.equal(w.packed_obs(), packed), t
        obs, *_ = w.step(agent.act(obs), out={"packed": packed})


# ----------------------------------------------------------------------------- caller-owned outputs and the sink
def test_step_writes_into_the_tensors_it_is_given(hip):
    m, n, k, nenv, c = 9, 9, 5, 300, 81
    a, b = (hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=4) for _ in range(2))
    for w in (a, b):
        w.set_opponent(hip.policy.RandomPolicy(c, seed=8))
    out = {"observation": torch.empty((nenv, 2, m, n), device=DEV), "action_mask": torch.empty((nenv, c), dtype=torch.bool, device=DEV),
           "rewards": torch.empty(nenv, device=DEV), "terminated": torch.empty(nenv, dtype=torch.bool, device=DEV)}
    oa, _ = a.reset()
    ob, _ = b.reset(out=out)
    assert ob["observation"] is out["observation"] and ob["action_mask"] is out["action_mask"]
    agent = hip.policy.RandomPolicy(c, seed=1)
    for t in range(60):
        acts = agent.act(oa)
        oa, ra, ta, tra, _ = a.step(acts)
        ob, rb, tb, trb, _ = b.step(acts, out=out)
        assert rb is out["rewards"] and tb is out["terminated"] and ob["observation"] is out["observation"]
        assert torch.equal(oa["observation"], ob["observation"]) and torch.equal(oa["action_mask"], ob["action_mask"])
        assert torch.equal(ra, rb) and torch.equal(ta, tb) and not bool(trb.any())
    for bad in ({"observation": torch.empty((nenv, 2, n, m + 1), device=DEV)},
                {"rewards": torch.empty(nenv + 1, device=DEV)},
                {"action_mask": torch.empty((c, nenv), dtype=torch.bool, device=DEV).t()},
                {"terminated": torch.empty(nenv, dtype=torch.bool)}):
        with pytest.raises(ValueError):
            b.step(acts, out=bad)
    with pytest.raises(TypeError):
        b.step(acts, out={"terminated": torch.empty(nenv, dtype=torch.uint8, device=DEV)})


def _set_sides(wrapper, sides):
    wrapper.force_sides(torch.from_numpy(sides.astype(np.int64)))


@pytest.mark.parametrize("idx", range(2))
def test_ppo_learn_rollout_through_the_sink_equals_the_reference(hip, golden_dir, idx):
    """The reference's two consecutive PPOAgent.learn rollouts (tests/golden/ppo_learn_*.npz) replayed with ONE buffer
    kept across the calls and the wrapper attached to it (``attach_sink``): the call sequence is the reference's
    (alg/ppo.py:81-146, buffer.reset() at the end of learn), every buffer field still equals the reference buffer's bit
    for bit, and per learn call ``add`` copied only the small per-env vectors plus the one spill row."""
    log = np.load(golden_files(golden_dir, "ppo_learn_")[idx])
    m, n, k, nenv, n_steps = (int(v) for v in log["geom"])
    wrap = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV))
    wrap.set_opponent(MaskHashPolicy(0 if m == 3 else 1))
    wrap.track_episodes()
    buf = hip.Buffer(n_steps, nenv, (2, m, n), m * n, device=DEV)
    wrap.attach_sink(buf)
    ptrs = {name: getattr(buf, name).data_ptr() for name in buf._FIELDS}
    copied = []

    def make_buffer(*_):
        if copied or buf.ptr:  # the end of the previous learn(): ppo.py:146
            copied.append(buf.copied_bytes)
            buf.reset()
        else:
            copied.append(0)
        return buf

    replay_ppo_learn(wrap, make_buffer, log, _set_sides, episode_stats=wrap.pop_episode_stats)
    copied.append(buf.copied_bytes)
    assert {name: getattr(buf, name).data_ptr() for name in buf._FIELDS} == ptrs  # reset() kept the storage
    row = nenv * (2 * m * n * 4 + m * n)                       # one observation + mask row
    small = n_steps * nenv * (8 + 4 + 4 + 1)                   # actions, values, log_probs, dones per learn call
    first, second = copied[1] - copied[0], copied[2] - copied[1]
    assert first == small, (first, small)                      # reset() wrote row 0 in place
    assert second == small + row, (second, small + row)        # the carried-over observation: one row per learn call


@pytest.mark.parametrize("obs_dtype", NARROW)
def test_ppo_learn_rollout_through_a_narrow_sink_equals_the_reference(hip, golden_dir, obs_dtype):
    """The same replay of the reference's learn rollouts with bf16 / u8 observations end to end: the env writes them
    narrow, straight into a RolloutBuffer allocated in that dtype; every cell of every stored observation (and every
    other field) still equals the reference buffer's."""
    log = np.load(golden_files(golden_dir, "ppo_learn_")[1])
    m, n, k, nenv, n_steps = (int(v) for v in log["geom"])
    wrap = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV, obs_dtype=obs_dtype))
    wrap.set_opponent(MaskHashPolicy(0 if m == 3 else 1))
    wrap.track_episodes()
    buf = hip.Buffer(n_steps, nenv, (2, m, n), m * n, device=DEV, obs_dtype=obs_dtype)
    assert buf.observations.dtype == obs_dtype
    wrap.attach_sink(buf)

    def make_buffer(*_):
        if buf.ptr:
            buf.reset()
        return buf

    replay_ppo_learn(wrap, make_buffer, log, _set_sides, episode_stats=wrap.pop_episode_stats)
    es = torch.empty((), dtype=obs_dtype).element_size()
    assert buf.copied_bytes == 2 * n_steps * nenv * 17 + nenv * (2 * m * n * es + m * n)  # small vectors + one spill row


def test_sink_rows_are_what_the_step_returns(hip):
    """With a sink attached the tensors a step returns ARE rows of the buffer; past the last row the step goes back
    to fresh tensors; a packed buffer takes the packed planes."""
    m, n, k, nenv, c, steps = 9, 9, 5, 128, 81, 5
    w = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=3)
    w.set_opponent(hip.policy.RandomPolicy(c, seed=2))
    buf = hip.Buffer(steps, nenv, (2, m, n), c, device=DEV)
    w.attach_sink(buf)
    obs, _ = w.reset()
    assert obs["observation"].data_ptr() == buf.observations[0].data_ptr()
    agent = hip.policy.RandomPolicy(c, seed=5)
    zeros = torch.zeros(nenv, device=DEV)
    for t in range(steps):
        acts = agent.act(obs)
        nxt, rew, term, trunc, _ = w.step(acts)
        assert rew.data_ptr() == buf.rewards[t].data_ptr() and term.data_ptr() == buf.dones[t].data_ptr()
        assert nxt["observation"].data_ptr() == buf.row(t + 1)["observation"].data_ptr()
        assert nxt["action_mask"].data_ptr() == buf.row(t + 1)["action_mask"].data_ptr()
        buf.add(obs["observation"], acts, rew, zeros.view(-1, 1), zeros, term | trunc, obs["action_mask"])
        obs = nxt
    assert buf.copied_bytes == steps * nenv * (8 + 4 + 4 + 1)
    nxt, rew, *_ = w.step(agent.act(obs))          # buffer full: ordinary fresh tensors again
    assert rew.data_ptr() not in {buf.rewards[t].data_ptr() for t in range(steps)}
    with pytest.raises(IndexError, match="Buffer was full."):
        buf.add(obs["observation"], acts, rew, zeros.view(-1, 1), zeros, term, obs["action_mask"])

    # the packed buffer: the step kernel writes the canonical planes of the next observation into row t+1
    pbuf = hip.PackedBuffer(steps, nenv, m, n, device=DEV)
    dense = hip.Buffer(steps, nenv, (2, m, n), c, device=DEV)
    w2 = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=3)
    w2.set_opponent(hip.policy.RandomPolicy(c, seed=2))
    w2.attach_sink(pbuf)
    obs, _ = w2.reset()
    agent = hip.policy.RandomPolicy(c, seed=5)
    for t in range(steps):
        acts = agent.act(obs)
        nxt, rew, term, trunc, _ = w2.step(acts)
        pbuf.add(pbuf.row(t)["packed"], acts, rew, zeros, zeros, term | trunc)
        dense.add(obs["observation"], acts, rew, zeros.view(-1, 1), zeros, term | trunc, obs["action_mask"])
        obs = nxt
    assert pbuf.copied_bytes == steps * nenv * (8 + 4 + 4 + 1)
    every = torch.arange(steps * nenv, device=DEV)
    o, msk = pbuf.gather(every)
    assert torch.equal(o, dense.observations.reshape(-1, 2, m, n)) and torch.equal(msk, dense.action_masks.reshape(-1, c))
    assert torch.equal(pbuf.rewards, buf.rewards) and torch.equal(pbuf.dones, buf.dones)  # same seeds as the dense run


# ----------------------------------------------------------------------------- BASELINE config 2 in one launch per ply
@pytest.mark.parametrize("m,n,k,nenv", [(9, 9, 5, 1000), (3, 3, 3, 257), (4, 6, 3, 96), (13, 13, 5, 64)])
def test_step_random_is_sample_step_reset_observe_in_one_launch(hip, m, n, k, nenv):
    """mnk_step_random == mnk_sample_legal -> mnk_step(AUTORESET) on the HIP side, == the oracle's raw loop
    (RandomPolicy.act -> env.step -> env.reset(nonzero(done)) -> observe) with the Philox draw, ply after ply,
    == the records of the fused rollout kernel."""
    c, seed, plies = m * n, 31, 2 * m * n + 3
    one = hip.Env(m, n, k, nenv, device=DEV)
    two = hip.Env(m, n, k, nenv, device=DEV)
    ora = OracleVectorEnv(m, n, k, nenv)
    planes, meta, _ = random_rollout(ora, seed=seed, step0=0, steps=plies)
    ora2 = OracleVectorEnv(m, n, k, nenv)

    def bufs():
        return (torch.empty(nenv, dtype=torch.float32, device=DEV), torch.empty(nenv, dtype=torch.bool, device=DEV),
                torch.empty((nenv, c), dtype=torch.bool, device=DEV), torch.empty((nenv, 2, m, n), device=DEV),
                torch.empty(nenv, dtype=torch.long, device=DEV))

    r1, d1, m1, o1, a1 = bufs()
    r2, d2, m2, o2, a2 = bufs()
    for t in range(plies):
        one.step_random_into(r1, d1, m1, o1, a1, seed=seed, step=t)
        two.sample_legal_into(a2, seed=seed, step=t)
        two.step_into(a2, r2, d2, m2, o2, autoreset=True)
        for x, y in ((r1, r2), (d1, d2), (m1, m2), (o1, o2), (a1, a2), (one._planes, two._planes), (one._meta, two._meta)):
            assert torch.equal(x, y), t
        mw = meta[t]
        assert np.array_equal(a1.cpu().numpy(), (mw & 0xFFFF).astype(np.int64)), t
        assert np.array_equal(r1.cpu().numpy(), ((mw >> 16) & 0xFF).astype(np.int8).astype(np.float32)), t
        assert np.array_equal(d1.cpu().numpy(), ((mw >> 24) & 1).astype(bool)), t
        # the position after the ply (and after the reset of finished games) is what the oracle observes next
        _, _, dn = ora2.step(torch.from_numpy((mw & 0xFFFF).astype(np.int64)))
        if bool(dn.any()):
            ora2.reset(torch.nonzero(dn).squeeze(1))
        want = ora2.observe()
        assert torch.equal(o1.cpu(), want["observation"]) and torch.equal(m1.cpu(), want["action_mask"]), t
    # without autoreset a finished game stays finished (the raw step), and outputs may be skipped
    e = hip.Env(m, n, k, nenv, device=DEV)
    for t in range(c + 2):
        e.step_random_into(r1, d1, seed=seed, step=t, autoreset=False)
    assert int(e.move_counts.min()) == c + 2


# ----------------------------------------------------------------------------- the whole rollout as one hipGraph
@pytest.mark.parametrize("agent,opponent,packed", [("random", "random", False), ("net", "nn", False), ("random", "random", True),
                                                   ("net", "random", True)])
def test_graphed_rollout_fills_the_buffer_like_the_eager_loop(hip, agent, opponent, packed):
    """selfplay/graphed.py GraphedRollout: n_steps agent-steps as one captured graph writing into the buffer's rows ==
    the eager PPO loop (net -> fused draw -> wrapper.step -> buffer.add) with the same seeds, rollout after rollout
    (the observation carried over between rollouts included)."""
    graphed_rollout_against_the_eager_loop(hip, agent, opponent, packed, (3, 3, 3))


def graphed_rollout_against_the_eager_loop(hip, agent, opponent, packed, board):
    """(tests/test_gpu_jit_api.py runs the same comparison on boards whose kernels are compiled at run time)"""
    import copy

    import torch.nn as nn

    from selfplay.graphed import GraphedRollout

    m, n, k = board
    nenv, c, steps, rollouts = 200, m * n, 7, 3  # an odd number of steps: the scratch slots swap roles

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.body = nn.Sequential(nn.Flatten(), nn.Linear(2 * c, 32), nn.Tanh())
            self.pi, self.v = nn.Linear(32, c), nn.Linear(32, 1)

        def forward(self, obs, action_mask=None):
            h = self.body(obs)
            return torch.distributions.Categorical(logits=self.pi(h), validate_args=False), torch.tanh(self.v(h))

    torch.manual_seed(1)
    net = Net().to(DEV).eval() if agent == "net" else None
    opp_net = Net().to(DEV).eval()

    def make():
        w = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=5)
        w.set_opponent(hip.policy.RandomPolicy(c, seed=3) if opponent == "random"
                       else hip.policy.FusedNNPolicy(copy.deepcopy(opp_net), seed=3))
        buf = hip.PackedBuffer(steps, nenv, m, n, device=DEV) if packed else hip.Buffer(steps, nenv, (2, m, n), c, device=DEV)
        return w, buf

    w, buf = make()
    roll = GraphedRollout(w, buf, net, seed=11)   # plays one rollout while warming up
    # the eager twin
    w2, buf2 = make()
    w2.attach_sink(buf2)
    sampler = hip.policy._HipSampler(seed=11)
    obs, _ = w2.reset()
    dense_obs = []
    for r in range(rollouts):
        if r:
            roll.run()
            buf2.reset()
        for t in range(steps):
            if net is not None:
                with torch.no_grad():
                    dist, values = net(obs["observation"], None)
                logits = dist.logits
            else:
                logits, values = None, torch.zeros(nenv, 1, device=DEV)
            actions, logp = sampler.draw(logits, obs["action_mask"], False, want_logp=True)
            # the packed planes of the observation acted on: where reset() / the previous step put them (the spill row
            # after the last step of a rollout, copied into row 0 by add)
            cur_packed = buf2.row(steps if (r and t == 0) else t).get("packed")
            nxt, rew, term, trunc, _ = w2.step(actions)
            if packed:
                buf2.add(cur_packed, actions, rew, values, logp, term | trunc)
            else:
                buf2.add(obs["observation"], actions, rew, values, logp, term | trunc, obs["action_mask"])
            obs = nxt
        where = f"rollout {r}"
        assert buf.ptr == steps
        for name in ("actions", "rewards", "dones", "log_probs") + (("values",) if net is not None else ()):
            assert torch.equal(getattr(buf, name), getattr(buf2, name)), (where, name)
        if packed:
            assert torch.equal(buf.planes, buf2.planes), where
        else:
            assert torch.equal(buf.observations, buf2.observations) and torch.equal(buf.action_masks, buf2.action_masks), where
        nxt_graph = roll.next_obs()
        assert torch.equal(nxt_graph["observation"], obs["observation"]) and torch.equal(nxt_graph["action_mask"], obs["action_mask"]), where


# ----------------------------------------------------------------------------- round 4: sink hardening
def _copying_loop(hip, wrap, buf, agent, obs, rollouts, fields):
    """the reference-shaped loop (alg/ppo.py:93-108, :148): act, step, add, ... then buffer.reset()"""
    zeros = torch.zeros(wrap.num_envs, device=DEV)
    snaps = []
    for _ in range(rollouts):
        for _ in range(buf.n_steps):
            actions = agent.act(obs)
            nxt, rew, term, trunc, _ = wrap.step(actions)
            buf.add(obs["observation"], actions, rew, zeros, zeros, term | trunc, obs["action_mask"])
            obs = nxt
        snaps.append({f: getattr(buf, f)[:buf.n_steps].clone() for f in fields})
        buf.reset()
    return snaps, obs


@pytest.mark.parametrize("n_steps", [1, 2, 3])
@pytest.mark.parametrize("opponent", ["random", "scripted"])
def test_consecutive_short_rollouts_through_the_sink_equal_the_copying_loop(hip, n_steps, opponent):
    """with ``n_steps == 1`` the observation acted on at t = 0 of every rollout after the first IS the
    spill row, which the same step would overwrite before ``add`` reads it.  A one-step buffer therefore keeps two spill
    rows and uses them in turn: rollouts of 1, 2 and 3 steps through the sink equal the reference-shaped copying loop
    (no sink: fresh tensors per step, ``add`` copies everything), every field, every rollout."""
    m, n, k, nenv, rollouts = 9, 9, 5, 300, 5
    c = m * n
    fields = ("observations", "action_masks", "actions", "rewards", "dones")
    got = {}
    for sink in (True, False):
        wrap = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=11)
        wrap.set_opponent(hip.policy.RandomPolicy(c, seed=12) if opponent == "random" else MaskHashPolicy())
        buf = hip.Buffer(n_steps, nenv, (2, m, n), c, device=DEV)
        if sink:
            wrap.attach_sink(buf)
        obs, _ = wrap.reset()
        got[sink], last = _copying_loop(hip, wrap, buf, hip.policy.RandomPolicy(c, seed=13), obs, rollouts, fields)
        got[sink].append({"observations": last["observation"].clone(), "action_masks": last["action_mask"].clone()})
        if sink:
            assert buf.copied_bytes < rollouts * n_steps * nenv * 17 + rollouts * nenv * 729 + 1  # per rollout at most one observation copy
    for j, (a, b) in enumerate(zip(got[True], got[False])):
        for f in a:
            assert torch.equal(a[f], b[f]), (j, f)


@pytest.mark.parametrize("obs_dtype", [torch.float32] + NARROW)
@pytest.mark.parametrize("packed", [False, True])
def test_buffer_reset_keeps_the_observation_the_caller_acts_on(hip, obs_dtype, packed):
    """``buffer.reset()`` between ``learn`` calls (ppo.py:148) zeroes the storage in place while a sink is attached -- but
    never the row that holds the observation handed out last: the spill row after a full rollout, row 0 right after
    ``wrapper.reset()`` (a caller that resets the buffer at the START of its rollout), any row in between.  f32, bf16 and
    u8 observations; dense and packed buffers."""
    m, n, k, nenv, steps = 9, 9, 5, 200, 4
    c = m * n
    if packed and obs_dtype != torch.float32:
        pytest.skip("the packed buffer stores planes, not observations")
    wrap = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV, obs_dtype=obs_dtype), seed=5)
    wrap.set_opponent(hip.policy.RandomPolicy(c, seed=6))
    buf = hip.PackedBuffer(steps, nenv, m, n, device=DEV) if packed else hip.Buffer(steps, nenv, (2, m, n), c, device=DEV, obs_dtype=obs_dtype)
    store = buf._plane_store if packed else buf._obs_store
    wrap.attach_sink(buf)
    assert buf.keep_storage
    agent = hip.policy.RandomPolicy(c, seed=7)
    zeros = torch.zeros(nenv, device=DEV)

    def current():
        return wrap.packed_obs() if packed else wrap.get_agent_obs()["observation"]

    obs, _ = wrap.reset()
    base = store.data_ptr()
    for stop in (0, 2, steps):          # reset right after wrapper.reset(), mid-rollout, after a full rollout
        for _ in range(stop):
            actions = agent.act(obs)
            key = store[buf._live].clone()
            nxt, rew, term, trunc, _ = wrap.step(actions)
            if packed:
                buf.add(key, actions, rew, zeros, zeros, term | trunc)
            else:
                buf.add(obs["observation"], actions, rew, zeros, zeros, term | trunc, obs["action_mask"])
            obs = nxt
        live = buf._live
        buf.reset()
        assert store.data_ptr() == base and buf.ptr == 0            # same storage
        assert torch.equal(store[live], current())                 # the observation to act on is still there
        others = [r for r in range(steps) if r != live]
        assert not bool(store[others].any())                       # everything else was zeroed
        if not packed:
            assert torch.equal(obs["observation"], current()) and bool(obs["action_mask"].any(dim=1).all())
    # without a sink the buffer allocates fresh tensors like the reference: views kept by the caller keep their data
    plain = hip.Buffer(steps, nenv, (2, m, n), c, device=DEV)
    plain.add(torch.ones(nenv, 2, m, n, device=DEV), zeros.long(), zeros, zeros, zeros, zeros.bool(), torch.ones(nenv, c, dtype=torch.bool, device=DEV))
    kept = plain.observations
    plain.reset()
    assert bool((kept[0] == 1).all()) and not bool(plain.observations.any()) and plain.observations.data_ptr() != kept.data_ptr()

"""GPU parity of the step kernels with the masked draw folded in (round 4; ABI v5: ``mnk_selfplay_pre_logits``,
``mnk_selfplay_post_logits``, ``mnk_selfplay_step_random_logits``) against what they replace --
``mnk_sample_logits`` followed by the plain step kernels, itself pinned to the reference's masked ``Categorical``
(``masked_logits.npz``) and to the oracle -- on the same Philox streams: bit for bit."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle.policies import LowestLegalPolicy

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    import __graft_entry__ as entry

    entry.build_hip()
    entry._ensure_path()
    import mnk_hip
    from alg.rollout_buffer import RolloutBuffer
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay import graphed, policy
    from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper

    mnk_hip.load()

    class NS:
        pass

    ns = NS()
    ns.lib, ns.Env, ns.Wrapper, ns.policy, ns.graphed, ns.Buffer = mnk_hip, TorchVectorMnkEnv, TorchSelfPlayWrapper, policy, graphed, RolloutBuffer
    return ns


class TinyNet(nn.Module):
    """``net(obs, mask) -> (dist, value)`` with ``dist.logits`` the raw head output, like every reference architecture
    called with ``action_mask=None`` (cnn.py:69-80).  Row-dependent logits so every env plays its own game."""

    def __init__(self, m, n, seed, dtype=torch.float32):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.conv = nn.Conv2d(2, 4, 3, padding=1)
        self.bn = nn.BatchNorm2d(4)
        self.head = nn.Linear(4 * m * n, m * n)
        self.value = nn.Linear(4 * m * n, 1)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.7)
            self.bn.running_mean.copy_(torch.randn(4, generator=g) * 0.1)
            self.bn.running_var.copy_(torch.rand(4, generator=g) + 0.5)
        self.out_dtype = dtype

    def forward(self, obs, action_mask=None):
        f = torch.relu(self.bn(self.conv(obs.float()))).flatten(1)
        logits = self.head(f).to(self.out_dtype)
        if action_mask is not None:
            logits = torch.where(action_mask, logits, torch.full_like(logits, -torch.inf))

        class Dist:
            pass

        d = Dist()
        d.logits = logits
        return d, self.value(f)


def _state(w):
    env = w.env
    return (env._planes.clone(), env._meta.clone(), w.agent_side.clone(), w.pending_resets.clone())


def _same_state(a, b):
    return all(torch.equal(x, y) for x, y in zip(_state(a), _state(b)))


BOARDS = [(3, 3, 3, 300), (9, 9, 5, 257), (9, 9, 5, 65536), (13, 13, 5, 130), (15, 15, 5, 70), (19, 19, 5, 300),
          (7, 9, 4, 100), (9, 9, 4, 64)]  # the last two have no built-in draw shape: two launches inside the call (the fold
# comes from hiprtc once the kernel is hot: tests/test_gpu_jit_api.py)


@pytest.mark.parametrize("opponent", ["random", "scripted", "net"])
@pytest.mark.parametrize("m,n,k,nenv", BOARDS)
def test_step_logits_equals_sample_then_step(hip, m, n, k, nenv, opponent):
    """``wrapper.step_logits`` (the agent's draw inside ``mnk_selfplay_pre_logits`` / ``_step_random_logits``, a
    ``FusedNNPolicy`` opponent's inside ``mnk_selfplay_post_logits``) == ``sampler.draw`` + ``wrapper.step`` with the
    opponent drawing through ``mnk_sample_logits``: actions, log-probabilities, every output of every step, the whole
    state -- f32, bf16 and absent logits, stochastic and deterministic, a ragged last workgroup."""
    c = m * n
    steps = 6 if nenv > 10000 else 14

    def build(fused):
        w = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=21)
        if opponent == "random":
            w.set_opponent(hip.policy.RandomPolicy(c, seed=5))
        elif opponent == "scripted":
            w.set_opponent(LowestLegalPolicy())
        else:
            w.set_opponent(hip.policy.FusedNNPolicy(TinyNet(m, n, 3).to(DEV), seed=6))
        w.fuse_opponent_draw = fused
        return w, hip.policy.HipSampler(seed=77)

    (wa, sa), (wb, sb) = build(False), build(True)
    oa, _ = wa.reset()
    ob, _ = wb.reset()
    assert torch.equal(oa["observation"], ob["observation"]) and torch.equal(oa["action_mask"], ob["action_mask"])
    g = torch.Generator(device="cpu").manual_seed(c + nenv)
    for t in range(steps):
        kind = t % 4
        logits = None if kind == 3 else (torch.randn(nenv, c, generator=g) * 2.5).to(DEV)
        if kind == 1:
            logits = logits.to(torch.bfloat16)
        det = t == steps - 1
        acts, logp = sa.draw(logits, oa["action_mask"], det, want_logp=True)
        oa, ra, ta, _, _ = wa.step(acts)
        ob, rb, tb, _, info = wb.step_logits(logits, ob["action_mask"], sb, deterministic=det)
        assert torch.equal(info["actions"], acts), (t, kind)
        assert torch.equal(info["log_probs"], logp), (t, kind)
        assert torch.equal(oa["observation"], ob["observation"]) and torch.equal(oa["action_mask"], ob["action_mask"]), t
        assert torch.equal(ra, rb) and torch.equal(ta, tb) and _same_state(wa, wb), t
        if opponent != "random":
            assert torch.equal(wa.last_opponent_actions, wb.last_opponent_actions), t
    assert sa.calls == sb.calls == steps
    wa.env.check_errors()
    wb.env.check_errors()


def test_fused_opponent_draw_is_the_default_and_counts_two_launches(hip, monkeypatch):
    """A network agent against a network opponent: the env side of an agent-step is TWO library calls (was four:
    sample, pre, sample, post)."""
    m, n, k, nenv = 9, 9, 5, 512
    w = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=1)
    w.set_opponent(hip.policy.FusedNNPolicy(TinyNet(m, n, 1).to(DEV), seed=2))
    agent = hip.policy.FusedNNPolicy(TinyNet(m, n, 2).to(DEV), seed=3)
    obs, _ = w.reset()
    calls = []
    real = hip.lib.call
    monkeypatch.setattr(hip.lib, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    obs, *_ = w.step_logits(agent.logits(obs), obs["action_mask"], agent._sampler)
    assert calls == ["mnk_selfplay_pre_logits", "mnk_selfplay_post_logits"], calls
    del calls[:]
    w.set_opponent(hip.policy.RandomPolicy(m * n, seed=4))
    w.step_logits(agent.logits(obs), obs["action_mask"], agent._sampler)
    assert calls == ["mnk_selfplay_step_random_logits"], calls


@pytest.mark.parametrize("arch", ["cnn_b_s", "resnet_b_s"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_folded_draw_against_the_reference_head(hip, golden_dir, arch, dtype):
    """``masked_logits.npz`` (the reference nets' raw logits, masks and the log-probabilities its masked ``Categorical``
    holds, cnn.py:69-80) through the FOLDED draw: argmax and log-probability within 1e-5, sampled actions legal with the
    reference's log-probability.  The mask is an input of the kernel, so the fixture's masks are used as they are."""
    g = np.load(f"{golden_dir}/masked_logits.npz")
    raw = torch.from_numpy(g[arch + "_raw_logits"]).to(DEV).to(dtype)
    mask = torch.from_numpy(g[arch + "_mask"]).to(DEV)
    rows, c = mask.shape
    assert c == 81
    # the reference's own dist.logits (normalised, -inf on illegal cells) for the f32 head
    want = torch.from_numpy(g[arch + "_masked_logits"]).to(DEV) if dtype == torch.float32 else None
    ref_logits = torch.where(mask, raw.float(), torch.full_like(raw.float(), -torch.inf))
    dead = ~mask.any(dim=1)
    ref_logits[dead] = 0.0  # cnn.py:76-77
    ref = torch.distributions.Categorical(logits=ref_logits)
    for opp in ("random", "scripted"):
        w = hip.Wrapper(hip.Env(9, 9, 5, rows, device=DEV), seed=3)
        w.set_opponent(hip.policy.RandomPolicy(81, seed=1) if opp == "random" else LowestLegalPolicy())
        w.reset()
        sampler = hip.policy.HipSampler(seed=12)
        *_, info = w.step_logits(raw, mask, sampler, deterministic=True)
        assert torch.equal(info["actions"], torch.argmax(ref.logits, dim=1))
        assert torch.allclose(info["log_probs"], ref.log_prob(info["actions"]), atol=1e-5, rtol=0)
        if want is not None:
            assert torch.allclose(info["log_probs"], want.gather(1, info["actions"].unsqueeze(1)).squeeze(1), atol=1e-5, rtol=0)
        w.reset()
        *_, info = w.step_logits(raw, mask, sampler)
        act = info["actions"]
        assert bool(mask[~dead].gather(1, act[~dead].unsqueeze(1)).all())
        assert torch.allclose(info["log_probs"], ref.log_prob(act), atol=1e-5, rtol=0)


@pytest.mark.parametrize("arch,dtype,kernel", [("cnn_b_s", torch.float32, "step_random"), ("resnet_b_s", torch.float32, "pre"),
                                              ("cnn_b_s", torch.bfloat16, "post")])
def test_folded_draw_distribution_chi_square(hip, golden_dir, arch, dtype, kernel):
    """The chi-square test of ``test_sampler_distribution_chi_square`` THROUGH the folded draw: 8 rows of the reference
    nets' masked logits x 250 000 draws each (the batch is 250 000 envs that all hold the row; Philox is keyed by the
    env id), against the probabilities the reference's masked Categorical holds; 99.9 % quantile per row, Bonferroni-
    corrected.  One run per kernel: the agent's draw in ``_step_random_logits`` and ``_pre_logits``, the opponent's in
    ``_post_logits``."""
    from scipy.stats import chi2

    g = np.load(f"{golden_dir}/masked_logits.npz")
    masks = g[arch + "_mask"]
    rows = [r for r in range(masks.shape[0]) if masks[r].sum() >= 2][3::max(1, masks.shape[0] // 9)][:8]
    draws = 250000
    w = hip.Wrapper(hip.Env(9, 9, 5, draws, device=DEV), seed=8)
    sampler = hip.policy.HipSampler(seed=9)

    class FixedLogits:  # an opponent "network" whose head returns the row under test
        fused_logits = True

        def __init__(self):
            self._sampler, self.row = sampler, None

        def logits(self, obs):
            return self.row

    opp = FixedLogits()
    w.set_opponent(hip.policy.RandomPolicy(81, seed=1) if kernel == "step_random" else LowestLegalPolicy() if kernel == "pre" else opp)
    for row in rows:
        raw = torch.from_numpy(g[arch + "_raw_logits"][row]).to(DEV).to(dtype).expand(draws, -1).contiguous()
        mask = torch.from_numpy(masks[row]).to(DEV).expand(draws, -1).contiguous()
        if dtype == torch.float32:
            probs = g[arch + "_probs"][row].astype(np.float64)
        else:
            lg = raw[0].float().cpu().numpy().astype(np.float64)
            wgt = np.where(masks[row], np.exp(lg - lg[masks[row]].max()), 0.0)
            probs = wgt / wgt.sum()
        opp.row = raw
        if kernel != "post":
            w.reset(options={"agent_side": 1})
            *_, info = w.step_logits(raw, mask, sampler)
            acts = info["actions"].cpu().numpy()
        else:
            # agent white: the opponent (black) opens, every env needs its reply.  The mask the post kernel reads is the
            # one `pre` wrote for the opponent (all cells legal on the empty board): hand it the fixture's instead -- the
            # draw takes its mask from the pointer it is given
            real = hip.lib.call

            def spy(name, *a, _real=real, _mask=mask):
                if name == "mnk_selfplay_post_logits":
                    a = a[:8] + (hip.lib.ptr(_mask),) + a[9:]
                return _real(name, *a)

            hip.lib.call = spy
            try:
                w.reset(options={"agent_side": 1})
            finally:
                hip.lib.call = real
            acts = w.last_opponent_actions.cpu().numpy()
        counts = np.bincount(acts, minlength=len(probs)).astype(np.float64)
        assert counts[probs == 0].sum() == 0
        keep = probs * draws >= 5
        rest_p, rest_c = probs[~keep].sum(), counts[~keep].sum()
        stat = (((counts - draws * probs) ** 2)[keep] / (draws * probs[keep])).sum()
        dof = int(keep.sum()) - 1
        if rest_p * draws >= 5:
            stat += (rest_c - draws * rest_p) ** 2 / (draws * rest_p)
            dof += 1
        assert stat < chi2.ppf(1 - 0.001 / len(rows), dof), (arch, kernel, row, stat, dof)


def test_seed_dev_replaces_the_key_and_step_dev_adds_to_the_step(hip):
    """ABI v5: the sampler's Philox key may live in device memory (``seed_dev``): a draw with (seed_dev = K, step_dev = S)
    equals a draw with the host-side key K at step S -- in ``mnk_sample_logits`` and in the folded forms."""
    nenv, c = 1000, 81
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(nenv, c, generator=g).to(DEV)
    mask = (torch.rand(nenv, c, generator=g) > 0.3).to(DEV)
    mask[:, 0] = True
    plain = hip.policy.HipSampler(seed=0xDEADBEEFCAFEF00D)  # a key above 2^63: the int64 tensor holds its bit pattern
    plain.calls = 41
    want = plain.draw(logits, mask, False)
    dev = hip.policy.HipSampler(seed=1)
    dev.seed_dev = torch.tensor([0xDEADBEEFCAFEF00D - (1 << 64)], dtype=torch.int64, device=DEV)
    dev.step_dev = torch.tensor([40], dtype=torch.int64, device=DEV)
    dev.calls = 1
    assert torch.equal(dev.draw(logits, mask, False), want)
    for opp in ("random", "scripted"):
        ws = []
        for sampler in (plain, dev):
            w = hip.Wrapper(hip.Env(9, 9, 5, nenv, device=DEV), seed=2)
            w.set_opponent(hip.policy.RandomPolicy(c, seed=3) if opp == "random" else LowestLegalPolicy())
            obs, _ = w.reset()
            plain.calls = 41
            *_, info = w.step_logits(logits, obs["action_mask"], sampler)
            ws.append((w, info))
        assert torch.equal(ws[0][1]["actions"], ws[1][1]["actions"]) and _same_state(ws[0][0], ws[1][0])


def _eager_rollout(hip, w, buf, net, agent_sampler, obs):
    """the loop of alg/ppo.py:93-108 with FusedNNPolicy-style sampling, the sink attached"""
    buf.reset()
    for _ in range(buf.n_steps):
        with torch.no_grad():
            dist, values = net(obs["observation"], None)
        actions, logp = agent_sampler.draw(dist.logits, obs["action_mask"], False, want_logp=True)
        nxt, rew, term, trunc, _ = w.step(actions)
        buf.add(obs["observation"], actions, rew, values.reshape(-1), logp, term | trunc, obs["action_mask"])
        obs = nxt
    return obs


FIELDS = ("observations", "action_masks", "actions", "rewards", "dones", "values", "log_probs")


@pytest.mark.parametrize("how", ["set_opponent_weights", "wrapper.set_opponent"])
@pytest.mark.parametrize("nenv", [96, 700])
def test_opponent_swap_without_recapture_equals_the_eager_loop(hip, nenv, how):
    """The reference installs a fresh deepcopy of the agent as opponent before EVERY rollout (train.py:106-114).
    ``GraphedRollout.set_opponent_weights`` does that to the captured opponent in place (weights, BatchNorm statistics,
    a new Philox key through the device word) with no new capture; the rollouts equal the eager loop with
    ``wrapper.set_opponent(FusedNNPolicy(deepcopy(net), seed=s))`` before each, every buffer field, bit for bit."""
    m, n, k, steps, rollouts = 9, 9, 5, 6, 5
    c = m * n
    net = TinyNet(m, n, 10).to(DEV).eval()
    sources = [TinyNet(m, n, 20 + j).to(DEV).eval() for j in range(rollouts)]

    # graphed: ONE capture
    wg = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=31)
    wg.set_opponent(hip.policy.FusedNNPolicy(copy.deepcopy(sources[0]), seed=100))
    bg = hip.Buffer(steps, nenv, (2, m, n), c, device=DEV)
    roll = hip.graphed.GraphedRollout(wg, bg, net, seed=55)  # the constructor's warm-up rollout = rollout 0
    graph = roll.graph
    got = [{f: getattr(bg, f)[:steps].clone() for f in FIELDS}]
    for j in range(1, rollouts):
        if how == "set_opponent_weights":
            roll.set_opponent_weights(sources[j], seed=100 + j)
        else:  # the reference's own line (train.py:114): the captured wrapper adopts the policy's weights and key in place
            kept = wg.opponent_policy
            wg.set_opponent(hip.policy.FusedNNPolicy(copy.deepcopy(sources[j]), seed=100 + j))
            assert wg.opponent_policy is kept
        roll.run()
        got.append({f: getattr(bg, f)[:steps].clone() for f in FIELDS})
    assert roll.graph is graph, "a new capture happened"

    # eager: a fresh FusedNNPolicy(deepcopy) before every rollout
    we = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=31)
    be = hip.Buffer(steps, nenv, (2, m, n), c, device=DEV)
    we.attach_sink(be)
    we.set_opponent(hip.policy.FusedNNPolicy(copy.deepcopy(sources[0]), seed=100))
    agent = hip.policy.HipSampler(seed=55)
    agent.env_id0 = we.env_id0
    obs, _ = we.reset()
    for j in range(rollouts):
        if j:
            we.set_opponent(hip.policy.FusedNNPolicy(copy.deepcopy(sources[j]), seed=100 + j))
        obs = _eager_rollout(hip, we, be, net, agent, obs)
        for f in FIELDS:
            assert torch.equal(getattr(be, f)[:steps], got[j][f]), (j, f)
    nxt = roll.next_obs()
    assert torch.equal(nxt["observation"], obs["observation"]) and torch.equal(nxt["action_mask"], obs["action_mask"])
    assert _same_state(wg, we)


def test_recapture_mid_training_is_one_rollout_of_the_same_stream(hip):
    """``recapture()`` (another KIND of opponent: here a scripted policy replaces the network) plays one real rollout
    as its warm-up: counting it, graphed rollouts before and after equal the eager loop's."""
    m, n, k, nenv, steps = 9, 9, 5, 200, 5
    c = m * n
    net = TinyNet(m, n, 10).to(DEV).eval()
    oppnet = TinyNet(m, n, 11).to(DEV).eval()

    wg = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=7)
    wg.set_opponent(hip.policy.FusedNNPolicy(copy.deepcopy(oppnet), seed=8))
    bg = hip.Buffer(steps, nenv, (2, m, n), c, device=DEV)
    roll = hip.graphed.GraphedRollout(wg, bg, net, seed=9)
    got = [{f: getattr(bg, f)[:steps].clone() for f in FIELDS}]
    roll.run()
    got.append({f: getattr(bg, f)[:steps].clone() for f in FIELDS})
    wg.set_opponent(LowestLegalPolicy())
    roll.recapture()                                   # rollout 2 happens in here
    got.append({f: getattr(bg, f)[:steps].clone() for f in FIELDS})
    roll.run()
    got.append({f: getattr(bg, f)[:steps].clone() for f in FIELDS})

    we = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=7)
    be = hip.Buffer(steps, nenv, (2, m, n), c, device=DEV)
    we.attach_sink(be)
    we.set_opponent(hip.policy.FusedNNPolicy(copy.deepcopy(oppnet), seed=8))
    agent = hip.policy.HipSampler(seed=9)
    obs, _ = we.reset()
    for j in range(4):
        if j == 2:
            we.set_opponent(LowestLegalPolicy())
        obs = _eager_rollout(hip, we, be, net, agent, obs)
        for f in FIELDS:
            assert torch.equal(getattr(be, f)[:steps], got[j][f]), (j, f)


def test_graphed_rollout_checkpoint_continues_its_streams(hip, tmp_path):
    """under a captured rollout the Philox position lives in device counters.  ``state_dict()`` folds
    them in: a GraphedRollout restored from a checkpoint -- and a plain eager wrapper restored from the same wrapper
    state -- continue the side / opponent / agent streams instead of replaying them."""
    m, n, k, nenv, steps = 9, 9, 5, 150, 4
    c = m * n
    net = TinyNet(m, n, 10).to(DEV).eval()
    oppnet = TinyNet(m, n, 11).to(DEV).eval()

    def build():
        w = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=3)
        w.set_opponent(hip.policy.FusedNNPolicy(copy.deepcopy(oppnet), seed=4))
        b = hip.Buffer(steps, nenv, (2, m, n), c, device=DEV)
        return w, b, hip.graphed.GraphedRollout(w, b, net, seed=5)

    w1, b1, r1 = build()
    r1.run()
    r1.set_opponent_weights(net, seed=44)
    r1.run()
    torch.save(r1.state_dict(), tmp_path / "roll.pt")
    want = []
    for _ in range(2):
        r1.run()
        want.append({f: getattr(b1, f)[:steps].clone() for f in FIELDS})
    state = torch.load(tmp_path / "roll.pt", weights_only=False)  # our own file
    assert state["wrapper"]["step_count"] == 1 + 3 * steps and state["steps_done"] == 3 * steps

    w2, b2, r2 = build()
    r2._opp.set_weights(net, seed=1)  # the resumed run's opponent weights are the caller's to restore; the key is ours
    r2.load_state_dict(state)
    for j in range(2):
        r2.run()
        for f in FIELDS:
            assert torch.equal(getattr(b2, f)[:steps], want[j][f]), (j, f)

    # the same wrapper state in a plain eager wrapper: the loop goes on where the graphed one stood
    w3 = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=99)
    w3.load_state_dict(state["wrapper"])
    assert w3.step_count == 1 + 3 * steps
    opp3 = hip.policy.FusedNNPolicy(copy.deepcopy(net), seed=state["opponent"]["seed"])
    opp3._sampler.calls = state["opponent"]["step"]
    w3.set_opponent(opp3)
    b3 = hip.Buffer(steps, nenv, (2, m, n), c, device=DEV)
    w3.attach_sink(b3)
    agent = hip.policy.HipSampler(seed=5)
    agent.calls = state["steps_done"]
    obs = {"observation": state["spill"]["observation"].to(DEV), "action_mask": state["spill"]["action_mask"].to(DEV)}
    obs = _eager_rollout(hip, w3, b3, net, agent, obs)
    for f in FIELDS:
        assert torch.equal(getattr(b3, f)[:steps], want[0][f]), f


def test_graphed_agent_step_swaps_its_opponent_in_place(hip):
    """``GraphedAgentStep.set_opponent_weights``: the one-step-per-graph collector after an in-place opponent swap equals
    the eager loop after ``set_opponent(FusedNNPolicy(deepcopy(source), seed=s))`` -- no new capture."""
    m, n, k, nenv = 9, 9, 5, 130
    net = TinyNet(m, n, 10).to(DEV).eval()
    first, second = TinyNet(m, n, 20).to(DEV).eval(), TinyNet(m, n, 21).to(DEV).eval()
    wg = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=17)
    wg.set_opponent(hip.policy.FusedNNPolicy(copy.deepcopy(first), seed=50))
    col = hip.graphed.GraphedAgentStep(wg, net, seed=60)       # (its warm-up plays 4 steps)
    graphs = list(col.graphs)
    outs = []
    for t in range(12):
        if t == 5:
            col.set_opponent_weights(second, seed=51)
        o = col.step()
        outs.append({key: o[key].clone() for key in ("obs", "mask", "actions", "log_probs", "rewards", "terminated")})
    assert col.graphs == graphs

    we = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=17)
    we.set_opponent(hip.policy.FusedNNPolicy(copy.deepcopy(first), seed=50))
    agent = hip.policy.HipSampler(seed=60)
    obs, _ = we.reset()
    for t in range(4 + 12):
        if t == 4 + 5:
            we.set_opponent(hip.policy.FusedNNPolicy(copy.deepcopy(second), seed=51))
        with torch.no_grad():
            dist, _ = net(obs["observation"], None)
        actions, logp = agent.draw(dist.logits, obs["action_mask"], False, want_logp=True)
        prev = obs
        obs, rew, term, _, _ = we.step(actions)
        if t >= 4:
            o = outs[t - 4]
            assert torch.equal(o["obs"], prev["observation"]) and torch.equal(o["mask"], prev["action_mask"]), t
            assert torch.equal(o["actions"], actions) and torch.equal(o["log_probs"], logp), t
            assert torch.equal(o["rewards"], rew) and torch.equal(o["terminated"], term), t


def test_set_opponent_of_another_kind_recaptures_at_the_next_run(hip):
    """``wrapper.set_opponent`` on a captured wrapper with a policy the graph cannot adopt (no network / another
    architecture): installed as it is, the graph captures again at its next ``run()`` -- whose warm-up IS that rollout --
    and the stream of rollouts still equals the eager loop's.  An ``NNPolicy`` of the captured architecture is adopted
    (weights in place, a fresh key)."""
    m, n, k, nenv, steps = 9, 9, 5, 150, 4
    c = m * n
    net = TinyNet(m, n, 10).to(DEV).eval()
    oppnet = TinyNet(m, n, 11).to(DEV).eval()
    wg = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=7)
    wg.set_opponent(hip.policy.FusedNNPolicy(copy.deepcopy(oppnet), seed=8))
    bg = hip.Buffer(steps, nenv, (2, m, n), c, device=DEV)
    roll = hip.graphed.GraphedRollout(wg, bg, net, seed=9)
    first = roll.graph
    got = [{f: getattr(bg, f)[:steps].clone() for f in FIELDS}]
    wg.set_opponent(LowestLegalPolicy())            # cannot be adopted
    assert isinstance(wg.opponent_policy, LowestLegalPolicy) and roll._stale
    roll.run()                                      # recaptures; plays rollout 1 while doing so
    assert roll.graph is not first and not roll._stale
    got.append({f: getattr(bg, f)[:steps].clone() for f in FIELDS})
    roll.run()
    got.append({f: getattr(bg, f)[:steps].clone() for f in FIELDS})

    we = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=7)
    be = hip.Buffer(steps, nenv, (2, m, n), c, device=DEV)
    we.attach_sink(be)
    we.set_opponent(hip.policy.FusedNNPolicy(copy.deepcopy(oppnet), seed=8))
    agent = hip.policy.HipSampler(seed=9)
    obs, _ = we.reset()
    for j in range(3):
        if j == 1:
            we.set_opponent(LowestLegalPolicy())
        obs = _eager_rollout(hip, we, be, net, agent, obs)
        for f in FIELDS:
            assert torch.equal(getattr(be, f)[:steps], got[j][f]), (j, f)
    # an NNPolicy around a network of the captured architecture: adopted in place (no new capture), weights equal
    w2 = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=1)
    w2.set_opponent(hip.policy.FusedNNPolicy(copy.deepcopy(oppnet), seed=2))
    r2 = hip.graphed.GraphedRollout(w2, hip.Buffer(steps, nenv, (2, m, n), c, device=DEV), net, seed=3)
    graph, kept = r2.graph, w2.opponent_policy
    w2.set_opponent(hip.policy.NNPolicy(copy.deepcopy(net)))
    assert w2.opponent_policy is kept and not r2._stale
    for a, b in zip(kept.model.state_dict().values(), net.state_dict().values()):
        assert torch.equal(a, b)
    r2.run()
    assert r2.graph is graph

"""GPU: randomized differential testing of the HIP env and wrapper against the oracle -- random board shapes,
batch sizes and operation sequences (full and subset steps with legal, occupied, negative and mixed actions; resets by
index list, by bool mask and of everything; pokes through the dense views; wrapper steps with forced sides), every
output and the whole state compared after every operation, bit for bit.  Fixed seeds: a failure reproduces."""
import os

import numpy as np
import pytest
import torch

from oracle.env_torch import OracleVectorEnv
from oracle.packing import pack_boards
from oracle.policies import HighestLegalPolicy, LowestLegalPolicy, MaskHashPolicy
from oracle.rollout import encode_action_log, random_rollout
from oracle.selfplay_torch import OracleSelfPlay

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    import __graft_entry__ as entry

    entry.build_hip()
    entry._ensure_path()
    import mnk_hip
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper

    mnk_hip.load()

    class NS:
        pass

    from selfplay import random_rollout as rr

    ns = NS()
    ns.Env, ns.Wrapper, ns.rollout, ns.lib = TorchVectorMnkEnv, TorchSelfPlayWrapper, rr, mnk_hip
    return ns


def _seeds(default):
    """the suite's seeds, or ``MNK_FUZZ_SEEDS=lo:hi`` for a one-off longer soak on other seeds"""
    spec = os.environ.get("MNK_FUZZ_SEEDS")
    if spec:
        lo, hi = (int(v) for v in spec.split(":"))
        return range(lo, hi)
    return range(default)


_MAX_WRAPPER_STEPS = [10 ** 9]  # (the large-board runs bound the 3 * cells steps of the wrapper fuzzer)


def _shape(rng):
    while True:
        m, n = int(rng.integers(2, 20)), int(rng.integers(2, 20))
        if m * (n + 1) <= 512:
            return m, n, int(rng.integers(1, min(m, n) + 1))


def _same_state(env, ora, where):
    assert torch.equal(env.boards[...].cpu(), ora.boards), f"{where}: boards"
    assert torch.equal(env.current_player.cpu(), ora.current_player), f"{where}: current_player"
    assert torch.equal(env.move_counts.cpu(), ora.move_counts), f"{where}: move_counts"


def _same_obs(a, b, where):
    assert torch.equal(a["observation"].cpu(), b["observation"]), f"{where}: observation"
    assert torch.equal(a["action_mask"].cpu(), b["action_mask"]), f"{where}: action_mask"


@pytest.mark.parametrize("seed", _seeds(10))
def test_env_fuzz(hip, seed):
    rng = np.random.default_rng(7000 + seed)
    m, n, k = _shape(rng)
    nenv = int(rng.choice([1, 2, 3, 31, 64, 65, 130, 257]))
    c = m * n
    env, ora = hip.Env(m, n, k, nenv, device=DEV), OracleVectorEnv(m, n, k, nenv)
    _same_obs(env.reset(), ora.reset(), "reset")
    for t in range(60):
        op = rng.choice(["step", "step", "step", "subset", "subset", "reset_idx", "reset_mask", "reset_all", "poke", "observe"])
        where = f"seed {seed} {m}x{n}x{k} N={nenv} op {t} {op}"
        mask = ora.observe()["action_mask"].numpy()
        acts = np.zeros(nenv, dtype=np.int64)
        for i in range(nenv):
            legal = np.nonzero(mask[i])[0]
            kind = rng.random()
            if len(legal) and kind > 0.2:
                acts[i] = rng.choice(legal)
            elif kind > 0.1:
                acts[i] = rng.integers(0, c)          # possibly occupied: accepted, both planes may end up set
            else:
                acts[i] = rng.integers(-c, 0)         # negative index: wraps like torch indexing
        if op == "step":
            o1, r1, d1 = env.step(torch.from_numpy(acts).to(DEV))
            o2, r2, d2 = ora.step(torch.from_numpy(acts))
        elif op == "subset":
            pick = rng.random(nenv) < rng.choice([0.1, 0.5, 0.9])
            if not pick.any():
                pick[rng.integers(0, nenv)] = True
            idx = torch.from_numpy(np.nonzero(pick)[0])
            o1, r1, d1 = env.step_subset(torch.from_numpy(acts[pick]).to(DEV), idx.to(DEV))
            o2, r2, d2 = ora.step_subset(torch.from_numpy(acts[pick]), idx)
        elif op == "reset_idx":
            idx = torch.from_numpy(np.nonzero(rng.random(nenv) < 0.3)[0])
            _same_obs(env.reset(idx.to(DEV)), ora.reset(idx), where)
            _same_state(env, ora, where)
            continue
        elif op == "reset_mask":
            pick = torch.from_numpy(rng.random(nenv) < 0.3)
            _same_obs(env.reset(pick.to(DEV)), ora.reset(torch.nonzero(pick).squeeze(1)), where)
            _same_state(env, ora, where)
            continue
        elif op == "reset_all":
            _same_obs(env.reset(), ora.reset(), where)
            _same_state(env, ora, where)
            continue
        elif op == "poke":  # what the reference's tests do: env.boards[i, p, r, c] = 1 and friends
            i, p, r, col = int(rng.integers(0, nenv)), int(rng.integers(0, 2)), int(rng.integers(0, m)), int(rng.integers(0, n))
            env.boards[i, p, r, col] = 1
            ora.boards[i, p, r, col] = 1
            side, moves = int(rng.integers(0, 2)), int(rng.integers(0, c))
            env.current_player[i] = side
            ora.current_player[i] = side
            env.move_counts[i] = moves
            ora.move_counts[i] = moves
            _same_state(env, ora, where)
            continue
        else:
            _same_obs(env.observe(), ora.observe(), where)
            assert torch.equal(env.legal_mask().cpu(), ora.observe()["action_mask"]), where
            continue
        _same_obs(o1, o2, where)
        assert torch.equal(r1.cpu(), r2) and torch.equal(d1.cpu(), d2), where
        _same_state(env, ora, where)
    env.check_errors()


class _ForcedSides(OracleSelfPlay):
    """oracle wrapper whose fresh sides come from the array the test sets before every call"""

    def __init__(self, env):
        super().__init__(env, side_source=self._draw)
        self.sides = None
        self._resetting = None

    def _draw(self, count):
        if count == self.num_envs:
            return self.sides.clone()
        return self.sides[torch.nonzero(self._resetting).squeeze(1)]

    def step(self, actions):
        self._resetting = self.pending_resets.clone()
        return super().step(actions)


@pytest.mark.parametrize("seed", _seeds(8))
def test_wrapper_fuzz(hip, seed):
    rng = np.random.default_rng(9000 + seed)
    m, n, k = _shape(rng)
    if k < 2:
        k = 2 if min(m, n) >= 2 else 1
    nenv = int(rng.choice([1, 5, 64, 67, 200]))
    c = m * n
    opp = [LowestLegalPolicy, HighestLegalPolicy, MaskHashPolicy][seed % 3]
    # round 3: the env hands out f32, bf16 or u8 observations, and some steps write into caller-owned tensors
    obs_dtype = [torch.float32, torch.bfloat16, torch.uint8][seed % 3]
    wrap = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV, obs_dtype=obs_dtype), seed=seed)
    mine = {"observation": torch.empty((nenv, 2, m, n), dtype=[torch.uint8, torch.float32, torch.bfloat16][seed % 3], device=DEV),
            "action_mask": torch.empty((nenv, c), dtype=torch.bool, device=DEV),
            "rewards": torch.empty(nenv, dtype=torch.float32, device=DEV),
            "terminated": torch.empty(nenv, dtype=torch.bool, device=DEV),
            "packed": torch.empty((2, wrap.env.words, nenv), dtype=torch.int64, device=DEV)}
    ora = _ForcedSides(OracleVectorEnv(m, n, k, nenv))
    wrap.set_opponent(opp())
    ora.set_opponent(opp())
    from selfplay.policy import HipSampler

    sampler = HipSampler(seed=seed)
    sides = torch.from_numpy(rng.integers(0, 2, nenv))
    wrap.force_sides(sides)
    ora.sides = sides
    o1, _ = wrap.reset()
    o2, _ = ora.reset()
    _same_obs(o1, o2, "reset")
    for t in range(min(3 * c, _MAX_WRAPPER_STEPS[0])):
        where = f"seed {seed} {m}x{n}x{k} N={nenv} step {t}"
        mask = o2["action_mask"].numpy()
        acts = np.array([rng.choice(np.nonzero(row)[0]) if rng.random() > 0.05 else rng.integers(0, c) for row in mask])
        sides = torch.from_numpy(rng.integers(0, 2, nenv))
        wrap.force_sides(sides)
        ora.sides = sides
        out = None
        if rng.random() < 0.4:  # a random subset of the outputs goes to caller-owned tensors
            out = {key: t_ for key, t_ in mine.items() if rng.random() < 0.6}
        if rng.random() < 0.35:
            # round 4: the same moves through the step kernels with the draw folded in (mnk_selfplay_pre_logits; two
            # launches inside the call on boards without a compile-time draw shape): logits whose argmax is the move,
            # an all-True mask (the mask is an input of the draw: occupied cells stay playable as in wrapper.step)
            logits = torch.zeros((nenv, c), dtype=torch.bfloat16 if rng.random() < 0.5 else torch.float32, device=DEV)
            logits[torch.arange(nenv, device=DEV), torch.from_numpy(acts).to(DEV)] = 10.0
            o1, r1, t1, tr1, info = wrap.step_logits(logits, torch.ones((nenv, c), dtype=torch.bool, device=DEV), sampler,
                                                     deterministic=True, out=out)
            assert np.array_equal(info["actions"].cpu().numpy(), acts), where
        else:
            o1, r1, t1, tr1, _ = wrap.step(torch.from_numpy(acts).to(DEV), out=out)
        o2, r2, t2, tr2, _ = ora.step(torch.from_numpy(acts))
        want_dtype = out["observation"].dtype if out and "observation" in out else obs_dtype
        assert o1["observation"].dtype == want_dtype, where
        for key, got in (("observation", o1["observation"]), ("action_mask", o1["action_mask"]), ("rewards", r1), ("terminated", t1)):
            assert (got is out[key]) if out and key in out else True, f"{where}: {key} is not the caller's tensor"
        if out and "packed" in out:
            assert np.array_equal(out["packed"].cpu().numpy().view(np.uint64), pack_boards(o2["observation"].numpy(), m, n)), where
        o1 = {"observation": o1["observation"].float(), "action_mask": o1["action_mask"]}
        _same_obs(o1, o2, where)
        assert torch.equal(r1.cpu(), r2) and torch.equal(t1.cpu(), t2) and not bool(tr1.any()), where
        assert torch.equal(wrap.agent_side.cpu(), ora.agent_side) and torch.equal(wrap.pending_resets.cpu(), ora.pending_resets), where
        _same_state(wrap.env, ora.env, where)
        if t % 7 == 3:
            got = wrap.get_agent_obs()
            assert got["observation"].dtype == obs_dtype
            _same_obs({"observation": got["observation"].float(), "action_mask": got["action_mask"]}, ora.get_agent_obs(),
                      where + " get_agent_obs")
    wrap.env.check_errors()


@pytest.mark.parametrize("seed", _seeds(10))
def test_rollout_and_log_fuzz(hip, seed):
    """Random boards, batch sizes and chunkings of the fused rollout with a random action-log format (byte, 16-bit, 7-bit
    stream where the board allows), message kind (with / without the chunk-start state) and kernel choice (generic or
    run-time specialised): records, statistics and final state == the oracle's raw loop; the log == the oracle's packing
    of the recorded actions; a replay of every chunk -- from the message's own state or from the receiver's running
    state -- rebuilds the records; one-launch plies (mnk_step_random) continue the same game stream ply for ply."""

    rng = np.random.default_rng(11000 + seed)
    m, n, k = _shape(rng)
    if n < 2:
        n = 2
    k = min(k, m, n)
    nenv = int(rng.choice([1, 3, 64, 65, 129, 300]))
    c = m * n
    fmts = [f for f in (hip.rollout.ACT_U16, hip.rollout.ACT_U8, hip.rollout.ACT_BITS7, hip.rollout.ACT_U8P1)
            if hip.rollout.action_log_fits(f, c)]
    saved = os.environ.get("MNK_JIT")
    saved_pair = os.environ.get("MNK_ROLLOUT_PAIR")
    os.environ["MNK_JIT"] = str(seed % 2)  # both kernels for boards without a built-in variant
    # ... and of the run-time specialised ones both forms: one lane per env, two lanes per env (round 4), the launcher's choice
    os.environ.pop("MNK_ROLLOUT_PAIR", None)
    if (seed // 2) % 3:
        os.environ["MNK_ROLLOUT_PAIR"] = str((seed // 2) % 3 - 1)
    hip.lib.reload_config()
    try:
        env, ora = hip.Env(m, n, k, nenv, device=DEV), OracleVectorEnv(m, n, k, nenv)
        roll = hip.rollout.RandomRollout(env, seed=400 + seed, env_id0=17 * seed)
        state = hip.rollout.gather_start_state(env)
        step0 = 0
        chunks = [int(4 * rng.integers(1, 12)) for _ in range(3)] + [int(rng.integers(1, 40))]  # only the last one ragged
        for j, t in enumerate(chunks):
            where = f"seed {seed} {m}x{n}x{k} N={nenv} chunk {j} ({t} plies)"
            fmt = int(rng.choice(fmts))
            with_state = bool(rng.random() < 0.5)
            rec = roll.alloc(t, log_actions=fmt, with_state=with_state)
            roll.run(t, out=rec)
            planes, meta, stats = random_rollout(ora, seed=400 + seed, step0=step0, steps=t, env_id0=17 * seed)
            step0 += t
            assert np.array_equal(rec.planes.cpu().numpy().view(np.uint64), planes), where
            assert np.array_equal(rec.meta.cpu().numpy().view(np.uint32), meta), where
            want_log = encode_action_log((meta & 0xFFFF).astype(np.int64), fmt)
            assert np.array_equal(rec.act.cpu().numpy().view(want_log.dtype), want_log), where + f" log format {fmt}"
            logs = hip.rollout.GatheredLogs.empty(1, env.words if with_state else 0, nenv, t, c, DEV, fmt=fmt, with_state=with_state)
            logs.msg.copy_(rec.msg.unsqueeze(0))
            again = hip.rollout.replay_shard(logs, 0, m, n, k, state=None if with_state else state)
            if with_state:  # the receiver's running state moves on either way
                hip.rollout.replay_shard(logs, 0, m, n, k, state=state, record=False)
            assert torch.equal(again.planes, rec.planes) and torch.equal(again.meta, rec.meta), where + " replay"
            assert torch.equal(state.planes[0], env._planes) and torch.equal(state.meta[0], env._meta), where + " replay state"
            _same_state(env, ora, where)
        # the same stream continued ply by ply through the one-launch step
        rew = torch.empty(nenv, dtype=torch.float32, device=DEV)
        done = torch.empty(nenv, dtype=torch.bool, device=DEV)
        mask = torch.empty((nenv, c), dtype=torch.bool, device=DEV)
        acts = torch.empty(nenv, dtype=torch.long, device=DEV)
        t = int(rng.integers(2, 10))
        planes, meta, _ = random_rollout(ora, seed=400 + seed, step0=step0, steps=t, env_id0=17 * seed)
        for j in range(t):
            env.step_random_into(rew, done, mask, actions=acts, seed=400 + seed, step=step0 + j, env_id0=17 * seed)
            assert np.array_equal(acts.cpu().numpy(), (meta[j] & 0xFFFF).astype(np.int64)), f"seed {seed} one-launch ply {j}"
            assert np.array_equal(done.cpu().numpy(), ((meta[j] >> 24) & 1).astype(bool))
        _same_state(env, ora, f"seed {seed} after the one-launch plies")
        assert torch.equal(mask.cpu(), ora.observe()["action_mask"])
    finally:
        if saved is None:
            os.environ.pop("MNK_JIT", None)
        else:
            os.environ["MNK_JIT"] = saved
        if saved_pair is None:
            os.environ.pop("MNK_ROLLOUT_PAIR", None)
        else:
            os.environ["MNK_ROLLOUT_PAIR"] = saved_pair
        hip.lib.reload_config()


# ----------------------------------------------------------------------------- round 4: boards of up to 1 024 bits
LARGE = [(25, 25, 5), (31, 31, 6), (16, 61, 5), (23, 40, 7), (22, 23, 10)]


@pytest.mark.parametrize("shape", LARGE)
def test_large_boards_through_the_fuzzers(hip, shape, monkeypatch):
    """The reference takes any board (env/torch_vector_mnk_env.py:9).  Round 4 lifted the packed layout from 512 to 1 024
    bits per plane (16 u64 words: 25x25, 31x31, rows of up to 61 cells): the three differential fuzzers on boards beyond
    the old limit -- every env operation, the wrapper (a third of its steps through the folded draw), the rollout with
    its log formats (two bytes per action above 512 cells), replays and the one-launch step, generic and run-time
    specialised kernels -- against the oracle."""
    import sys

    me = sys.modules[__name__]
    monkeypatch.setattr(me, "_shape", lambda rng: shape)
    _MAX_WRAPPER_STEPS[0] = 260
    try:
        base = 500 + 10 * LARGE.index(shape)
        for seed in (base, base + 1):
            test_env_fuzz(hip, seed)
            test_rollout_and_log_fuzz(hip, seed)
        test_wrapper_fuzz(hip, base)
    finally:
        _MAX_WRAPPER_STEPS[0] = 10 ** 9

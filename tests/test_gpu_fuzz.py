"""GPU: randomized differential testing of the HIP env and wrapper against the oracle -- random board shapes,
batch sizes and operation sequences (full and subset steps with legal, occupied, negative and mixed actions; resets by
index list, by bool mask and of everything; pokes through the dense views; wrapper steps with forced sides), every
output and the whole state compared after every operation, bit for bit.  Fixed seeds: a failure reproduces."""
import numpy as np
import pytest
import torch

from oracle.env_torch import OracleVectorEnv
from oracle.policies import HighestLegalPolicy, LowestLegalPolicy, MaskHashPolicy
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

    ns = NS()
    ns.Env, ns.Wrapper = TorchVectorMnkEnv, TorchSelfPlayWrapper
    return ns


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


@pytest.mark.parametrize("seed", range(10))
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


@pytest.mark.parametrize("seed", range(8))
def test_wrapper_fuzz(hip, seed):
    rng = np.random.default_rng(9000 + seed)
    m, n, k = _shape(rng)
    if k < 2:
        k = 2 if min(m, n) >= 2 else 1
    nenv = int(rng.choice([1, 5, 64, 67, 200]))
    c = m * n
    opp = [LowestLegalPolicy, HighestLegalPolicy, MaskHashPolicy][seed % 3]
    wrap = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=seed)
    ora = _ForcedSides(OracleVectorEnv(m, n, k, nenv))
    wrap.set_opponent(opp())
    ora.set_opponent(opp())
    sides = torch.from_numpy(rng.integers(0, 2, nenv))
    wrap.force_sides(sides)
    ora.sides = sides
    o1, _ = wrap.reset()
    o2, _ = ora.reset()
    _same_obs(o1, o2, "reset")
    for t in range(3 * c):
        where = f"seed {seed} {m}x{n}x{k} N={nenv} step {t}"
        mask = o2["action_mask"].numpy()
        acts = np.array([rng.choice(np.nonzero(row)[0]) if rng.random() > 0.05 else rng.integers(0, c) for row in mask])
        sides = torch.from_numpy(rng.integers(0, 2, nenv))
        wrap.force_sides(sides)
        ora.sides = sides
        o1, r1, t1, tr1, _ = wrap.step(torch.from_numpy(acts).to(DEV))
        o2, r2, t2, tr2, _ = ora.step(torch.from_numpy(acts))
        _same_obs(o1, o2, where)
        assert torch.equal(r1.cpu(), r2) and torch.equal(t1.cpu(), t2) and not bool(tr1.any()), where
        assert torch.equal(wrap.agent_side.cpu(), ora.agent_side) and torch.equal(wrap.pending_resets.cpu(), ora.pending_resets), where
        _same_state(wrap.env, ora.env, where)
        if t % 7 == 3:
            _same_obs(wrap.get_agent_obs(), ora.get_agent_obs(), where + " get_agent_obs")
    wrap.env.check_errors()

"""CPU: the oracle against the golden vectors made from the imported reference,
and (when the reference tree is mounted) against the reference itself."""
import numpy as np
import pytest
import torch

from oracle import philox
from oracle.env_torch import OracleVectorEnv
from oracle.packing import pack_boards, pack_cells, unpack_boards, unpack_cells, valid_cell_words, words_per_plane
from oracle.pin_against_reference import ENV_CASES, SELFPLAY_CASES, check_env, check_selfplay, reference_available
from oracle.policies import HighestLegalPolicy, LowestLegalPolicy, MaskHashPolicy
from oracle.selfplay_torch import OracleSelfPlay
from replay import golden_files, play_scenario, replay_env_log, replay_selfplay_trace
from scenarios import SCENARIOS

OPP = {"lowest": LowestLegalPolicy, "highest": HighestLegalPolicy, "hash": MaskHashPolicy}


def test_golden_present(golden_dir):
    assert len(golden_files(golden_dir, "env_")) >= 10
    assert len(golden_files(golden_dir, "selfplay_")) >= 6


@pytest.mark.parametrize("idx", range(11))
def test_oracle_env_oplog(golden_dir, idx):
    path = golden_files(golden_dir, "env_")[idx]
    log = np.load(path)
    m, n, k, nenv, _ = (int(v) for v in log["geom"])
    replay_env_log(OracleVectorEnv(m, n, k, nenv), log)


def _oracle_set_sides(wrapper, sides):
    sides_t = torch.from_numpy(sides.astype(np.int64))

    def source(count):
        if count == wrapper.num_envs:  # reset(): one side per env
            return sides_t.clone()
        return sides_t[torch.nonzero(wrapper._resetting).squeeze(1)]

    wrapper._side_source = source


class _ReplayOracleSelfPlay(OracleSelfPlay):
    def step(self, actions):
        self._resetting = self.pending_resets.clone()
        return super().step(actions)


@pytest.mark.parametrize("idx", range(7))
def test_oracle_selfplay_trace(golden_dir, idx):
    path = golden_files(golden_dir, "selfplay_")[idx]
    log = np.load(path)
    m, n, k, nenv, _ = (int(v) for v in log["geom"])
    opp = path.split("_")[-2]
    wrap = _ReplayOracleSelfPlay(OracleVectorEnv(m, n, k, nenv))
    wrap.set_opponent(OPP[opp]())
    replay_selfplay_trace(wrap, log, _oracle_set_sides)


def test_oracle_on_the_fixtures_of_more_boards(golden_dir):
    """Round 4: env op-logs and wrapper traces recorded from the reference on boards WITHOUT a built-in kernel variant
    (1 / 2 / 4 / 5 register words per plane, a connect-four shape, rows of 33 cells; ``make_golden.py --more-boards``):
    the oracle replays them like the older ones, so they pin it on the boards whose HIP kernels are compiled at run
    time (tests/test_gpu_jit_api.py replays the same files on those)."""
    envs, traces = golden_files(golden_dir, "boards_env_"), golden_files(golden_dir, "boards_selfplay_")
    assert len(envs) == 5 and len(traces) == 4
    for path in envs:
        log = np.load(path)
        m, n, k, nenv, _ = (int(v) for v in log["geom"])
        replay_env_log(OracleVectorEnv(m, n, k, nenv), log)
    for path in traces:
        log = np.load(path)
        m, n, k, nenv, _ = (int(v) for v in log["geom"])
        wrap = _ReplayOracleSelfPlay(OracleVectorEnv(m, n, k, nenv))
        wrap.set_opponent(OPP[path.split("_")[-2]]())
        replay_selfplay_trace(wrap, log, _oracle_set_sides)


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_oracle_edge_scenarios(golden_dir, name):
    sc = SCENARIOS[name]
    want = np.load(f"{golden_dir}/edges.npz")[name]
    got = play_scenario(OracleVectorEnv(sc["m"], sc["n"], sc["k"], 1), sc)
    assert np.array_equal(got, want)


def test_reference_test_expectations():
    """What the reference's own tests assert (src/tests/test_mnk_integration.py:50-65, 117-132)."""
    want = SCENARIOS["row_win_3x3"]
    rows = play_scenario(OracleVectorEnv(3, 3, 3, 1), want)
    reward, done = rows[0][-4], rows[0][-3]
    assert reward == 1 and done == 1


def test_packing_roundtrip():
    rng = np.random.default_rng(0)
    for (m, n) in [(3, 3), (4, 6), (9, 9), (13, 13), (19, 19), (7, 9)]:
        b = (rng.random((17, 2, m, n)) < 0.4).astype(np.float32)
        p = pack_boards(b, m, n)
        assert p.shape == (2, words_per_plane(m, n), 17) and p.dtype == np.uint64
        assert np.array_equal(unpack_boards(p, m, n), b)
        # guard column and padding bits stay clear
        assert not np.any(p & ~valid_cell_words(m, n)[None, :, None])
        cells = rng.random((5, m * n)) < 0.5
        assert np.array_equal(unpack_cells(pack_cells(cells, m, n), m, n), cells.astype(np.uint8))
    assert [words_per_plane(*s) for s in [(3, 3), (9, 9), (13, 13), (19, 19)]] == [1, 2, 3, 6]


def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32-10 (kat_vectors of the Random123 distribution)."""
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
         (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    for ctr, key, want in kat:
        got = philox.philox4x32_10(*ctr, *key)
        assert tuple(int(v) for v in got) == want


def test_pick_legal_is_uniform_and_legal():
    rng = np.random.default_rng(1)
    mask = rng.random((4000, 9)) < 0.6
    mask[0] = False
    x = philox.rand_u32(7, np.arange(4000), 0, philox.STREAM_MOVE)
    a = philox.pick_legal(mask, x)
    rows = np.nonzero(mask.any(axis=1))[0]
    assert mask[rows, a[rows]].all()
    assert 0 <= a[0] < 9
    one = np.zeros((60000, 9), dtype=bool)
    one[:, [1, 4, 7]] = True
    a = philox.pick_legal(one, philox.rand_u32(3, np.arange(60000), 5, philox.STREAM_MOVE))
    counts = np.bincount(a, minlength=9)[[1, 4, 7]]
    assert counts.sum() == 60000 and np.all(np.abs(counts - 20000) < 600)


@pytest.mark.skipif(not reference_available(), reason="reference tree not mounted (GPU box)")
@pytest.mark.parametrize("case", ENV_CASES[:5])
def test_oracle_env_equals_reference(case):
    assert check_env(*case)


@pytest.mark.skipif(not reference_available(), reason="reference tree not mounted (GPU box)")
@pytest.mark.parametrize("case", SELFPLAY_CASES[:6])
def test_oracle_selfplay_equals_reference(case):
    assert check_selfplay(*case)


def test_oracle_canonical_view_of_policy_fixture(golden_dir):
    """G6 fixture: the oracle's canonical observation of the stored position equals what the reference handed
    to its network."""
    g = np.load(f"{golden_dir}/policy_cnn_b_s.npz")
    env = OracleVectorEnv(9, 9, 5, 64)
    env.boards.copy_(torch.from_numpy(unpack_boards(g["planes"], 9, 9)))
    env.current_player.copy_(torch.from_numpy(g["meta_side"].astype(np.int64)))
    env.move_counts.copy_(torch.from_numpy(g["meta_moves"].astype(np.int64)))
    wrap = OracleSelfPlay(env)
    wrap.agent_side.copy_(torch.from_numpy(g["agent_side"].astype(np.int64)))
    obs = wrap.canonical_obs()
    assert np.array_equal(pack_boards(obs["observation"].numpy(), 9, 9), g["obs_planes"])
    assert np.array_equal(pack_cells(obs["action_mask"].numpy(), 9, 9), g["obs_mask"])
    assert np.isfinite(g["logp"]).sum(axis=1).tolist() == obs["action_mask"].sum(dim=1).tolist()


@pytest.mark.skipif(not reference_available(), reason="reference tree not mounted (GPU box)")
def test_oracle_equals_reference_on_random_geometries():
    """Property test: random board shapes, batch sizes and seeds; oracle == imported
    reference after every operation, env and wrapper."""
    from hypothesis import HealthCheck, given, settings, strategies as st

    @settings(max_examples=12, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(m=st.integers(2, 12), n=st.integers(2, 12), kk=st.integers(1, 6), nenv=st.integers(1, 24),
           seed=st.integers(0, 10 ** 6))
    def run(m, n, kk, nenv, seed):
        k = min(kk, m, n)
        assert check_env(m, n, k, nenv, 3 * max(m, n), seed)
        assert check_selfplay(m, n, k, nenv, 2 * max(m, n), seed, "random" if seed % 2 else "hash", None)

    run()

"""CPU: the oracle's restatements of the callers and the sink (oracle/callers.py, oracle/rollout.py:gae) against
the fixtures recorded from the imported reference by tests/golden/make_golden_callers.py -- SURVEY.md section 8
rows a18 (PPOAgent.learn rollout), a19/f1 (RolloutBuffer + GAE), a20 (validate_gpu), f3 (tournament loop)."""
import os

import numpy as np
import pytest

from oracle.callers import OracleRolloutBuffer, oracle_play_batch_games, oracle_validate
from oracle.env_torch import OracleVectorEnv
from oracle.policies import (HighestLegalPolicy, LowestLegalPolicy, MaskHashPolicy, OpeningByRowPolicy,
                             RowSaltedHashPolicy)
from oracle.rollout import gae as oracle_gae
from replay import golden_files, replay_ppo_learn
from test_oracle_golden import _oracle_set_sides, _ReplayOracleSelfPlay


def named_policy(name):
    """the policy names used in the keys of validate.npz / tournament.npz"""
    if name == "lowest":
        return LowestLegalPolicy()
    if name == "highest":
        return HighestLegalPolicy()
    for prefix, cls in (("rowhash", RowSaltedHashPolicy), ("hash", MaskHashPolicy), ("open", OpeningByRowPolicy)):
        if name.startswith(prefix):
            return cls(int(name[len(prefix):]))
    raise KeyError(name)


def fixture_cases(golden_dir, name):
    """key -> (m, n, k, count, policy a, policy b) of validate.npz / tournament.npz"""
    data = np.load(os.path.join(golden_dir, name))
    out = {}
    for key in data.files:
        board, count, a, b = key.split("_")
        m, n, k = (int(v) for v in board.split("x"))
        out[key] = (m, n, k, int(count), a, b, data[key])
    return out


def gae_cases(golden_dir):
    data = np.load(os.path.join(golden_dir, "gae.npz"))
    names = sorted({f.split("/")[0] for f in data.files})
    return data, names


def test_oracle_gae_equals_the_reference_buffer(golden_dir):
    """oracle/rollout.py:gae == RolloutBuffer.compute_advantages_and_returns (rollout_buffer.py:60-80), bit for bit"""
    data, names = gae_cases(golden_dir)
    assert len(names) == 5
    for name in names:
        n_steps, steps, nenv, gamma, lam = data[name + "/hyper"]
        steps = int(steps)
        adv, ret = oracle_gae(data[name + "/rewards"], data[name + "/values"], data[name + "/dones"],
                              data[name + "/last_values"], float(gamma), float(lam))
        assert np.array_equal(adv, data[name + "/advantages"][:steps]), name
        assert np.array_equal(ret, data[name + "/returns"][:steps]), name
        assert not data[name + "/advantages"][steps:].any() and not data[name + "/returns"][steps:].any()


def test_oracle_validate_equals_the_reference(golden_dir):
    cases = fixture_cases(golden_dir, "validate.npz")
    assert len(cases) == 6
    for key, (m, n, k, episodes, agent, opp, want) in cases.items():
        res = oracle_validate(named_policy(agent), named_policy(opp), (m, n, k), n_episodes=episodes)
        got = [res[f"validation/vs_benchmark/{f}"] for f in ("win_rate", "loss_rate", "draw_rate", "score_rate",
                                                             "games_played")]
        assert got == want.tolist(), key


def test_oracle_tournament_equals_the_reference(golden_dir):
    cases = fixture_cases(golden_dir, "tournament.npz")
    assert len(cases) == 5
    for key, (m, n, k, games, p1, p2, want) in cases.items():
        for row, p1_black in enumerate((True, False)):
            got = oracle_play_batch_games(named_policy(p1), named_policy(p2), (m, n, k), games, p1_black)
            assert list(got) == want[row].tolist(), (key, p1_black)
            assert sum(got) == games


@pytest.mark.parametrize("idx", range(2))
def test_oracle_ppo_learn_rollout(golden_dir, idx):
    path = golden_files(golden_dir, "ppo_learn_")[idx]
    log = np.load(path)
    m, n, k, nenv, _ = (int(v) for v in log["geom"])
    wrap = _ReplayOracleSelfPlay(OracleVectorEnv(m, n, k, nenv))
    wrap.set_opponent(MaskHashPolicy(0 if m == 3 else 1))
    assert log["call0/dones"].any() and log["call1/dones"].any()  # games end inside both learn calls
    replay_ppo_learn(wrap, OracleRolloutBuffer, log, _oracle_set_sides)


def test_oracle_ppo_learn_rollout_on_a_board_without_a_built_in_variant(golden_dir):
    """Round 4: the same recording on 6x7x4 (``make_golden_callers.py --more-boards``) -- a board whose HIP kernels are
    compiled at run time; tests/test_gpu_jit_api.py replays it on those, through the sink"""
    (path,) = golden_files(golden_dir, "boards_ppo_learn_")
    log = np.load(path)
    m, n, k, nenv, _ = (int(v) for v in log["geom"])
    assert (m, n, k) == (6, 7, 4)
    wrap = _ReplayOracleSelfPlay(OracleVectorEnv(m, n, k, nenv))
    wrap.set_opponent(MaskHashPolicy(1))
    assert log["call0/dones"].any() and log["call1/dones"].any()
    replay_ppo_learn(wrap, OracleRolloutBuffer, log, _oracle_set_sides)

"""The C ABI without Python in the loop: ``examples/c_host.c`` (plain C99, the HIP runtime's C API, built by
``__graft_entry__.build()``) runs the fused rollout and the one-launch-per-ply loop on its own device buffers; its
statistics and checksums must equal what the ctypes binding gets for the same board, batch and seed -- and what the
oracle gets where the oracle finishes in seconds."""
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle.env_torch import OracleVectorEnv
from oracle.packing import pack_boards
from oracle.rollout import random_rollout
from test_gpu_callers import hip  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _weighted(words):
    """sum of word[i] * (2i + 1) modulo 2^64 -- the checksum of examples/c_host.c"""
    w = np.ascontiguousarray(words).reshape(-1).astype(np.uint64)
    with np.errstate(over="ignore"):
        return int((w * (2 * np.arange(w.size, dtype=np.uint64) + 1)).sum(dtype=np.uint64))


def _run_c_host(*args):
    import __graft_entry__ as entry

    exe = entry.build_c_host()
    res = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    return {k: int(v) for k, v in (kv.split("=") for kv in res.stdout.split())}


@pytest.mark.parametrize("m,n,k,nenv,plies,seed", [(9, 9, 5, 4096, 64, 7), (3, 3, 3, 1000, 40, 1), (19, 19, 5, 300, 48, 3),
                                                   (7, 9, 7, 130, 32, 5)])
def test_c_host_equals_the_python_binding(hip, m, n, k, nenv, plies, seed):  # noqa: F811
    got = _run_c_host(m, n, k, nenv, plies, seed)
    assert got["per_ply_state_equals_rollout"] == 1
    env = hip.Env(m, n, k, nenv, device=DEV)
    env.reset()
    roll = hip.rollout.RandomRollout(env, seed=seed)
    rec = roll.run(plies)
    torch.cuda.synchronize()
    stats = [int(v) for v in roll.stats.cpu()]
    assert [got[f] for f in ("episodes", "black_wins", "white_wins", "draws", "length_sum")] == stats
    assert got["per_ply_finished"] == stats[0] == got["dones_in_records"] and got["per_ply_wins"] == stats[1] + stats[2]
    assert got["planes_sum"] == _weighted(env._planes.cpu().numpy().view(np.uint64))
    assert got["meta_sum"] == _weighted(env._meta.cpu().numpy().view(np.uint32))
    assert got["rec_planes_sum"] == _weighted(rec.planes.cpu().numpy().view(np.uint64))
    assert got["rec_meta_sum"] == _weighted(rec.meta.cpu().numpy().view(np.uint32))
    assert got["per_ply_legal_last"] == int(env.observe()["action_mask"].sum().item())
    # part (C), ABI v5: self-play agent-steps with the draw folded into the step kernel == the wrapper's step_logits
    from selfplay.policy import HipSampler, RandomPolicy
    from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper

    wrap = TorchSelfPlayWrapper(hip.Env(m, n, k, nenv, device=DEV), seed=seed)
    wrap.set_opponent(RandomPolicy(m * n, seed=0))  # (folds into the step kernel: draws on the wrapper's OPP stream)
    obs, _ = wrap.reset()
    sampler = HipSampler(seed=seed + 1)
    term_count, reward_sum = 0, 0
    for _ in range(got["sp_steps"]):
        obs, rew, term, _, info = wrap.step_logits(None, obs["action_mask"], sampler)
        term_count += int(term.sum().item())
        reward_sum += int(rew.sum().item())
    assert got["sp_terminated"] == term_count and got["sp_reward_sum"] == reward_sum
    assert got["sp_actions_sum"] == _weighted(info["actions"].cpu().numpy().view(np.uint64))
    assert got["sp_planes_sum"] == _weighted(wrap.env._planes.cpu().numpy().view(np.uint64))
    assert got["sp_meta_sum"] == _weighted(wrap.env._meta.cpu().numpy().view(np.uint32))


def test_c_host_equals_the_oracle():
    """the same program against the CPU oracle's rollout (oracle/rollout.py: Philox + the reference's env ops)"""
    m, n, k, nenv, plies, seed = 9, 9, 5, 256, 48, 11
    got = _run_c_host(m, n, k, nenv, plies, seed)
    ora = OracleVectorEnv(m, n, k, nenv)
    ora.reset()
    rec_planes, rec_meta, stats = random_rollout(ora, seed, 0, plies)
    assert [got[f] for f in ("episodes", "black_wins", "white_wins", "draws", "length_sum")] == [int(v) for v in stats]
    assert got["rec_planes_sum"] == _weighted(rec_planes) and got["rec_meta_sum"] == _weighted(rec_meta)
    assert got["planes_sum"] == _weighted(pack_boards(ora.boards.numpy(), m, n))

"""CPU, world_size 2 over gloo: the env axis shards across ranks and the all-gather of the
packed rollout records rebuilds exactly what one process with all the envs produces.

The records here come from the oracle (there is no GPU in this container); what is under
test is the product's sharding rule (global env id = rank * N + i keys the RNG) and
``selfplay.random_rollout.gather_records`` -- the same function the GPU ranks call with the
``nccl`` backend (RCCL over xGMI)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as entry


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, m, n, k, nenv, steps, seed, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    entry._ensure_path()
    from oracle.env_torch import OracleVectorEnv
    from oracle.rollout import random_rollout
    from selfplay.random_rollout import RolloutRecords, gather_records

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        env = OracleVectorEnv(m, n, k, nenv)
        planes, meta, stats = random_rollout(env, seed=seed, step0=0, steps=steps, env_id0=rank * nenv)
        rec = RolloutRecords(planes=torch.from_numpy(planes.view(np.int64)), meta=torch.from_numpy(meta.view(np.int32)))
        full = gather_records(rec)
        assert full.planes.shape == (steps, planes.shape[1], world * nenv)
        assert full.meta.shape == (steps, world * nenv)
        # the compact exchange format: chunk-start state + action log
        from selfplay.random_rollout import gather_action_logs
        from selfplay.random_rollout import unpack_action_log
        acts = (meta & 0xFFFF).astype(np.int64)
        pad = np.zeros(((-steps) % 4, nenv), dtype=np.int64)
        quads = np.concatenate([acts, pad]).reshape(-1, 4, nenv)
        from selfplay.random_rollout import _msg_views, _msg_words
        from oracle.packing import words_per_plane
        words = words_per_plane(m, n)
        rec.msg = torch.zeros(_msg_words(words, nenv, steps, m * n), dtype=torch.int64)
        rec.planes0, rec.act, rec.meta0 = _msg_views(rec.msg, words, nenv, steps, m * n)
        rec.act.copy_(torch.from_numpy((quads[:, 0] | quads[:, 1] << 8 | quads[:, 2] << 16 | quads[:, 3] << 24).astype(np.int32)))
        rec.planes0.zero_()  # state layout [2, W, N]; every env starts from an empty board
        assert not rec.planes[0].any()
        rec.meta0.zero_()  # every env starts from reset
        logs = gather_action_logs(rec)
        assert logs.act.shape == (world, (steps + 3) // 4, nenv) and logs.planes0.shape[0] == world and logs.steps == steps
        assert torch.equal(logs.act[rank], rec.act) and torch.equal(logs.planes0[rank], rec.planes0)
        for r in range(world):
            assert torch.equal(unpack_action_log(logs.act[r], steps),
                               (full.meta[:, r * nenv:(r + 1) * nenv] & 0xFFFF).to(torch.int64))
        totals = torch.from_numpy(stats)
        dist.all_reduce(totals)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), planes=full.planes.numpy(), meta=full.meta.numpy(),
                 stats=totals.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("m,n,k,nenv,steps", [(3, 3, 3, 16, 24), (9, 9, 5, 24, 70)])
def test_sharded_rollout_gathers_to_the_single_process_result(tmp_path, m, n, k, nenv, steps):
    world, seed = 2, 13
    mp.spawn(_worker, args=(world, _free_port(), m, n, k, nenv, steps, seed, str(tmp_path)), nprocs=world, join=True)

    from oracle.env_torch import OracleVectorEnv
    from oracle.rollout import random_rollout

    planes, meta, stats = random_rollout(OracleVectorEnv(m, n, k, world * nenv), seed=seed, step0=0, steps=steps)
    for rank in range(world):
        got = np.load(tmp_path / f"rank{rank}.npz")
        assert np.array_equal(got["planes"].view(np.uint64), planes), f"rank {rank}: gathered boards differ"
        assert np.array_equal(got["meta"].view(np.uint32), meta), f"rank {rank}: gathered records differ"
        assert np.array_equal(got["stats"], stats)


def test_gather_is_identity_for_one_rank(tmp_path):
    entry._ensure_path()
    from selfplay.random_rollout import RolloutRecords, gather_records

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        rec = RolloutRecords(planes=torch.zeros((3, 3, 8), dtype=torch.int64), meta=torch.ones((3, 8), dtype=torch.int32))
        assert gather_records(rec) is rec
    finally:
        dist.destroy_process_group()

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
        # the compact exchange formats: the action log as bytes / a 7-bit stream, with the chunk-start state in the
        # message or alone (then every rank holds every shard's replay state, gathered once before the first chunk)
        from oracle.packing import words_per_plane
        from oracle.rollout import encode_action_log
        from selfplay.random_rollout import (ACT_BITS7, ACT_U8, _msg_views, _msg_words, action_log_words,
                                             gather_action_logs, unpack_action_log)
        acts = (meta & 0xFFFF).astype(np.int64)
        words = words_per_plane(m, n)
        for fmt in ([ACT_U8, ACT_BITS7] if m * n <= 128 else [ACT_U8]):
            for with_state in (True, False):
                rec.fmt = fmt
                rec.msg = torch.zeros(_msg_words(words, nenv, steps, fmt, with_state), dtype=torch.int64)
                rec.planes0, rec.act, rec.meta0 = _msg_views(rec.msg, words, nenv, steps, fmt, with_state)
                rec.act.copy_(torch.from_numpy(encode_action_log(acts, fmt).view(np.int32)))
                if with_state:
                    rec.planes0.zero_()  # state layout [2, W, N]; every env starts from an empty board
                    assert not rec.planes[0].any()
                    rec.meta0.zero_()  # every env starts from reset
                else:
                    assert rec.planes0 is None and rec.msg.numel() * 8 == (action_log_words(fmt, steps) * nenv * 4 + 7) // 8 * 8
                logs = gather_action_logs(rec)
                assert logs.act.shape == (world, action_log_words(fmt, steps), nenv) and logs.steps == steps and logs.fmt == fmt
                assert (logs.planes0 is not None) == with_state and (not with_state or logs.planes0.shape[0] == world)
                assert torch.equal(logs.act[rank], rec.act)
                for r in range(world):
                    assert torch.equal(unpack_action_log(logs.act[r], steps, fmt),
                                       (full.meta[:, r * nenv:(r + 1) * nenv] & 0xFFFF).to(torch.int64))
        # the replay state every rank keeps when the log travels alone: gathered once, before the first chunk
        from types import SimpleNamespace

        from selfplay.random_rollout import gather_start_state
        fake_env = SimpleNamespace(num_envs=nenv, words=words, _dev=torch.device("cpu"),
                                   _planes=torch.full((2, words, nenv), rank + 1, dtype=torch.int64),
                                   _meta=torch.full((nenv,), rank + 5, dtype=torch.int32))
        start = gather_start_state(fake_env)
        assert start.planes.shape == (world, 2, words, nenv) and start.meta.shape == (world, nenv)
        for r in range(world):
            assert bool((start.planes[r] == r + 1).all()) and bool((start.meta[r] == r + 5).all())
        if m * n <= 128:  # 0.875 B per env-step on the wire (+ nothing else once the receivers hold the state)
            assert action_log_words(ACT_BITS7, 256) * 4 / 256 == 0.875
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


def _fields_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    entry._ensure_path()
    from alg.packed_rollout_buffer import all_gather_fields

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t, n = 5, 7
        g = torch.Generator().manual_seed(100 + rank)
        fields = {"planes": torch.randint(-2 ** 40, 2 ** 40, (t, 2, 2, n), generator=g),
                  "actions": torch.randint(0, 81, (t, n), generator=g),
                  "log_probs": torch.randn(t, n, generator=g),
                  "dones": torch.rand(t, n, generator=g) < 0.3}
        out = all_gather_fields(fields)
        again = all_gather_fields(fields, out=out)          # the buffers of a previous call are reused
        assert all(again[k] is out[k] for k in fields)
        try:
            all_gather_fields({"planes": fields["planes"]}, out={"planes": torch.empty((t, 2, 2, n), dtype=torch.int64)})
            raise SystemExit("a receive buffer for one rank was accepted")
        except ValueError:
            pass
        try:
            all_gather_fields({"x": fields["planes"].transpose(0, 3)})
            raise SystemExit("a non-contiguous field was accepted")
        except ValueError:
            pass
        torch.save({"mine": fields, "all": out}, os.path.join(out_dir, f"fields{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_buffer_fields_all_gather_rank_major(tmp_path):
    """``alg.packed_rollout_buffer.all_gather_fields`` -- the exchange step of a sharded self-play rollout buffer
    (``PackedRolloutBuffer.all_gather``: SURVEY 8e's all-gather of rollout buffers with the log-probabilities and values
    a network policy adds) on plain tensors, world size 2 over gloo: every rank ends up with every rank's steps, rank r's
    at ``[r * T, (r + 1) * T)``, int64 / f32 / bool alike."""
    world = 2
    mp.spawn(_fields_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(tmp_path / f"fields{r}.pt") for r in range(world)]
    for name in ("planes", "actions", "log_probs", "dones"):
        want = torch.cat([got[r]["mine"][name] for r in range(world)], dim=0)
        for r in range(world):
            assert got[r]["all"][name].dtype == want.dtype and torch.equal(got[r]["all"][name], want), (name, r)


def test_action_log_formats_round_trip_between_the_oracle_and_the_host_code():
    """Every log format: the oracle's numpy packing of random actions, read back by the product's torch unpacker
    (``selfplay.random_rollout.unpack_action_log``), gives the actions; word counts agree with the library's."""
    entry._ensure_path()
    import mnk_hip
    from oracle.rollout import decode_action_log, encode_action_log
    from selfplay.random_rollout import ACT_BITS7, ACT_U8, ACT_U8P1, ACT_U16, action_log_words, unpack_action_log

    rng = np.random.default_rng(3)
    lib = mnk_hip.load()
    for fmt, cells in ((ACT_U8, 256), (ACT_U16, 484), (ACT_BITS7, 128), (ACT_U8P1, 484)):
        for steps in (1, 3, 4, 5, 31, 32, 33, 100, 256):
            acts = rng.integers(0, cells, (steps, 7))
            log = encode_action_log(acts, fmt)
            assert np.array_equal(decode_action_log(log, steps, fmt), acts)
            as_torch = torch.from_numpy(log.view(np.int64 if fmt == ACT_U16 else np.int32).copy())
            assert torch.equal(unpack_action_log(as_torch, steps, fmt), torch.from_numpy(acts)), (fmt, steps)
            words = log.shape[0] * (2 if fmt == ACT_U16 else 1)
            assert words == action_log_words(fmt, steps) == lib.mnk_action_log_words(fmt, steps), (fmt, steps)


def test_exchange_message_layout_is_a_partition_of_the_buffer():
    """The one flat int64 message that is all-gathered -- [planes0 |] log [| meta0] -- for every log format, with and
    without the chunk-start state, odd env counts and ragged chunk lengths: the views tile the buffer without overlap
    (writing a distinct value through each view and reading the flat buffer back), every part starts on an 8-byte
    boundary, and the leading (rank) dimension of a gathered buffer is carried through."""
    entry._ensure_path()
    from selfplay.random_rollout import (ACT_BITS7, ACT_U8, ACT_U8P1, ACT_U16, _msg_layout, _msg_views, _msg_words,
                                         action_log_words)

    rng = np.random.default_rng(11)
    for fmt in (ACT_U8, ACT_U16, ACT_BITS7, ACT_U8P1):
        for with_state in (True, False):
            for _ in range(6):
                words, nenv, steps = int(rng.integers(1, 7)), int(rng.integers(1, 70)), int(rng.integers(1, 300))
                total = _msg_words(words, nenv, steps, fmt, with_state)
                assert total == sum(_msg_layout(words, nenv, steps, fmt, with_state))
                for lead in ((), (3,)):
                    msg = torch.zeros(lead + (total,), dtype=torch.int64)
                    planes0, act, meta0 = _msg_views(msg, words, nenv, steps, fmt, with_state)
                    assert (planes0 is not None) == with_state == (meta0 is not None)
                    if fmt == ACT_U16:
                        assert act.dtype == torch.int64 and act.shape == lead + ((steps + 3) // 4, nenv)
                    else:
                        assert act.dtype == torch.int32 and act.shape == lead + (action_log_words(fmt, steps), nenv)
                    act.fill_(-1)
                    flat_bytes = msg.view(torch.uint8).reshape(lead + (-1,))
                    n_planes, n_act, n_meta = _msg_layout(words, nenv, steps, fmt, with_state)
                    lo, hi = n_planes * 8, n_planes * 8 + act.numel() // max(1, int(np.prod(lead))) * act.element_size()
                    assert bool((flat_bytes[..., lo:hi] == 255).all()) and int((flat_bytes == 255).sum()) == act.numel() * act.element_size()
                    if with_state:
                        planes0.fill_(0x0101010101010101)
                        meta0.fill_(0x02020202)
                        assert planes0.shape == lead + (2, words, nenv) and meta0.shape == lead + (nenv,)
                        assert int((flat_bytes == 1).sum()) == planes0.numel() * 8 and int((flat_bytes == 2).sum()) == meta0.numel() * 4
                        assert bool((flat_bytes[..., :lo] == 1).all())
                        assert bool((flat_bytes[..., (n_planes + n_act) * 8:(n_planes + n_act) * 8 + nenv * 4] == 2).all())


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


def _group_worker(rank, world, port, m, n, k, nenv, steps, chunks, seed, out_dir):
    """``bench.py --exchange-every J`` on the CPU: J chunks' messages (a keyframe + J-1 logs, or all keyframes) travel
    in ONE all-gather (``selfplay.random_rollout.LogGroup``); what every rank reads out of the gathered buffer are the
    chunk-start states and logs of every shard, chunk by chunk."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    entry._ensure_path()
    from oracle.env_torch import OracleVectorEnv
    from oracle.packing import pack_boards, words_per_plane
    from oracle.rollout import encode_action_log, random_rollout
    from selfplay.random_rollout import ACT_BITS7, ACT_U8, KeyframedLogs, LogGroup, unpack_action_log

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        words = words_per_plane(m, n)
        fmt = ACT_BITS7 if m * n <= 128 else ACT_U8
        got = {}
        for name, pattern in (("key_first", (True,) + (False,) * (chunks - 1)), ("all_keys", (True,) * chunks)):
            env = OracleVectorEnv(m, n, k, nenv)
            group = LogGroup(world, words, nenv, steps, fmt, pattern, torch.device("cpu"))
            assert len(group) == chunks and group.bytes == group.send.numel() * 8
            for j in range(chunks):
                rec = group.records(j)
                assert (rec.planes0 is not None) == pattern[j]
                if pattern[j]:
                    rec.planes0.copy_(torch.from_numpy(pack_boards(env.boards.numpy(), m, n).view(np.int64)))
                    rec.meta0.copy_((env.current_player.to(torch.int32) | (env.move_counts.to(torch.int32) << 1)))
                _, meta, _ = random_rollout(env, seed=seed, step0=j * steps, steps=steps, env_id0=rank * nenv)
                rec.act.copy_(torch.from_numpy(encode_action_log((meta & 0xFFFF).astype(np.int64), fmt).view(np.int32)))
            group.gather()
            # the receiving side of a keyframed stream takes the chunks as they are
            stream = KeyframedLogs(m, n, k)
            for j in range(chunks):
                stream.push(group.logs(j))
            assert stream.chunks() == (chunks if name == "key_first" else 1)
            got[name + "_acts"] = torch.stack([torch.cat([unpack_action_log(group.logs(j).act[r], steps, fmt)
                                                          for r in range(world)], dim=1) for j in range(chunks)]).numpy()
            got[name + "_planes0"] = torch.stack([torch.cat([group.logs(j).planes0[r] for r in range(world)], dim=2)
                                                  for j in range(chunks) if pattern[j]]).numpy()
            got[name + "_meta0"] = torch.stack([torch.cat([group.logs(j).meta0[r] for r in range(world)])
                                                for j in range(chunks) if pattern[j]]).numpy()
        np.savez(os.path.join(out_dir, f"group{rank}.npz"), **got)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("m,n,k,nenv,steps,chunks", [(3, 3, 3, 12, 8, 2), (9, 9, 5, 10, 20, 3)])
def test_grouped_exchange_equals_the_single_process_result(tmp_path, m, n, k, nenv, steps, chunks):
    """--exchange-every J: one all-gather per J chunks == what one process holding all the envs plays."""
    world, seed = 2, 5
    mp.spawn(_group_worker, args=(world, _free_port(), m, n, k, nenv, steps, chunks, seed, str(tmp_path)), nprocs=world,
             join=True)
    from oracle.env_torch import OracleVectorEnv
    from oracle.packing import pack_boards
    from oracle.rollout import random_rollout

    env = OracleVectorEnv(m, n, k, world * nenv)
    acts, planes0, meta0 = [], [], []
    for j in range(chunks):
        planes0.append(pack_boards(env.boards.numpy(), m, n).view(np.int64))
        meta0.append((env.current_player.to(torch.int32) | (env.move_counts.to(torch.int32) << 1)).numpy())
        _, meta, _ = random_rollout(env, seed=seed, step0=j * steps, steps=steps)
        acts.append((meta & 0xFFFF).astype(np.int64))
    for rank in range(world):
        got = np.load(tmp_path / f"group{rank}.npz")
        for name in ("key_first", "all_keys"):
            assert np.array_equal(got[name + "_acts"], np.stack(acts)), (rank, name)
            keys = range(chunks) if name == "all_keys" else (0,)
            assert np.array_equal(got[name + "_planes0"], np.stack([planes0[j] for j in keys])), (rank, name)
            assert np.array_equal(got[name + "_meta0"], np.stack([meta0[j] for j in keys])), (rank, name)

"""Oracle for the fused random-policy rollout and for GAE (test infrastructure).

``random_rollout`` restates, on top of ``OracleVectorEnv`` and the numpy Philox
sampler, the loop the reference's random rollout runs
("raw random-vs-random loop"):

    a = RandomPolicy.act(obs)        # /root/reference/src/selfplay/policy.py:18-29
    obs, r, d = env.step(a)          # src/env/torch_vector_mnk_env.py:55-84
    env.reset(nonzero(d))            # src/env/torch_vector_mnk_env.py:34-44

with the one substitution documented in ``philox.py``: the uniform legal move
comes from Philox keyed by (seed, global env id, step) instead of
``torch.multinomial``.  The HIP kernel ``mnk_rollout_random`` must reproduce the
records and the final state of this function bit for bit.

``gae`` restates ``RolloutBuffer.compute_advantages_and_returns``
(``src/alg/rollout_buffer.py:60-80``) in f32 numpy, operation for operation.
"""
import numpy as np
import torch

from . import philox
from .packing import pack_boards, record_rows, record_words

REC_ACTION_MASK = 0xFFFF
REC_REWARD_SHIFT = 16  # i8 in bits 16..23
REC_DONE_BIT = 24
REC_SIDE_BIT = 25


def encode_record(action, reward, done, side) -> np.ndarray:
    a = np.asarray(action, dtype=np.int64) & REC_ACTION_MASK
    r = (np.asarray(reward).astype(np.int64) & 0xFF) << REC_REWARD_SHIFT
    d = np.asarray(done).astype(np.int64) << REC_DONE_BIT
    s = np.asarray(side).astype(np.int64) << REC_SIDE_BIT
    return (a | r | d | s).astype(np.uint32)


def random_rollout(env, seed: int, step0: int, steps: int, env_id0: int = 0):
    """Run ``steps`` random plies on every env of ``env`` (an OracleVectorEnv).

    Returns (rec_planes u64[T,R,N], rec_meta u32[T,N], stats i64[5]) where the
    planes are the boards *before* the ply as record rows, mover's plane first (``packing.record_rows``), and stats =
    [episodes finished, black wins, white wins, draws, sum of finished-episode lengths].
    """
    m, n = env.m, env.n
    nenv = env.num_envs
    ids = np.arange(env_id0, env_id0 + nenv, dtype=np.uint64)
    rec_planes, rec_meta = [], []
    stats = np.zeros(5, dtype=np.int64)
    for t in range(steps):
        before = env.observe()
        side = env.current_player.numpy().copy()
        rec_planes.append(record_rows(pack_boards(before["observation"].numpy(), m, n), m, n, side))
        x = philox.rand_u32(seed, ids, step0 + t, philox.STREAM_MOVE)
        act = philox.pick_legal(before["action_mask"].numpy(), x)
        _, rew, done = env.step(torch.from_numpy(act))
        rew_np = rew.numpy()
        done_np = done.numpy()
        rec_meta.append(encode_record(act, rew_np.astype(np.int64), done_np, side))
        if done_np.any():
            lengths = env.move_counts.numpy()[done_np]
            won = rew_np[done_np] == 1.0
            stats[0] += int(done_np.sum())
            stats[1] += int((won & (side[done_np] == 0)).sum())
            stats[2] += int((won & (side[done_np] == 1)).sum())
            stats[3] += int((~won).sum())
            stats[4] += int(lengths.sum())
            env.reset(torch.from_numpy(np.nonzero(done_np)[0]))
    planes = np.stack(rec_planes) if rec_planes else np.zeros((0, record_words(m, n), nenv), np.uint64)
    meta = np.stack(rec_meta) if rec_meta else np.zeros((0, nenv), np.uint32)
    return planes, meta, stats


def replay_actions(env, actions):
    """Replays an action log on ``env`` (an OracleVectorEnv in its chunk-start state): the records the
    fused rollout would have written for those actions -- what ``mnk_replay_actions`` must rebuild.
    actions: int array [T, N].  Returns (rec_planes u64[T,R,N], rec_meta u32[T,N])."""
    m, n = env.m, env.n
    rec_planes, rec_meta = [], []
    for act in np.asarray(actions, dtype=np.int64):
        side = env.current_player.numpy().copy()
        rec_planes.append(record_rows(pack_boards(env.boards.numpy(), m, n), m, n, side))
        _, rew, done = env.step(torch.from_numpy(act))
        done_np = done.numpy()
        rec_meta.append(encode_record(act, rew.numpy().astype(np.int64), done_np, side))
        if done_np.any():
            env.reset(torch.from_numpy(np.nonzero(done_np)[0]))
    return np.stack(rec_planes), np.stack(rec_meta)


def gae(rewards, values, dones, last_values, gamma=0.99, lam=0.95):
    """f32 restatement of rollout_buffer.py:60-80.  Inputs are [T, N] (+ [N]); returns (adv, ret)."""
    rewards = np.asarray(rewards, dtype=np.float32)
    values = np.asarray(values, dtype=np.float32)
    nonterm = np.float32(1.0) - np.asarray(dones).astype(np.float32)
    steps = rewards.shape[0]
    adv = np.zeros_like(rewards)
    run = np.zeros(rewards.shape[1], dtype=np.float32)
    g = np.float32(gamma)
    gl = np.float32(gamma * lam)
    for t in reversed(range(steps)):
        nxt = np.asarray(last_values, dtype=np.float32).reshape(-1) if t == steps - 1 else values[t + 1]
        delta = rewards[t] + g * nxt * nonterm[t] - values[t]
        run = delta + gl * nonterm[t] * run
        adv[t] = run
    return adv, adv + values


# ---- action-log formats of the multi-GPU exchange (include/mnk_hip.h MNK_ACT_*), restated in numpy
ACT_U8, ACT_U16, ACT_BITS7, ACT_U8P1 = 1, 2, 3, 4


def encode_action_log(actions, fmt: int) -> np.ndarray:
    """actions int [T, N] -> the packed log ``mnk_rollout_random`` writes for them:
    ACT_U8    uint32 [ceil(T/4), N]   action of ply 4q+j in byte j of word q
    ACT_U16   uint64 [ceil(T/4), N]   ... in 16-bit field j
    ACT_BITS7 uint32 [ceil(7 ceil(T/4) / 8), N]   a bit stream per env, action of ply p at bits [7p, 7p+7)
    ACT_U8P1  uint32 [ceil(T/4) + ceil(T/32), N]  the ACT_U8 words of the low bytes, then bit 8 of ply p at bit p % 32
                                                  of word ceil(T/4) + p // 32
    Plies past T count as action 0."""
    a = np.asarray(actions, dtype=np.uint64)
    t, n = a.shape
    q = (t + 3) // 4
    if fmt == ACT_U8P1:
        assert (a < 512).all()
        high = np.zeros(((t + 31) // 32, n), dtype=np.uint32)
        for p in range(t):
            high[p // 32] |= ((a[p] >> np.uint64(8)) & np.uint64(1)).astype(np.uint32) << np.uint32(p % 32)
        return np.concatenate([encode_action_log(a & np.uint64(0xFF), ACT_U8), high])
    a = np.concatenate([a, np.zeros((4 * q - t, n), dtype=np.uint64)])
    if fmt in (ACT_U8, ACT_U16):
        bits = 8 if fmt == ACT_U8 else 16
        quads = a.reshape(q, 4, n)
        word = quads[:, 0] | quads[:, 1] << np.uint64(bits) | quads[:, 2] << np.uint64(2 * bits) | quads[:, 3] << np.uint64(3 * bits)
        return word.astype(np.uint32) if fmt == ACT_U8 else word
    assert fmt == ACT_BITS7 and (a < 128).all()
    words = (7 * q + 7) // 8
    out = np.zeros((words + 1, n), dtype=np.uint64)
    for p in range(4 * q):
        w, sh = divmod(7 * p, 32)
        v = a[p] << np.uint64(sh)
        out[w] |= v & np.uint64(0xFFFFFFFF)
        out[w + 1] |= v >> np.uint64(32)
    assert not out[words].any()
    return out[:words].astype(np.uint32)


def decode_action_log(log, steps: int, fmt: int) -> np.ndarray:
    """inverse of ``encode_action_log``: int64 actions [steps, N]"""
    log = np.asarray(log)
    if fmt == ACT_U8P1:
        q = (steps + 3) // 4
        low = decode_action_log(log[:q], steps, ACT_U8)
        high = np.stack([(log[q + p // 32].astype(np.int64) >> (p % 32)) & 1 for p in range(steps)]) if steps else low
        return low | (high << 8)
    if fmt in (ACT_U8, ACT_U16):
        bits = 8 if fmt == ACT_U8 else 16
        w = log.astype(np.uint64)
        fields = [(w >> np.uint64(bits * j)) & np.uint64((1 << bits) - 1) for j in range(4)]
        return np.stack(fields, axis=1).reshape(-1, log.shape[1])[:steps].astype(np.int64)
    w = np.concatenate([log.astype(np.uint64), np.zeros((1, log.shape[1]), dtype=np.uint64)])
    out = np.zeros((steps, log.shape[1]), dtype=np.int64)
    for p in range(steps):
        i, sh = divmod(7 * p, 32)
        out[p] = (((w[i] | (w[i + 1] << np.uint64(32))) >> np.uint64(sh)) & np.uint64(0x7F)).astype(np.int64)
    return out

"""Fused random-policy rollout: the BASELINE.json workload as one kernel launch per chunk.

The reference has no rollout driver for random play; the loop it runs is the one in
SURVEY.md Appendix A -- ``RandomPolicy.act`` (``src/selfplay/policy.py:18-29``) ->
``env.step`` (``src/env/torch_vector_mnk_env.py:55-84``) -> ``env.reset(nonzero(done))``
(``:34-44``) -- about 45 eager launches and two host syncs per ply.  Here T plies of
every env run inside ``mnk_rollout_random`` with the state in registers; what reaches
HBM is the packed record of each ply:

    planes  u64[T][R][N]      the board *before* the ply, in the mover's view: row w = mover's word w |
                              other side's word w << 32 (32-bit words of the two planes; R = ceil(m*(n+1)/32),
                              no padding; the mover's colour is the side bit of the same ply's meta word)
    meta    u32[T][N]         action | reward << 16 | done << 24 | mover side << 25

Envs are independent, so a node shards them by contiguous blocks: rank r owns global
env ids [r*N, (r+1)*N) and the Philox key uses the global id, which makes the records
independent of the number of GPUs.  There is one exchange step, an all-gather over the
process group (RCCL over xGMI when the backend is ``nccl``), in one of two formats:

  * ``gather_records``      the packed records themselves (28 B per env-step at 9x9);
  * ``gather_action_logs``  the action log (0.875 B per env-step on boards of up to 128 cells -- 7 bits per
                            action --, 1 B up to 256 cells, 2 B beyond), with or without the chunk-start state:
                            a rollout is a pure function of state + log, and ``replay_shard`` rebuilds any
                            shard's full records bit-identically on the receiving GPU.  With the state in the
                            message (``with_state=True``, +36 B per env and chunk at 9x9) any chunk can be replayed
                            on its own; without it the receiver keeps every shard's replay state itself
                            (``gather_start_state`` once, then every chunk replayed in order) and the log alone
                            crosses the links.
"""
from dataclasses import dataclass
from typing import Optional

import torch

import mnk_hip


@dataclass
class RolloutRecords:
    planes: torch.Tensor  # int64 (u64 bits) [T, R, N], R = mnk_hip.record_words(m, n)
    meta: torch.Tensor    # int32 (u32 bits) [T, N]
    # only when the action log is on (alloc(..., log_actions=True)); all three are views into `msg`
    act: Optional[torch.Tensor] = None      # action log: int32 [words, N] (ACT_U8: 4 plies per word; ACT_BITS7: a 7-bit
                                            # stream) or int64 [ceil(T/4), N] (ACT_U16)
    meta0: Optional[torch.Tensor] = None    # env meta words at the start of the chunk, int32 [N]    (with_state only)
    planes0: Optional[torch.Tensor] = None  # env planes at the start of the chunk, int64 [2, W, N]  (with_state only)
    msg: Optional[torch.Tensor] = None      # the one flat int64 buffer that is all-gathered
    fmt: int = 0                            # ACT_U8 / ACT_U16 / ACT_BITS7 (mnk_hip.h MNK_ACT_*); 0 = no log

    @property
    def steps(self) -> int:
        return self.meta.shape[0]

    def actions(self) -> torch.Tensor:
        return (self.meta & mnk_hip.REC_ACTION_MASK).to(torch.int64)

    def rewards(self) -> torch.Tensor:
        return ((self.meta >> mnk_hip.REC_REWARD_SHIFT) & 0xFF).to(torch.int8).to(torch.float32)

    def dones(self) -> torch.Tensor:
        return ((self.meta >> mnk_hip.REC_DONE_BIT) & 1).to(torch.bool)

    def sides(self) -> torch.Tensor:
        return ((self.meta >> mnk_hip.REC_SIDE_BIT) & 1).to(torch.int64)


class RandomRollout:
    """Drives ``mnk_rollout_random`` on one env shard.

    ``env``: a ``TorchVectorMnkEnv``; ``seed``: Philox key; ``env_id0``: global id of
    this shard's first env (rank * num_envs when sharded).
    """

    def __init__(self, env, seed: int = 0, env_id0: int = 0):
        self.env = env
        self.seed = int(seed)
        self.env_id0 = int(env_id0)
        self.step = 0  # plies played per env so far = Philox step counter
        # replicated device counters (see MNK_STATS_REPLICAS in include/mnk_hip.h)
        self._stats = torch.zeros((mnk_hip.STATS_REPLICAS, mnk_hip.STATS_STRIDE), dtype=torch.int64, device=env._dev)

    @property
    def stats(self) -> torch.Tensor:
        """int64[5]: episodes finished, black wins, white wins, draws, sum of finished-episode lengths"""
        return self._stats.sum(dim=0)[:mnk_hip.STATS_COUNTERS]

    def state_dict(self) -> dict:
        """env state + Philox key, step counter and statistics: a driver restored from it continues bit-exactly"""
        return {"env": self.env.state_dict(), "seed": self.seed, "env_id0": self.env_id0, "step": self.step,
                "stats": self._stats.cpu()}

    def load_state_dict(self, state: dict) -> None:
        self.env.load_state_dict(state["env"])
        self.seed, self.env_id0, self.step = int(state["seed"]), int(state["env_id0"]), int(state["step"])
        self._stats.copy_(state["stats"])

    def alloc(self, steps: int, log_actions=False, with_state: bool = True) -> RolloutRecords:
        """Record buffers for ``steps`` plies.  ``log_actions``: False, True (the most compact format the board allows:
        7 bits per action up to 128 cells, a byte and a bit above 256) or one of ACT_U8 / ACT_U16 / ACT_BITS7 /
        ACT_U8P1; ``with_state``: the chunk-start
        planes / meta travel in the message too (a self-contained message), else the log alone."""
        env = self.env
        rec = RolloutRecords(
            planes=torch.empty((steps, mnk_hip.record_words(env.m, env.n), env.num_envs), dtype=torch.int64,
                               device=env._dev),
            meta=torch.empty((steps, env.num_envs), dtype=torch.int32, device=env._dev),
        )
        if log_actions:
            fmt = action_log_format(env.max_moves) if log_actions is True else int(log_actions)
            if not action_log_fits(fmt, env.max_moves):
                raise ValueError(f"action-log format {fmt} does not fit a board of {env.max_moves} cells")
            rec.fmt = fmt
            rec.msg = torch.zeros(_msg_words(env.words, env.num_envs, steps, fmt, with_state), dtype=torch.int64,
                                  device=env._dev)
            rec.planes0, rec.act, rec.meta0 = _msg_views(rec.msg, env.words, env.num_envs, steps, fmt, with_state)
        return rec

    def run(self, steps: int, out: Optional[RolloutRecords] = None, record: bool = True) -> Optional[RolloutRecords]:
        """Plays ``steps`` random plies on every env (finished games restart in place).
        One launch; returns the records (written into ``out`` when given)."""
        env = self.env
        if record and out is None:
            out = self.alloc(steps)
        act = None
        if record:
            assert out.meta.shape == (steps, env.num_envs) and out.planes.shape[0] == steps
            act = out.act
            if out.meta0 is not None:  # the chunk-start state a replay needs travels with the log
                out.meta0.copy_(env._meta)
                out.planes0.copy_(env._planes)
        if env.num_envs and steps:
            mnk_hip.call("mnk_rollout_random", mnk_hip.ptr(env._planes), mnk_hip.ptr(env._meta), env.num_envs,
                         env.m, env.n, env.k, steps, self.seed, self.step, self.env_id0,
                         mnk_hip.ptr(out.planes) if record else None, mnk_hip.ptr(out.meta) if record else None,
                         mnk_hip.ptr(self._stats), mnk_hip.ptr(act), out.fmt if act is not None else 0, env._stream())
        self.step += steps
        return out if record else None


ACT_U8, ACT_U16, ACT_BITS7, ACT_U8P1 = 1, 2, 3, 4  # include/mnk_hip.h MNK_ACT_*


def action_log_format(num_actions: int, compact: bool = True) -> int:
    """The most compact log format a board of ``num_actions`` cells allows: 7 bits per action up to 128 cells, one byte
    up to 256, a byte and a bit up to 512, two bytes beyond (without ``compact``: one byte up to 256 cells, two bytes
    beyond -- round 2's)."""
    if compact and num_actions <= 128:
        return ACT_BITS7
    if num_actions <= 256:
        return ACT_U8
    return ACT_U8P1 if compact and num_actions <= 512 else ACT_U16


def action_log_fits(fmt: int, num_actions: int) -> bool:
    return fmt == ACT_U16 or (fmt == ACT_U8 and num_actions <= 256) or (fmt == ACT_BITS7 and num_actions <= 128) or \
        (fmt == ACT_U8P1 and 256 < num_actions <= 512)


def action_log_words(fmt: int, steps: int) -> int:
    """32-bit words per env of a ``steps``-ply log (= mnk_action_log_words)"""
    q = (steps + 3) // 4
    return {ACT_U8: q, ACT_U16: 2 * q, ACT_BITS7: (7 * q + 7) // 8, ACT_U8P1: q + (steps + 31) // 32}[fmt]


def unpack_action_log(act: torch.Tensor, steps: int, fmt: Optional[int] = None) -> torch.Tensor:
    """packed log -> int64 actions [T, N] (``fmt`` defaults to ACT_U8 for an int32 log, ACT_U16 for an int64 one)"""
    if fmt is None:
        fmt = ACT_U8 if act.dtype == torch.int32 else ACT_U16
    if fmt == ACT_BITS7:
        words = act.to(torch.int64) & 0xFFFFFFFF  # [W, N] u32 values
        t = torch.arange(steps, device=act.device)
        bit = 7 * t
        w, sh = bit // 32, bit % 32
        lo = words[w]
        hi = words[torch.clamp(w + 1, max=words.shape[0] - 1)]
        return (((lo | (hi << 32)) >> sh.unsqueeze(1)) & 0x7F).to(torch.int64)
    if fmt == ACT_U8P1:
        q = (steps + 3) // 4
        low = unpack_action_log(act[:q], steps, ACT_U8)
        t = torch.arange(steps, device=act.device)
        high = (act[q:].to(torch.int64)[t // 32] >> (t % 32).unsqueeze(1)) & 1
        return low | (high << 8)
    bits = 8 if fmt == ACT_U8 else 16
    fields = [(act >> (bits * j)) & ((1 << bits) - 1) for j in range(4)]
    return torch.stack(fields, dim=1).reshape(-1, act.shape[1])[:steps].to(torch.int64)


def _msg_layout(words: int, nenv: int, steps: int, fmt: int, with_state: bool):
    """Sizes (in int64 words) of the three parts of the exchange message: planes0 | act | meta0 (the first and the
    last only ``with_state``)."""
    act_bytes = action_log_words(fmt, steps) * nenv * 4
    n_planes = 2 * words * nenv if with_state else 0
    n_act = (act_bytes + 7) // 8
    n_meta = (nenv * 4 + 7) // 8 if with_state else 0
    return n_planes, n_act, n_meta


def _msg_words(words, nenv, steps, fmt, with_state=True) -> int:
    return sum(_msg_layout(words, nenv, steps, fmt, with_state))


def _msg_views(msg, words, nenv, steps, fmt, with_state=True):
    """(planes0 [.., 2, W, N] int64, act [.., words, N] int32 / [.., ceil(T/4), N] int64, meta0 [.., N] int32) views of
    a message buffer whose last dimension is the flat message (leading dimensions, e.g. the rank axis, are kept);
    planes0 / meta0 are None for a log-only message."""
    n_planes, n_act, n_meta = _msg_layout(words, nenv, steps, fmt, with_state)
    lead = msg.shape[:-1]
    q = (steps + 3) // 4
    planes0 = msg[..., :n_planes].reshape(lead + (2, words, nenv)) if with_state else None
    act64 = msg[..., n_planes:n_planes + n_act]
    if fmt == ACT_U16:
        act = act64[..., :q * nenv].reshape(lead + (q, nenv))
    else:
        aw = action_log_words(fmt, steps)
        act = act64.view(torch.int32)[..., :aw * nenv].reshape(lead + (aw, nenv))
    meta0 = msg[..., n_planes + n_act:].view(torch.int32)[..., :nenv] if with_state else None
    return planes0, act, meta0


@dataclass
class GatheredLogs:
    """What ``gather_action_logs`` leaves on every rank: per shard r the log (and, in a self-contained message, the
    chunk-start state), as views of the gathered message buffer ``msg`` [world, L]."""
    planes0: Optional[torch.Tensor]  # int64 [world, 2, W, N]; None for log-only messages
    meta0: Optional[torch.Tensor]    # int32 [world, N]; None for log-only messages
    act: torch.Tensor                # int32 [world, words, N] / int64 [world, ceil(T/4), N]
    steps: int = 0                   # T
    msg: Optional[torch.Tensor] = None
    fmt: int = ACT_U8

    @staticmethod
    def empty(world: int, words: int, nenv: int, steps: int, num_actions: int, device, fmt: Optional[int] = None,
              with_state: bool = True) -> "GatheredLogs":
        """``fmt`` defaults to the byte / two-byte format of ``num_actions`` (the round-2 message)"""
        if fmt is None:
            fmt = action_log_format(num_actions, compact=False)
        msg = torch.empty((world, _msg_words(words, nenv, steps, fmt, with_state)), dtype=torch.int64, device=device)
        planes0, act, meta0 = _msg_views(msg, words, nenv, steps, fmt, with_state)
        return GatheredLogs(planes0=planes0, meta0=meta0, act=act, steps=steps, msg=msg, fmt=fmt)


@dataclass
class ReplayState:
    """Every shard's env state on this rank, advanced by ``replay_shard`` chunk after chunk: what lets the ranks
    exchange the action log alone.  ``gather_start_state`` fills it once, before the first chunk."""
    planes: torch.Tensor  # int64 [world, 2, W, N]
    meta: torch.Tensor    # int32 [world, N]
    msg: torch.Tensor     # the gathered buffer the two are views of


def gather_start_state(env, group=None, exchange=None, stream=None) -> ReplayState:
    """All-gather of every shard's current env state (planes | meta: 36 B per env at 9x9), once: afterwards a rank
    that replays every chunk of a shard in order (``replay_shard(..., state=...)``) always holds that shard's
    chunk-start state, and the per-chunk message is the log alone."""
    n, w = env.num_envs, env.words
    n_planes, n_meta = 2 * w * n, (n * 4 + 7) // 8
    send = torch.zeros(n_planes + n_meta, dtype=torch.int64, device=env._dev)
    send[:n_planes].copy_(env._planes.reshape(-1))
    send[n_planes:].view(torch.int32)[:n].copy_(env._meta)
    if exchange is not None:
        world = exchange.world
    else:
        import torch.distributed as dist

        world = dist.get_world_size(group) if dist.is_initialized() else 1
    recv = torch.empty((world, send.numel()), dtype=torch.int64, device=env._dev)
    if exchange is not None:
        exchange.all_gather(send, recv.view(-1), stream)
    elif world > 1:
        dist.all_gather_into_tensor(recv.view(-1), send, group=group)
    else:
        recv[0].copy_(send)
    return ReplayState(planes=recv[:, :n_planes].reshape(world, 2, w, n),
                       meta=recv[:, n_planes:].view(torch.int32)[:, :n], msg=recv)


def gather_action_logs(rec: RolloutRecords, group=None, out: Optional[GatheredLogs] = None, exchange=None,
                       stream=None) -> GatheredLogs:
    """All-gather of (chunk-start state, action log) as ONE message per rank: 1-2 bytes per env-step on the
    wire plus the 36 B/env state once per chunk.

    ``exchange``: a ``selfplay.exchange.RecordExchange`` -- the collective is then ``mnk_allgather_records`` of
    the C ABI (RCCL over xGMI) enqueued on ``stream`` (default: the current stream).  Without it the
    ``torch.distributed`` group is used (``gloo`` in the CPU tests, ``nccl`` = the same RCCL otherwise)."""
    assert rec.msg is not None, "run the rollout with alloc(..., log_actions=True)"
    if exchange is not None:
        world = exchange.world
    else:
        import torch.distributed as dist

        world = dist.get_world_size(group)
    t, n = rec.meta.shape
    with_state = rec.planes0 is not None
    w = rec.planes0.shape[1] if with_state else 0  # the chunk-start state travels in the state layout [2, W, N]
    if out is None:
        out = GatheredLogs.empty(world, w, n, t, 0, rec.msg.device, fmt=rec.fmt, with_state=with_state)
    assert out.msg.shape == (world, rec.msg.numel()) and out.steps == t and out.fmt == rec.fmt
    if exchange is not None:
        exchange.all_gather(rec.msg, out.msg.view(-1), stream)
    else:
        dist.all_gather_into_tensor(out.msg.view(-1), rec.msg, group=group)
    return out


class LogGroup:
    """J consecutive chunks' exchange messages laid end to end in ONE flat buffer, so that J chunks cost one
    all-gather of J times the bytes instead of J all-gathers (``bench.py --exchange-every J``: RCCL's per-call cost and
    the small-message regime of xGMI are what a larger message amortises).  Chunk j's message keeps the layout of
    ``_msg_views`` (chunk-start state | log | meta) and is self-contained or log-only according to ``with_state[j]``
    -- a keyframed stream puts a state into the chunks the keyframe cadence names, wherever they fall in the group.

    Sender: ``records(j)`` is the ``RolloutRecords`` chunk j's launch writes (its ``msg`` is a view into ``send``; the
    record rows ``planes`` / ``meta`` are shared by the chunks of the group unless ``own_records``).  Receiver:
    ``gather()`` all-gathers ``send`` into ``recv`` [world, L]; ``logs(j)`` are chunk j's ``GatheredLogs`` (views of
    ``recv``), ready for ``replay_shard`` / ``KeyframedLogs.push``."""

    def __init__(self, world: int, words: int, nenv: int, steps: int, fmt: int, with_state, device, planes=None,
                 meta=None, rows: int = 0):
        self.world, self.words, self.nenv, self.steps, self.fmt = world, words, nenv, steps, fmt
        self.with_state = tuple(bool(s) for s in with_state)
        sizes = [_msg_words(words, nenv, steps, fmt, s) for s in self.with_state]
        self.offsets = [sum(sizes[:j]) for j in range(len(sizes) + 1)]
        self.send = torch.zeros(self.offsets[-1], dtype=torch.int64, device=device)
        self.recv = torch.empty((world, self.offsets[-1]), dtype=torch.int64, device=device)
        self._recs, self._logs = [], []
        for j, state in enumerate(self.with_state):
            lo, hi = self.offsets[j], self.offsets[j + 1]
            msg = self.send[lo:hi]
            planes0, act, meta0 = _msg_views(msg, words, nenv, steps, fmt, state)
            rec = RolloutRecords(planes=planes, meta=meta, act=act, meta0=meta0, planes0=planes0, msg=msg, fmt=fmt)
            if planes is None and rows:  # records of its own (tests; the bench shares one set per slot)
                rec.planes = torch.empty((steps, rows, nenv), dtype=torch.int64, device=device)
                rec.meta = torch.empty((steps, nenv), dtype=torch.int32, device=device)
            self._recs.append(rec)
            got = self.recv[:, lo:hi]  # [world, L_j], row stride = the whole group's length
            gp0, gact, gm0 = _msg_views(got, words, nenv, steps, fmt, state)
            # the views must alias `recv` (a silent copy would be read before the all-gather has filled it)
            assert gact.data_ptr() == got.data_ptr() + (_msg_layout(words, nenv, steps, fmt, state)[0]) * 8
            self._logs.append(GatheredLogs(planes0=gp0, meta0=gm0, act=gact, steps=steps, msg=got, fmt=fmt))

    def __len__(self) -> int:
        return len(self.with_state)

    @property
    def bytes(self) -> int:
        """bytes one rank sends per exchange"""
        return self.send.numel() * 8

    def records(self, j: int) -> RolloutRecords:
        return self._recs[j]

    def logs(self, j: int) -> GatheredLogs:
        return self._logs[j]

    def gather(self, group=None, exchange=None, stream=None) -> "LogGroup":
        """ONE all-gather of the J messages (``exchange``: the C ABI's RCCL communicator, enqueued on ``stream``;
        else ``torch.distributed``)."""
        if exchange is not None:
            assert exchange.world == self.world
            exchange.all_gather(self.send, self.recv.view(-1), stream)
        else:
            import torch.distributed as dist

            if dist.is_initialized() and dist.get_world_size(group) > 1:
                assert dist.get_world_size(group) == self.world
                dist.all_gather_into_tensor(self.recv.view(-1), self.send, group=group)
            else:
                assert self.world == 1
                self.recv[0].copy_(self.send)
        return self


def replay_shard(logs: GatheredLogs, shard: int, m: int, n: int, k: int, err: Optional[torch.Tensor] = None,
                 out: Optional[RolloutRecords] = None, scratch=None, state: Optional[ReplayState] = None,
                 record: bool = True) -> Optional[RolloutRecords]:
    """Rebuilds shard ``shard``'s full packed records from its chunk-start state + action log
    (``mnk_replay_actions``, one launch); bit-identical to what the owning rank recorded.
    The state comes from the message itself (self-contained messages: a copy is advanced, ``scratch`` = optional
    ``(planes int64 [2, W, N], meta int32 [N])`` buffers for it) or from ``state`` -- a ``ReplayState`` (its entry for
    ``shard``) or a ``(planes [2, W, N], meta [N])`` pair -- which is advanced IN PLACE: for log-only messages, where
    every chunk must be replayed in order exactly once (``record=False`` only advances the state)."""
    act = logs.act[shard]
    t, nenv = logs.steps, act.shape[1]
    assert act.shape[0] == (action_log_words(logs.fmt, t) if logs.fmt != ACT_U16 else (t + 3) // 4), \
        "GatheredLogs.steps does not match the packed log"
    dev = act.device
    if isinstance(state, ReplayState):
        planes, meta = state.planes[shard], state.meta[shard]
    elif state is not None:
        planes, meta = state
    elif logs.planes0 is None:
        raise ValueError("log-only messages need the receiver's ReplayState (gather_start_state)")
    elif scratch is not None:
        planes, meta = scratch
        planes.copy_(logs.planes0[shard])
        meta.copy_(logs.meta0[shard])
    else:
        planes = logs.planes0[shard].clone()
        meta = logs.meta0[shard].clone()
    if not record:
        out = None
    elif out is None:
        out = RolloutRecords(planes=torch.empty((t, mnk_hip.record_words(m, n), nenv), dtype=torch.int64, device=dev),
                             meta=torch.empty((t, nenv), dtype=torch.int32, device=dev))
    if err is None:
        err = torch.zeros(2, dtype=torch.int32, device=dev)
    if t and nenv:
        mnk_hip.call("mnk_replay_actions", mnk_hip.ptr(planes), mnk_hip.ptr(meta), nenv, m, n, k, t,
                     mnk_hip.ptr(act), logs.fmt, mnk_hip.ptr(out.planes if out is not None else None),
                     mnk_hip.ptr(out.meta if out is not None else None), mnk_hip.ptr(err), mnk_hip.stream_ptr(dev))
    return out


class KeyframedLogs:
    """The receiving side of a KEYFRAMED log stream: every K-th chunk's message carries the chunk-start state (a
    keyframe), the chunks in between carry the log alone.  The wire then costs 0.875 + 36 / (K * T) bytes per env-step
    at 9x9 (0.893 at K = 8, T = 256) and nobody has to keep replaying: the records of any chunk since the last keyframe
    are rebuilt ON DEMAND -- state-only replays of the chunks before it (no record stores: ~60 us per 256 plies x 65 536
    envs), then one recording replay.  ``push`` COPIES what it is given (the gather buffers are reused)."""

    def __init__(self, m: int, n: int, k: int):
        self.geom = (m, n, k)
        self.key: Optional[GatheredLogs] = None
        self.tail = []  # log-only GatheredLogs since the keyframe

    @staticmethod
    def _copy(logs: GatheredLogs) -> GatheredLogs:
        msg = logs.msg.clone()
        world = msg.shape[0]
        with_state = logs.planes0 is not None
        w = logs.planes0.shape[2] if with_state else 0
        nenv = logs.act.shape[-1]
        planes0, act, meta0 = _msg_views(msg, w, nenv, logs.steps, logs.fmt, with_state)
        assert act.shape == logs.act.shape and world == logs.act.shape[0]
        return GatheredLogs(planes0=planes0, meta0=meta0, act=act, steps=logs.steps, msg=msg, fmt=logs.fmt)

    def push(self, logs: GatheredLogs) -> None:
        if logs.planes0 is not None:
            self.key, self.tail = self._copy(logs), []
        else:
            if self.key is None:
                raise ValueError("a log-only message before the first keyframe")
            self.tail.append(self._copy(logs))

    def chunks(self) -> int:
        """chunks held: the keyframe chunk plus the log-only ones after it"""
        return 0 if self.key is None else 1 + len(self.tail)

    def rebuild(self, shard: int, j: int, err: Optional[torch.Tensor] = None) -> RolloutRecords:
        """Records of chunk ``j`` since the keyframe (0 = the keyframe's own chunk) of shard ``shard``."""
        if not 0 <= j < self.chunks():
            raise IndexError(f"chunk {j} of {self.chunks()} held since the last keyframe")
        m, n, k = self.geom
        state = (self.key.planes0[shard].clone(), self.key.meta0[shard].clone())
        seq = [self.key] + self.tail
        for i in range(j):
            replay_shard(seq[i], shard, m, n, k, err=err, state=state, record=False)
        return replay_shard(seq[j], shard, m, n, k, err=err, state=state)


def gather_records(rec: RolloutRecords, group=None) -> RolloutRecords:
    """All-gather of the packed records over the env axis: every rank ends up with
    [T, R, world*N] / [T, world*N], rank r's envs at columns [r*N, (r+1)*N).

    Uses ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on the GPUs, ``gloo`` in
    the CPU tests).  The collective gathers rank-major buffers; the permute back to the
    env-minor layout is a local copy.
    """
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1:
        return rec
    t, rows, n = rec.planes.shape
    planes_all = torch.empty((world, t, rows, n), dtype=rec.planes.dtype, device=rec.planes.device)
    meta_all = torch.empty((world, t, n), dtype=rec.meta.dtype, device=rec.meta.device)
    # flat views: the concatenated form every backend accepts (gloo rejects the stacked shape)
    dist.all_gather_into_tensor(planes_all.view(-1), rec.planes.contiguous().view(-1), group=group)
    dist.all_gather_into_tensor(meta_all.view(-1), rec.meta.contiguous().view(-1), group=group)
    planes = planes_all.permute(1, 2, 0, 3).reshape(t, rows, world * n)
    meta = meta_all.permute(1, 0, 2).reshape(t, world * n)
    return RolloutRecords(planes=planes.contiguous(), meta=meta.contiguous())


def unpack_records(rec: RolloutRecords, env, obs_dtype=torch.float32) -> dict:
    """Packed records -> the field layout of the reference's ``RolloutBuffer``
    (``alg/rollout_buffer.py:14-44``): observations f32 [T,N,2,m,n] from the mover's point of view,
    action_masks bool [T,N,C], actions i64 [T,N], rewards f32 [T,N], dones bool [T,N].  One launch."""
    t, n = rec.meta.shape
    dev = rec.meta.device
    out = {
        "observations": torch.empty((t, n, 2, env.m, env.n), dtype=obs_dtype, device=dev),
        "action_masks": torch.empty((t, n, env.max_moves), dtype=torch.bool, device=dev),
        "actions": torch.empty((t, n), dtype=torch.long, device=dev),
        "rewards": torch.empty((t, n), dtype=torch.float32, device=dev),
        "dones": torch.empty((t, n), dtype=torch.bool, device=dev),
    }
    if t and n:
        mnk_hip.call("mnk_unpack_records", mnk_hip.ptr(rec.planes), mnk_hip.ptr(rec.meta), n, t, env.m, env.n,
                     mnk_hip.ptr(out["observations"]), mnk_hip.obs_code(out["observations"]),
                     mnk_hip.ptr(out["action_masks"]), mnk_hip.ptr(out["actions"]),
                     mnk_hip.ptr(out["rewards"]), mnk_hip.ptr(out["dones"]), mnk_hip.stream_ptr(dev))
    return out


def gae(rewards, values, dones, last_values, gamma: float = 0.99, gae_lambda: float = 0.95):
    """Advantages and returns of ``RolloutBuffer.compute_advantages_and_returns``
    (``alg/rollout_buffer.py:60-80``) in one launch; inputs [T,N] (+ [N]), f32 / bool on the GPU."""
    t, n = rewards.shape
    dev = rewards.device
    adv = torch.empty((t, n), dtype=torch.float32, device=dev)
    ret = torch.empty((t, n), dtype=torch.float32, device=dev)
    # the contiguous copies are bound to names: a temporary would be freed (and its block reused by the next
    # temporary) before the launch has read it
    r = rewards.to(torch.float32).contiguous()
    v = values.to(torch.float32).contiguous()
    d = dones.to(torch.bool).contiguous()
    last = last_values.to(torch.float32).reshape(-1).contiguous()
    assert v.shape == (t, n) and d.shape == (t, n) and last.numel() == n
    if t and n:
        mnk_hip.call("mnk_gae", mnk_hip.ptr(r), mnk_hip.ptr(v), mnk_hip.ptr(d), mnk_hip.ptr(last), n, t,
                     float(gamma), float(gamma * gae_lambda), mnk_hip.ptr(adv), mnk_hip.ptr(ret),
                     mnk_hip.stream_ptr(dev))
    return adv, ret

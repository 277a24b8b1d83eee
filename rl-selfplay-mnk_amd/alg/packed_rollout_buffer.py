"""Rollout buffer that keeps observations bit-packed.

Same role and the same minibatch tuple as the reference's ``RolloutBuffer``
(``/root/reference/src/alg/rollout_buffer.py:4-113``), but a stored agent-step is the packed canonical
planes (16*W B: 32 B at 9x9) instead of the f32 observation and the bool mask (8C + C B: 729 B at 9x9) --
the mask is a function of the planes.  65 536 envs x 256 steps fit in 0.9 GB instead of 12.6 GB.
``get_data_loader`` draws the same shuffled minibatches and expands only the drawn samples, straight into
the network's input layout, with one launch of ``mnk_gather_obs``; GAE is the same ``mnk_gae`` launch as in
the dense drop-in buffer.

The caller adds ``wrapper.packed_obs()`` -- taken when the observation was handed out, i.e. before the
``wrapper.step`` that consumes the action -- instead of the dense observation and mask; everything else
(``add`` order, ``ptr``, "Buffer was full.", ``compute_advantages_and_returns``, ``reset``) is the
reference's.  With ``wrapper.attach_sink(buffer)`` the step kernel itself writes the packed canonical planes into
row t+1 (and rewards / terminated into row t): ``add`` of a row the step wrote copies nothing, and the dense
observation the network needs for the next forward is the only other thing the step writes.  The spill-row rule of the
dense buffer holds here too: the planes that follow the LAST step of a rollout go to the spill row (``row(n_steps)``), so
the first ``add`` of every later rollout is handed the spill row and copies it into row 0 (32 B per env, once per
rollout) -- ``add(buffer.row(0)["packed"], ...)`` would store the zeros ``reset()`` left there.  Pass what the step (or
``wrapper.reset``) returned / ``wrapper.packed_obs()``, as examples/selfplay_ppo.py does.
"""
import contextlib

import torch

import mnk_hip

# what the PPO update reads of a rollout (get_data_loader; reference rollout_buffer.py:82-113) -- returns are
# advantages + values, the very f32 addition the GAE kernel made (rollout_buffer.py:79) -- and everything a buffer holds
UPDATE_FIELDS = ("planes", "actions", "log_probs", "values", "advantages")
ALL_FIELDS = UPDATE_FIELDS + ("returns", "rewards", "dones")


def all_gather_fields(fields, exchange=None, group=None, stream=None, out=None):
    """The exchange step of a sharded rollout buffer on plain tensors: every ``fields[name]`` (``[T, ...]``, contiguous,
    the same shape on every rank) is all-gathered rank-major into ``out[name]`` (``[world * T, ...]``: rank r's steps at
    ``[r * T, (r + 1) * T)``; allocated when ``out`` is None).  ``exchange``: a ``selfplay.exchange.RecordExchange`` --
    the collectives are then ``mnk_allgather_records`` of the C ABI (RCCL over xGMI) on ``stream``; without it the
    ``torch.distributed`` ``group`` carries them (``gloo`` in the CPU tests).  One collective per field, in the order of
    ``fields`` on every rank."""
    if exchange is not None:
        world = exchange.world
    else:
        import torch.distributed as dist

        world = dist.get_world_size(group) if dist.is_initialized() else 1
    out = {} if out is None else out
    for name, send in fields.items():
        if not send.is_contiguous():
            raise ValueError(f"field {name!r} must be contiguous")
        recv = out.get(name)
        if recv is None:
            recv = out[name] = torch.empty((world * send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        if recv.shape[0] != world * send.shape[0] or recv.shape[1:] != send.shape[1:] or recv.dtype != send.dtype or not recv.is_contiguous():
            raise ValueError(f"out[{name!r}] must be a contiguous {send.dtype} tensor of {world} x {tuple(send.shape)}")
        a, b = send.reshape(-1), recv.reshape(-1)
        if a.dtype == torch.bool:  # (byte for byte; not every backend takes bool)
            a, b = a.view(torch.uint8), b.view(torch.uint8)
        if exchange is not None:
            exchange.all_gather(a, b, stream)
            continue
        with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
            if world > 1:
                dist.all_gather_into_tensor(b, a, group=group)
            else:
                b.copy_(a)
    return out


class PackedRolloutBuffer:
    def __init__(self, n_steps, num_envs, m, n, device="cuda", keep_storage=False):
        self.keep_storage = keep_storage  # reset() zeroes in place instead of allocating (a sink / a graph holds row pointers)
        self._live = None                 # row of the plane store that holds the observation handed out last (sink only)
        if torch.device(device).type != "cuda":
            raise RuntimeError("PackedRolloutBuffer needs a GPU device (its gather and GAE are HIP kernels)")
        mnk_hip.load()
        self.n_steps, self.num_envs, self.m, self.n = n_steps, num_envs, m, n
        self.obs_shape, self.action_dim = (2, m, n), m * n
        self.words = mnk_hip.state_words(m, n)
        self.device = device
        self._err = torch.zeros(2, dtype=torch.int32, device=device)
        self.copied_bytes = 0  # bytes ``add`` really copied so far
        self.reset()

    def reset(self):
        t, n, dev = self.n_steps, self.num_envs, self.device
        if getattr(self, "planes", None) is not None and self.keep_storage:
            # keep the storage: rows are bound to the fused step.  The spill row(s) and the row that holds the observation
            # handed out last are left alone -- the next rollout starts from it
            for f in (self.actions, self.log_probs, self.rewards, self.values, self.returns, self.advantages, self.dones):
                f.zero_()
            if self._live is not None and self._live < t:
                self.planes[:self._live].zero_()
                self.planes[self._live + 1:].zero_()
            else:
                self.planes.zero_()
            self.ptr = 0
            return
        carried = None
        if getattr(self, "planes", None) is not None and self._live is not None:
            carried = self._plane_store[self._live].clone()
        spill = 1 if t >= 2 else 2  # a one-step buffer alternates two spill rows (see RolloutBuffer.reset)
        self._plane_store = torch.zeros((t + spill, 2, self.words, n), dtype=torch.int64, device=dev)
        if carried is not None:
            self._live = t
            self._plane_store[t].copy_(carried)
        self.planes = self._plane_store[:t]
        self.actions = torch.zeros((t, n), dtype=torch.long, device=dev)
        self.log_probs = torch.zeros((t, n), dtype=torch.float32, device=dev)
        self.rewards = torch.zeros((t, n), dtype=torch.float32, device=dev)
        self.values = torch.zeros((t, n), dtype=torch.float32, device=dev)
        self.returns = torch.zeros((t, n), dtype=torch.float32, device=dev)
        self.advantages = torch.zeros((t, n), dtype=torch.float32, device=dev)
        self.dones = torch.zeros((t, n), dtype=torch.bool, device=dev)
        self.ptr = 0

    # ------------------------------------------------------------------ the sink of the fused step
    def sink_attached(self, on: bool) -> None:
        if on:
            self.keep_storage = True

    def row(self, t: int) -> dict:
        if not 0 <= t < self._plane_store.shape[0]:
            raise IndexError(f"row {t} of a buffer of {self.n_steps} steps")
        out = {"packed": self._plane_store[t]}
        if t < self.n_steps:
            out.update(actions=self.actions[t], log_probs=self.log_probs[t], rewards=self.rewards[t],
                       values=self.values[t], dones=self.dones[t], terminated=self.dones[t])
        return out

    def reset_outputs(self):
        if self.ptr >= self.n_steps:
            return None
        self._live = self.ptr
        return {"packed": self._plane_store[self.ptr]}

    def step_outputs(self):
        t = self.ptr
        if t >= self.n_steps:
            return None
        nxt = t + 1
        if nxt == self.n_steps and self._live == nxt and self._plane_store.shape[0] > nxt + 1:
            nxt += 1  # a one-step buffer: the other spill row
        self._live = nxt
        return {"packed": self._plane_store[nxt], "rewards": self.rewards[t], "terminated": self.dones[t]}

    def add(self, packed_obs, action, reward, value, log_prob, done):
        if self.ptr >= self.n_steps:
            raise IndexError("Buffer was full.")
        from alg.rollout_buffer import _put

        row = self.ptr
        self.copied_bytes += (_put(self.planes[row], packed_obs) + _put(self.actions[row], action) +
                              _put(self.rewards[row], reward) + _put(self.values[row], value.view(-1)) +
                              _put(self.log_probs[row], log_prob) + _put(self.dones[row], done))
        self.ptr += 1

    def compute_advantages_and_returns(self, last_values, gamma=0.99, gae_lambda=0.95):
        steps = self.ptr
        if steps == 0 or self.num_envs == 0:
            return
        last_values = last_values.reshape(self.num_envs).to(torch.float32).contiguous()
        mnk_hip.call("mnk_gae", mnk_hip.ptr(self.rewards), mnk_hip.ptr(self.values), mnk_hip.ptr(self.dones),
                     mnk_hip.ptr(last_values), self.num_envs, steps, float(gamma), float(gamma * gae_lambda),
                     mnk_hip.ptr(self.advantages), mnk_hip.ptr(self.returns),
                     mnk_hip.stream_ptr(self.rewards.device))

    def gather(self, flat_idx, obs_dtype=torch.float32):
        """(obs [B,2,m,n] float32 / bfloat16 / uint8, mask bool [B,C]) of the samples ``flat_idx`` (= t*N + i), one launch."""
        idx = flat_idx.to(torch.long).contiguous()
        b = idx.numel()
        dev = self.planes.device
        obs = torch.empty((b, 2, self.m, self.n), dtype=obs_dtype, device=dev)
        mask = torch.empty((b, self.action_dim), dtype=torch.bool, device=dev)
        if b:
            mnk_hip.call("mnk_gather_obs", mnk_hip.ptr(self.planes), self.n_steps, self.num_envs, self.m, self.n,
                         mnk_hip.ptr(idx), b, mnk_hip.ptr(obs), mnk_hip.obs_code(obs), mnk_hip.ptr(mask), 1,
                         mnk_hip.ptr(self._err), mnk_hip.stream_ptr(dev))
        return obs, mask

    def all_gather(self, exchange=None, group=None, stream=None, out=None, fields=UPDATE_FIELDS):
        """The exchange step of a SHARDED self-play rollout (the RCCL all-gather of rollout buffers,
        with the log-probabilities and values a network policy adds): every rank has filled its buffer from its own
        block of envs (``wrapper.env_id0 = rank * num_envs``, the samplers alike) and computed its advantages; this
        returns a buffer of ``world * n_steps`` steps of ``num_envs`` envs holding everybody's -- rank r's rollout in steps
        ``[r * n_steps, (r + 1) * n_steps)``, i.e. flat sample id ``(r * n_steps + t) * num_envs + i`` -- whose
        ``get_data_loader`` draws minibatches over all ``world * n_steps * num_envs`` samples.  On the wire: the packed
        planes and 20 B of scalars per agent-step (52 B at 9x9; the reference's layout is 750 B) with ``fields`` =
        ``UPDATE_FIELDS`` (what the PPO update reads; ``returns`` are recomputed as ``advantages + values``, bit for bit
        what the GAE kernel stored), everything with ``ALL_FIELDS``.  ``exchange`` / ``group`` / ``stream``: see
        ``all_gather_fields``; ``out``: the buffer a previous call returned (reused)."""
        if self.ptr != self.n_steps:
            raise RuntimeError(f"all_gather of a buffer that holds {self.ptr} of {self.n_steps} steps")
        unknown = [f for f in fields if f not in ALL_FIELDS]
        if unknown:
            raise ValueError(f"unknown fields {unknown}")
        if exchange is not None:
            world = exchange.world
        else:
            import torch.distributed as dist

            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if out is None:
            out = PackedRolloutBuffer(world * self.n_steps, self.num_envs, self.m, self.n, device=self.device, keep_storage=True)
        if (out.n_steps, out.num_envs, out.m, out.n) != (world * self.n_steps, self.num_envs, self.m, self.n):
            raise ValueError("out was made for another world size / buffer shape")
        all_gather_fields({f: getattr(self, f) for f in fields}, exchange, group, stream,
                          out={f: getattr(out, f) for f in fields})
        if "returns" not in fields and "advantages" in fields and "values" in fields:
            ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
            with ctx:  # (after the collectives, on their stream)
                torch.add(out.advantages, out.values, out=out.returns)  # rollout_buffer.py:79
        out.ptr = out.n_steps
        return out

    def get_data_loader(self, batch_size, normalize_advantages=True):
        """reference rollout_buffer.py:82-113: (obs, actions, log_probs, returns, advantages, masks, values)"""
        steps = self.ptr
        total = steps * self.num_envs
        adv = self.advantages[:steps].reshape(total)
        if normalize_advantages:
            adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        flat = {"actions": self.actions[:steps].reshape(total), "log_probs": self.log_probs[:steps].reshape(total),
                "returns": self.returns[:steps].reshape(total), "advantages": adv,
                "values": self.values[:steps].reshape(total)}
        order = torch.randperm(total, device=self.device)
        for lo in range(0, total, batch_size):
            pick = order[lo:lo + batch_size]
            obs, mask = self.gather(pick)
            yield (obs, flat["actions"][pick], flat["log_probs"][pick], flat["returns"][pick],
                   flat["advantages"][pick], mask, flat["values"][pick])

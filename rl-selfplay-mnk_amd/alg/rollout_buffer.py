"""MI355X drop-in for the reference's ``RolloutBuffer`` (``/root/reference/src/alg/rollout_buffer.py:4-113``),
the sink of the rollout path.

Same constructor, fields (``[T, N, ...]`` device tensors, same names and dtypes), ``add``, ``ptr`` /
"Buffer was full." behaviour and minibatch generator, so ``PPOAgent`` (``alg/ppo.py:67-69, 106-108, 134,
148, 178``) uses it unchanged.  ``compute_advantages_and_returns`` -- in the reference a Python loop of T
steps x ~8 eager launches -- is one launch of ``mnk_gae`` (one lane per env, reverse scan over T, coalesced
over the env axis) with the reference's operation order and f32 roundings, so the results are bit-identical.

The buffer is also the SINK of the fused step: ``wrapper.attach_sink(buffer)`` makes
``TorchSelfPlayWrapper.step`` write the next observation / mask into row ``ptr + 1`` and the rewards / terminated flags
into row ``ptr`` directly (``row(t)``, ``step_outputs()``), and ``add`` skips every field it is handed back as its own
row -- the reference's loop (alg/ppo.py:93-108) runs unchanged while the 7 ``copy_`` of rollout_buffer.py:47-58 shrink to
the small per-env vectors.  While a sink is attached the storage survives ``reset()`` (zeroed in place -- except the row
that holds the observation handed out last, which the caller is about to act on); without one ``reset()`` allocates fresh
tensors as the reference does, so views a caller kept (logged batches, a live ``get_data_loader``) keep their data.

This directory has no ``__init__.py`` on purpose: ``alg`` is a namespace package in the reference too, so
with ``rl-selfplay-mnk_amd/`` ahead of the reference's ``src/`` on ``sys.path`` this module replaces
``alg.rollout_buffer`` while ``alg.ppo`` etc. keep resolving to the reference.
"""
import torch

import mnk_hip


def _same_storage(dst: torch.Tensor, src) -> bool:
    """is ``src`` the very row ``dst`` (written in place by the fused step)?"""
    return (isinstance(src, torch.Tensor) and src.data_ptr() == dst.data_ptr() and src.dtype == dst.dtype
            and src.shape == dst.shape and src.device == dst.device and src.is_contiguous())


def _put(dst: torch.Tensor, src) -> int:
    """dst <- src unless src is dst; returns the bytes copied"""
    if _same_storage(dst, src):
        return 0
    dst.copy_(src)
    return dst.numel() * dst.element_size()


class RolloutBuffer:
    def __init__(self, n_steps, num_envs, obs_shape, action_dim, device="cpu", obs_dtype=torch.float32, keep_storage=False):
        """``keep_storage``: ``reset()`` zeroes the tensors in place instead of allocating fresh ones (the reference
        allocates, rollout_buffer.py:13-45).  ``wrapper.attach_sink(buffer)`` and ``GraphedRollout`` switch it on: rows
        handed to the step kernels must stay where they are."""
        self.keep_storage = keep_storage
        self._live = None  # row of observations / action_masks that holds the observation handed out last (sink only)
        self.n_steps = n_steps
        self.num_envs = num_envs
        self.obs_shape = obs_shape
        self.action_dim = action_dim
        self.device = device
        self.obs_dtype = obs_dtype  # float32 = the reference's; bfloat16 / uint8: the env's opt-in narrow observations
        self.observations = None
        self.copied_bytes = 0  # bytes ``add`` really copied so far (fields written in place by the fused step cost none)
        if torch.device(device).type != "cuda":
            raise RuntimeError("RolloutBuffer: this is the MI355X implementation (GAE runs in a HIP kernel); "
                               "it needs a GPU device")
        mnk_hip.load()
        self.reset()

    _FIELDS = ("observations", "actions", "log_probs", "rewards", "values", "returns", "advantages", "dones",
               "action_masks")

    def reset(self):
        """reference rollout_buffer.py:13-45: all fields zeroed, write pointer at 0.  The reference allocates fresh
        tensors every time, and so does this buffer -- unless ``keep_storage`` (a sink is attached / a graph holds row
        pointers): then the storage is zeroed in place, so rows handed out to the fused step stay valid across
        ``PPOAgent.learn`` calls.  In-place zeroing spares the row that holds the observation the caller acts on next
        (the spill row after a full rollout; row 0 right after ``wrapper.reset()``): ``add`` takes it from there."""
        t, n, dev = self.n_steps, self.num_envs, self.device
        if self.observations is not None and self.keep_storage:
            for name in self._FIELDS:
                if name in ("observations", "action_masks") and self._live is not None and self._live < t:
                    store = getattr(self, name)
                    store[:self._live].zero_()
                    store[self._live + 1:].zero_()
                else:
                    getattr(self, name).zero_()
            self.ptr = 0
            return

        def field(*shape, dtype=torch.float32, rows=t):
            return torch.zeros((rows, n) + shape, dtype=dtype, device=dev)

        # one row more than n_steps behind observations / action_masks: the spill row that takes the observation
        # following the last step (PPOAgent._last_obs).  The public fields are views of the first n_steps rows.
        # A one-step buffer gets TWO spill rows, used in turn: there the step that acts on the carried-over observation
        # is also the last step, and its next observation must not land on the row `add` is still to read.
        spill = 1 if t >= 2 else 2
        carried = None
        if self.observations is not None and self._live is not None:  # a sink was detached: carry the live observation over
            carried = (self._obs_store[self._live].clone(), self._mask_store[self._live].clone())
        self._obs_store = field(*self.obs_shape, dtype=self.obs_dtype, rows=t + spill)
        self._mask_store = field(self.action_dim, dtype=torch.bool, rows=t + spill)
        if carried is not None:
            self._live = t
            self._obs_store[t].copy_(carried[0])
            self._mask_store[t].copy_(carried[1])
        self.observations = self._obs_store[:t]
        self.actions = field(dtype=torch.long)
        self.log_probs = field()
        self.rewards = field()
        self.values = field()
        self.returns = field()
        self.advantages = field()
        self.dones = field(dtype=torch.bool)
        self.action_masks = self._mask_store[:t]
        self.ptr = 0

    # ------------------------------------------------------------------ the sink of the fused step
    def sink_attached(self, on: bool) -> None:
        """called by ``TorchSelfPlayWrapper.attach_sink``: from now on the step kernels hold row pointers of this buffer"""
        if on:
            self.keep_storage = True

    def row(self, t: int) -> dict:
        """Views of row ``t`` of every per-step field (``t == n_steps``: the spill row, observation and mask only): what
        ``TorchSelfPlayWrapper.step(actions, out=...)`` and the step kernels' folded-in draw can write in place."""
        if not 0 <= t < self._obs_store.shape[0]:
            raise IndexError(f"row {t} of a buffer of {self.n_steps} steps")
        out = {"observation": self._obs_store[t], "action_mask": self._mask_store[t]}
        if t < self.n_steps:
            out.update(actions=self.actions[t], log_probs=self.log_probs[t], rewards=self.rewards[t],
                       values=self.values[t], dones=self.dones[t], terminated=self.dones[t])
        return out

    def reset_outputs(self):
        """where ``wrapper.reset()`` puts the first observation: the row the next ``add`` fills"""
        if self.ptr >= self.n_steps:
            return None
        self._live = self.ptr
        r = self.row(self.ptr)
        return {"observation": r["observation"], "action_mask": r["action_mask"]}

    def _next_row(self, t: int) -> int:
        """the row that takes the observation following step t: t + 1, the spill row after the last step -- of a
        one-step buffer's two spill rows the one that does not hold the observation being acted on"""
        nxt = t + 1
        if nxt == self.n_steps and self._live == nxt and self._obs_store.shape[0] > nxt + 1:
            nxt += 1
        return nxt

    def step_outputs(self):
        """where ``wrapper.step()`` puts its outputs while the write pointer stands at row t: next observation / mask
        -> row t+1 (the spill row after the last step), rewards / terminated -> row t"""
        t = self.ptr
        if t >= self.n_steps:
            return None
        self._live = self._next_row(t)
        nxt = self.row(self._live)
        return {"observation": nxt["observation"], "action_mask": nxt["action_mask"], "rewards": self.rewards[t],
                "terminated": self.dones[t]}

    def add(self, obs, action, reward, value, log_prob, done, action_mask):
        """reference rollout_buffer.py:47-58; a field that already IS this row (the fused step wrote it in place) is
        not copied"""
        if self.ptr >= self.n_steps:
            raise IndexError("Buffer was full.")
        row = self.ptr
        self.copied_bytes += (_put(self.observations[row], obs) + _put(self.actions[row], action) +
                              _put(self.rewards[row], reward) + _put(self.values[row], value.view(-1)) +
                              _put(self.log_probs[row], log_prob) + _put(self.dones[row], done) +
                              _put(self.action_masks[row], action_mask))
        self.ptr += 1

    def compute_advantages_and_returns(self, last_values, gamma=0.99, gae_lambda=0.95):
        """reference rollout_buffer.py:60-80, one kernel launch"""
        steps = self.ptr
        if steps == 0 or self.num_envs == 0:
            return
        last_values = last_values.reshape(self.num_envs).to(torch.float32).contiguous()
        dev = self.rewards.device
        mnk_hip.call("mnk_gae", mnk_hip.ptr(self.rewards), mnk_hip.ptr(self.values), mnk_hip.ptr(self.dones),
                     mnk_hip.ptr(last_values), self.num_envs, steps, float(gamma), float(gamma * gae_lambda),
                     mnk_hip.ptr(self.advantages), mnk_hip.ptr(self.returns), mnk_hip.stream_ptr(dev))

    def get_data_loader(self, batch_size, normalize_advantages=True):
        """reference rollout_buffer.py:82-113: one shuffled pass over the first ``ptr`` steps"""
        steps = self.ptr
        total = steps * self.num_envs
        flat = {
            "obs": self.observations[:steps].reshape(total, *self.obs_shape),
            "actions": self.actions[:steps].reshape(total),
            "log_probs": self.log_probs[:steps].reshape(total),
            "returns": self.returns[:steps].reshape(total),
            "advantages": self.advantages[:steps].reshape(total),
            "masks": self.action_masks[:steps].reshape(total, self.action_dim),
            "values": self.values[:steps].reshape(total),
        }
        if normalize_advantages:
            adv = flat["advantages"]
            flat["advantages"] = (adv - adv.mean()) / (adv.std() + 1e-8)
        order = torch.randperm(total, device=self.device)
        for lo in range(0, total, batch_size):
            pick = order[lo:lo + batch_size]
            yield tuple(flat[key][pick] for key in ("obs", "actions", "log_probs", "returns", "advantages",
                                                    "masks", "values"))

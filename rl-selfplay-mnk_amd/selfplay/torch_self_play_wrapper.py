"""MI355X drop-in for the reference's ``TorchSelfPlayWrapper``.

Surface of ``/root/reference/src/selfplay/torch_self_play_wrapper.py:6-115``
(SURVEY.md §8b): ``reset(seed, options) -> (obs, {})``, ``step(actions) ->
(obs, rewards, terminated, truncated, {})`` with NEXT_STEP autoreset, the opponent
replying inside ``step`` and zero-sum rewards; attributes ``env``, ``device``,
``num_envs``, ``opponent_policy``, ``agent_side`` (i64), ``pending_resets`` (bool).

What differs is how a step executes.  The reference builds ``nonzero`` index lists
for "envs to reset", "envs to play" and "envs where the opponent replies" -- 8-10
host synchronisations and 150-200 eager launches per step.  Here every env carries
those facts as bits and one step is

    mnk_selfplay_pre   reset-or-agent-ply, side draw, opponent's view   (1 launch)
    opponent_policy.act(full batch)                                      (caller's torch code, same stream)
    mnk_selfplay_post  opponent ply, zero-sum merge, agent's canonical view (1 launch)

with no host synchronisation; with the built-in ``RandomPolicy`` as opponent the
three collapse into the single launch ``mnk_selfplay_step_random``.  The masked draws fold into these kernels too
(SURVEY.md section 7 step 5): a ``FusedNNPolicy`` opponent only runs its network, its mask + softmax + draw happen inside
``mnk_selfplay_post_logits``; ``step_logits`` does the same for the agent (``mnk_selfplay_pre_logits`` /
``mnk_selfplay_step_random_logits``) -- a network-vs-network agent-step is 2 env-side launches, not 4.

Every output of a step is caller-ownable: ``step(actions, out={...})`` writes the next observation, mask, rewards and
terminated flags straight into the tensors it is given, and ``attach_sink(buffer)`` makes the wrapper take them from
the rollout buffer itself -- row t+1 of ``observations`` / ``action_masks``, row t of ``rewards`` / ``dones`` -- so an
agent-step reaches HBM once (the reference writes it, then ``RolloutBuffer.add`` reads and
writes it again, alg/rollout_buffer.py:47-58).

Differences a caller can observe, both deliberate:
  * the opponent policy is called ONCE per step on the full batch of N rows (rows that
    need no reply carry their current position and their answer is ignored), where the
    reference calls it up to twice on compacted subsets (wrapper:46 and :59).  A policy
    whose action for a row depends only on that row is unaffected;
  * fresh sides come from Philox keyed by (seed, env, step) instead of ``torch.randint``
    on the global generator (wrapper:26, :43-45); ``force_sides`` replays a given stream.
"""
from typing import Optional

import torch

import mnk_hip
from env.constants import PLAYER_BLACK, PLAYER_WHITE  # noqa: F401  (same import as the reference, wrapper:3)


class TorchSelfPlayWrapper:
    def __init__(self, env, seed: Optional[int] = None):
        self.env = env
        self.device = env.device
        self.num_envs = env.num_envs
        self._dev = env._dev

        self.opponent_policy = None
        self.agent_side = torch.zeros(self.num_envs, dtype=torch.long, device=self._dev)
        self.pending_resets = torch.zeros(self.num_envs, dtype=torch.bool, device=self._dev)

        from selfplay.policy import default_key

        self.seed = default_key(seed)  # side draws and the built-in random opponent: streams SIDE / OPP of this key
        self.env_id0 = 0          # global id of env 0 (rank * num_envs when the env axis is sharded)
        self.step_count = 0       # Philox step counter: one per reset()/step() call
        self._forced_sides = None
        self._flags = torch.zeros(self.num_envs, dtype=torch.uint8, device=self._dev)
        self._no_actions = torch.zeros(self.num_envs, dtype=torch.long, device=self._dev)
        self._ep_return = self._ep_length = self._ep_stats = None  # see track_episodes()
        self.step_dev = None  # optional device int64[1] added to step_count inside the kernels (graph replays)
        self._sink = None     # see attach_sink()
        self._captured_by = None  # the GraphedRollout / GraphedAgentStep that has captured this wrapper's step, if any
        self.fuse_opponent_draw = True  # a FusedNNPolicy opponent's draw runs inside mnk_selfplay_post_logits (False: as a
        self.last_opponent_actions = None  # launch of its own before mnk_selfplay_post -- same results, for A/B timing)
        self._truncated = torch.zeros(self.num_envs, dtype=torch.bool, device=self._dev)  # wrapper:66: always all-False

    def set_opponent(self, policy):  # reference wrapper:16-17
        # a captured graph (selfplay.graphed) holds the opponent it was captured with: it takes the new policy's weights
        # in place where it can (train.py:114 then works unchanged, no capture) and recaptures at its next run otherwise
        if self._captured_by is not None and self._captured_by.adopt_opponent(policy):
            return
        self.opponent_policy = policy

    def force_sides(self, sides) -> None:
        """Sides to hand out at the next (auto)reset instead of the Philox draw: an int, an
        (N,) tensor, or None to go back to drawing.  Only envs that actually reset read it."""
        self._forced_sides = None if sides is None else self._side_tensor(sides)

    def _side_tensor(self, sides) -> torch.Tensor:
        """int or (N,) -> contiguous int64 [N]; the kernels read element i for every env i < N"""
        t = torch.as_tensor(sides, device=self._dev).to(torch.long)
        if t.dim() == 0:
            return t.expand(self.num_envs).contiguous()
        t = t.reshape(-1).contiguous()
        if t.numel() != self.num_envs:
            raise IndexError(f"shape mismatch: {t.numel()} sides for {self.num_envs} envs")
        return t

    # ------------------------------------------------------------------ the rollout sink
    def attach_sink(self, sink) -> None:
        """Write every step's outputs straight into a rollout buffer (``alg.rollout_buffer.RolloutBuffer`` or
        ``alg.packed_rollout_buffer.PackedRolloutBuffer``; anything with ``reset_outputs()`` / ``step_outputs()``).

        With the buffer's write pointer at row t, ``step`` puts the next observation and mask into row t+1 of
        ``observations`` / ``action_masks``, the rewards into row t of ``rewards`` and the terminated flags into row t of
        ``dones``, and returns exactly those rows; ``reset`` puts the first observation into row t.  ``buffer.add``
        recognises its own rows and copies nothing, so the reference's rollout loop (alg/ppo.py:93-108) runs unchanged
        and the 7 ``copy_`` per step of alg/rollout_buffer.py:47-58 move ~1 MB instead of ~100 MB at 65 536 envs.  The
        observation that follows the buffer's last row goes to a spill row of the buffer (``PPOAgent._last_obs`` across
        ``learn`` calls); the buffer must keep its storage across ``reset()`` (the drop-in buffers do).
        ``attach_sink(None)`` detaches.  One-step buffers work too (they keep two spill rows and use them in turn)."""
        if sink is not None and hasattr(sink, "sink_attached"):
            sink.sink_attached(True)  # its reset() must zero in place from now on: the step kernels hold its row pointers
        self._sink = sink

    def _sink_outputs(self, is_reset: bool):
        if self._sink is None:
            return None
        return self._sink.reset_outputs() if is_reset else self._sink.step_outputs()

    # ------------------------------------------------------------------ device-side episode statistics
    def track_episodes(self, on: bool = True) -> None:
        """Keep per-env running return / length and finished-episode counters on the device, updated
        inside the step kernels -- what ``alg/ppo.py:110-120`` computes on the host with two
        synchronisations per step.  Read (and clear) them with ``pop_episode_stats()``."""
        if on and self._ep_stats is None:
            self._ep_return = torch.zeros(self.num_envs, dtype=torch.float32, device=self._dev)
            self._ep_length = torch.zeros(self.num_envs, dtype=torch.int32, device=self._dev)
            self._ep_stats = torch.zeros((mnk_hip.STATS_REPLICAS, mnk_hip.STATS_STRIDE), dtype=torch.int64,
                                         device=self._dev)
        elif not on:
            self._ep_return = self._ep_length = self._ep_stats = None

    def pop_episode_stats(self) -> dict:
        """Episodes finished since the last call (one host synchronisation): counts, mean reward, mean length.
        Also the place where a device-side error recorded by an earlier step (an action outside [-C, C): the
        env is left untouched and would stall) surfaces as an exception on non-strict envs."""
        if self._ep_stats is None:
            raise RuntimeError("call track_episodes() first")
        self.env.check_errors()
        episodes, wins, losses, draws, length = self._ep_stats.sum(dim=0)[:mnk_hip.STATS_COUNTERS].tolist()
        self._ep_stats.zero_()
        return {"episodes": episodes, "wins": wins, "losses": losses, "draws": draws,
                "mean_reward": (wins - losses) / episodes if episodes else 0.0,
                "mean_length": length / episodes if episodes else 0.0}

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self) -> dict:
        """Env state + sides, pending resets, Philox key and step counter (+ episode accounting when tracked): a
        wrapper restored from it continues exactly where this one stands.  ``step_count`` is the EFFECTIVE Philox step:
        under a captured graph (``selfplay.graphed``) the advancing part lives in the device counter ``step_dev`` and is
        added here, so a checkpoint taken after graphed rollouts does not replay their side / opponent draws."""
        step = self.step_count + (int(self.step_dev.item()) if self.step_dev is not None else 0)
        out = {"env": self.env.state_dict(), "agent_side": self.agent_side.cpu(), "pending_resets": self.pending_resets.cpu(),
               "seed": self.seed, "step_count": step, "env_id0": self.env_id0}
        if self._ep_stats is not None:
            out["episodes"] = (self._ep_return.cpu(), self._ep_length.cpu(), self._ep_stats.cpu())
        return out

    def load_state_dict(self, state: dict) -> None:
        self.env.load_state_dict(state["env"])
        self.agent_side.copy_(state["agent_side"])
        self.pending_resets.copy_(state["pending_resets"])
        self.seed, self.env_id0 = int(state["seed"]), int(state["env_id0"])
        if self.step_dev is not None:
            # captured: the kernels compute step_count (baked into the graph's nodes) + *step_dev -- move the device part
            self.step_dev.fill_(int(state["step_count"]) - self.step_count)
        else:
            self.step_count = int(state["step_count"])
        if "episodes" in state:
            self.track_episodes()
            for dst, src in zip((self._ep_return, self._ep_length, self._ep_stats), state["episodes"]):
                dst.copy_(src)

    # ------------------------------------------------------------------ reference surface
    def reset(self, seed=None, options=None, out=None):
        """reference wrapper:19-30 (``seed`` is accepted and ignored there; here it re-keys Philox).
        ``out``: see ``step``."""
        if seed is not None:
            self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        forced = self._forced_sides
        if options and "agent_side" in options:
            forced = self._side_tensor(options["agent_side"])
        # a reset is a step in which every env is pending: boards cleared, sides handed out,
        # the opponent opens wherever the agent is white, and its outcome is ignored (:28)
        self.pending_resets.fill_(True)
        if self._ep_stats is not None:
            self._ep_return.zero_()
            self._ep_length.fill_(-1)  # the reset itself is not a step of the first episode
        obs, _, _, _, _ = self._advance(self._no_actions, forced, out if out is not None else self._sink_outputs(True))
        return obs, {}

    def step(self, actions: torch.Tensor, out=None):
        """reference wrapper:32-67.

        ``out``: optional dict of caller-owned, contiguous device tensors the step writes into instead of fresh ones --
        ``"observation"`` (N, 2, m, n) float32 / bfloat16 / uint8, ``"action_mask"`` (N, C) bool, ``"rewards"`` (N,)
        float32, ``"terminated"`` (N,) bool, ``"packed"`` int64 (2, W, N) (the observation as packed planes, extra).
        Rows of a ``[T, N, ...]`` rollout buffer qualify; the returned tensors ARE the given ones.  Missing keys get fresh
        tensors (``env.obs_dtype`` for the observation), as the reference returns."""
        a = torch.as_tensor(actions, device=self._dev).to(torch.long).reshape(-1).contiguous()
        if a.numel() != self.num_envs:
            raise IndexError(f"shape mismatch: {a.numel()} actions for {self.num_envs} envs")
        return self._advance(a, self._forced_sides, out if out is not None else self._sink_outputs(False))

    def get_agent_obs(self):
        """reference wrapper:99-115: the agent's stones in channel 0, mask[.,0] forced on full boards"""
        env = self.env
        obs = torch.empty((self.num_envs, 2, env.m, env.n), dtype=env.obs_dtype, device=self._dev)
        mask = torch.empty((self.num_envs, env.max_moves), dtype=torch.bool, device=self._dev)
        env.observe_into(obs, mask, flip_side=self.agent_side, fix_empty_mask=True)
        return {"observation": obs, "action_mask": mask}

    _get_canonical_obs = get_agent_obs

    def packed_obs(self) -> torch.Tensor:
        """The current canonical observation as packed planes, int64 [2, W, N] (channel 0 = the agent's
        stones): 32 B per env at 9x9 instead of the 729 B of observation + mask.  Feed it to
        ``alg.packed_rollout_buffer.PackedRolloutBuffer.add``; take it before the next ``step``."""
        packed = torch.empty_like(self.env._planes)
        self.env.observe_into(flip_side=self.agent_side, packed=packed)
        return packed

    # ------------------------------------------------------------------ one fused step
    def _out_tensor(self, out, key, shape, dtype):
        """the caller's tensor for ``key`` (checked: it is handed to a kernel as a raw pointer) or a fresh one"""
        t = out.get(key) if out else None
        if t is None:
            return torch.empty(shape, dtype=dtype, device=self._dev)
        if dtype is None:  # the observation: any of the three dtypes the kernels write
            mnk_hip.obs_dtype_code(t.dtype)
        elif t.dtype != dtype:
            raise TypeError(f"out[{key!r}] must be {dtype}, not {t.dtype}")
        if tuple(t.shape) != tuple(shape) or not t.is_contiguous() or t.device != self._dev:
            raise ValueError(f"out[{key!r}] must be a contiguous {tuple(shape)} tensor on {self._dev}, got "
                             f"{tuple(t.shape)} (contiguous: {t.is_contiguous()}) on {t.device}")
        return t

    def step_logits(self, logits, action_mask, sampler, deterministic: bool = False, out=None, actions_out=None,
                    logp_out=None):
        """``step`` with the agent's masked draw folded into the step kernel: instead of actions the step takes the RAW
        logits of the agent's policy head on the current observation (f32 / bf16 ``[N, C]``; ``None`` = the uniformly
        random agent), the mask that observation came with, and a ``selfplay.policy.HipSampler`` (e.g. a
        ``FusedNNPolicy``'s ``_sampler``), and does what ``FusedNNPolicy.act`` + ``step`` do -- mask, softmax, draw
        (policy.py:46-52 over cnn.py:69-79), log-probability (ppo.py:96-97), then wrapper:32-67 -- with one launch
        fewer.  Same random stream: bit-identical to ``sampler.draw(logits, mask, ..., want_logp=True)`` followed by
        ``step(actions)``.  Returns ``(obs, rewards, terminated, truncated, {"actions": ..., "log_probs": ...})``;
        ``actions_out`` int64 ``[N]`` / ``logp_out`` float32 ``[N]``: caller-owned tensors for the two (rows of a
        rollout buffer)."""
        from selfplay.policy import HipSampler

        logits, dtype, mask = HipSampler.prepare(logits, action_mask)
        n = self.num_envs
        if mask.shape != (n, self.env.max_moves):
            raise IndexError(f"shape mismatch: mask {tuple(mask.shape)} for {n} envs of {self.env.max_moves} cells")
        acts = actions_out if actions_out is not None else torch.empty(n, dtype=torch.long, device=self._dev)
        logp = logp_out if logp_out is not None else torch.empty(n, dtype=torch.float32, device=self._dev)
        if acts.dtype != torch.long or logp.dtype != torch.float32 or acts.shape != (n,) or logp.shape != (n,):
            raise TypeError("actions_out must be int64 [N], logp_out float32 [N]")
        draw = (mnk_hip.ptr(logits), dtype, mnk_hip.ptr(mask)) + sampler.block(deterministic) + (mnk_hip.ptr(acts), mnk_hip.ptr(logp))
        keep = (logits, mask)  # alive until the launch is enqueued
        obs, rew, term, trunc, _ = self._advance(None, self._forced_sides, out if out is not None else self._sink_outputs(False),
                                                 agent_draw=draw)
        del keep
        sampler.advance()
        return obs, rew, term, trunc, {"actions": acts, "log_probs": logp}

    def _advance(self, actions, forced, out=None, agent_draw=None):
        """One reset / step.  ``agent_draw``: the sampler block of ``mnk_selfplay_*_logits`` (logits, dtype, mask, seed,
        seed_dev, step, step_dev, env_id0, deterministic, actions out, logp out) instead of ``actions``."""
        env = self.env
        n = self.num_envs
        dev = self._dev
        if out and "observation" in out and out["observation"] is not None:
            obs = self._out_tensor(out, "observation", (n, 2, env.m, env.n), None)
        else:
            obs = torch.empty((n, 2, env.m, env.n), dtype=env.obs_dtype, device=dev)
        mask = self._out_tensor(out, "action_mask", (n, env.max_moves), torch.bool)
        rewards = self._out_tensor(out, "rewards", (n,), torch.float32)
        terminated = self._out_tensor(out, "terminated", (n,), torch.bool)
        packed = self._out_tensor(out, "packed", (2, env.words, n), torch.int64) if out and out.get("packed") is not None else None
        step = self.step_count
        if self.step_dev is None:
            self.step_count += 1
        opp = self.opponent_policy
        if n == 0:
            return {"observation": obs, "action_mask": mask}, rewards, terminated, self._truncated, {}

        geo = (mnk_hip.ptr(env._planes), mnk_hip.ptr(env._meta), n, env.m, env.n, env.k)
        if getattr(opp, "fused_uniform_random", False):
            # RandomPolicy opponent: the whole step is one launch (with the agent's draw in it when it comes as logits)
            tail = (mnk_hip.ptr(self.pending_resets), mnk_hip.ptr(self.agent_side), mnk_hip.ptr(forced), self.seed, step,
                    mnk_hip.ptr(self.step_dev), self.env_id0, mnk_hip.ptr(rewards), mnk_hip.ptr(terminated),
                    mnk_hip.ptr(obs), mnk_hip.obs_code(obs), mnk_hip.ptr(mask), mnk_hip.ptr(packed), mnk_hip.ptr(env._err),
                    mnk_hip.ptr(self._ep_return), mnk_hip.ptr(self._ep_length), mnk_hip.ptr(self._ep_stats), env._flags(),
                    env._stream())
            if agent_draw is None:
                mnk_hip.call("mnk_selfplay_step_random", *geo, mnk_hip.ptr(actions), *tail)
            else:
                mnk_hip.call("mnk_selfplay_step_random_logits", *geo, *agent_draw, *tail)
        else:
            if opp is None:
                raise RuntimeError("TorchSelfPlayWrapper: set_opponent(policy) before reset()/step()")
            opp_obs = torch.empty((n, 2, env.m, env.n), dtype=env.obs_dtype, device=dev)
            fold_opp = getattr(opp, "fused_logits", False) and self.fuse_opponent_draw
            opp_mask = torch.empty((n, env.max_moves), dtype=torch.bool, device=dev)
            tail = (mnk_hip.ptr(self.pending_resets), mnk_hip.ptr(self.agent_side), mnk_hip.ptr(forced), self.seed, step,
                    mnk_hip.ptr(self.step_dev), self.env_id0, mnk_hip.ptr(rewards), mnk_hip.ptr(terminated),
                    mnk_hip.ptr(self._flags), mnk_hip.ptr(opp_obs), mnk_hip.obs_code(opp_obs), mnk_hip.ptr(opp_mask),
                    mnk_hip.ptr(env._err), env._flags(), env._stream())
            if agent_draw is None:
                mnk_hip.call("mnk_selfplay_pre", *geo, mnk_hip.ptr(actions), *tail)
            else:
                mnk_hip.call("mnk_selfplay_pre_logits", *geo, *agent_draw, *tail)
            tail = (mnk_hip.ptr(self._flags), mnk_hip.ptr(self.agent_side), mnk_hip.ptr(rewards), mnk_hip.ptr(terminated),
                    mnk_hip.ptr(self.pending_resets), mnk_hip.ptr(obs), mnk_hip.obs_code(obs), mnk_hip.ptr(mask),
                    mnk_hip.ptr(packed), mnk_hip.ptr(env._err), mnk_hip.ptr(self._ep_return), mnk_hip.ptr(self._ep_length),
                    mnk_hip.ptr(self._ep_stats), env._flags(), env._stream())
            opp_view = {"observation": opp_obs, "action_mask": opp_mask}
            if fold_opp:
                # FusedNNPolicy opponent: only its forward is the caller's; mask + softmax + draw happen inside the post
                # kernel (wrapper:91-96 + policy.py:46-52 + cnn.py:69-79), on the opponent's own sampler stream
                from selfplay.policy import HipSampler

                opp_logits, dtype, _ = HipSampler.prepare(opp.logits(opp_view), opp_mask)
                if opp_logits is not None and opp_logits.shape != (n, env.max_moves):  # (None: a uniformly random head)
                    raise IndexError(f"opponent policy returned logits {tuple(opp_logits.shape)} for {n} rows")
                opp_actions = torch.empty(n, dtype=torch.long, device=dev)
                sampler = opp._sampler
                mnk_hip.call("mnk_selfplay_post_logits", *geo, mnk_hip.ptr(opp_logits), dtype, mnk_hip.ptr(opp_mask),
                             *sampler.block(False), mnk_hip.ptr(opp_actions), None, *tail)
                sampler.advance()
            else:
                with torch.no_grad():  # wrapper:91-94: one positional argument, no `deterministic`
                    opp_actions = opp.act(opp_view)
                opp_actions = torch.as_tensor(opp_actions, device=dev).to(torch.long).reshape(-1).contiguous()
                if opp_actions.numel() != n:
                    raise IndexError(f"opponent policy returned {opp_actions.numel()} actions for {n} rows")
                mnk_hip.call("mnk_selfplay_post", *geo, mnk_hip.ptr(opp_actions), *tail)
            self.last_opponent_actions = opp_actions  # (debugging / tests: rows without MNK_SP_NEED_OPP were not played)
        if env.strict:
            env.check_errors()
        # truncated (wrapper:66) is all-False by construction: one persistent tensor instead of a memset per step
        return {"observation": obs, "action_mask": mask}, rewards, terminated, self._truncated, {}

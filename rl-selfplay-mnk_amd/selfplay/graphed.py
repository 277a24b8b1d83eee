"""One agent-step of self-play captured as a HIP graph.

At the reference's training scale (a few hundred to a few thousand envs) a rollout step is launch-bound: the
network forward, the masked draw and the two env kernels are ~20-40 launches of a few microseconds each, issued
from Python (``alg/ppo.py:93-108``: ~150 us of host time per step against ~40 us of GPU work).  ``GraphedAgentStep``
captures that sequence with ``torch.cuda.graph`` (hipGraph underneath) and replays it:

    net(obs, None) -> step kernels with the masked draw folded in (mask + softmax + draw + log-prob + plies) -> next obs

Everything a replay needs lives in static device buffers, including the advancing part of the Philox step
counter (``step_dev`` of the C ABI: a captured kernel's arguments are frozen, so the counter is read from
memory and bumped by one more node of the graph).  Semantics are those of the eager path with
``FusedNNPolicy`` as the agent: same kernels, same random stream.

A captured kernel's pointers are frozen too, so the step cannot write "row t+1 of the rollout buffer" the way the
eager sink does (``wrapper.attach_sink``).  Instead the observation ping-pongs between two static slots and there
are two graphs: graph A reads slot 0 and makes the step kernel write the next observation into slot 1, graph B the
other way round.  Nothing is cloned or copied inside a step -- round 2 cloned the observation it acted on and
copied the next one back, three extra passes over the 729 B/env of observation + mask; what a step returns stays
valid until the step after next overwrites that slot.

    collector = GraphedAgentStep(wrapper, net)      # wrapper.reset() is done inside
    for _ in range(n_steps):
        out = collector.step()                      # dict of static tensors, valid until the step after next
        buffer.add(out["obs"], out["actions"], out["rewards"], out["values"], out["log_probs"], out["dones"], out["mask"])

The opponent policy is whatever ``wrapper.set_opponent`` installed; it is captured too, so it must be
capture-safe (the built-in policies are: ``RandomPolicy`` folds into the step kernel, ``FusedNNPolicy`` runs its network
and its draw folds into the post kernel; its Philox key and position live in device words).  The reference installs a
fresh ``deepcopy`` of the agent as opponent before EVERY rollout (train.py:106-114): ``set_opponent_weights`` does that
to the captured opponent in place -- weights and buffers copied into the captured module's tensors, the sampler re-keyed
through its device words -- with no new capture.  ``wrapper.set_opponent(policy)`` on a captured wrapper does the
same by itself when ``policy`` carries a network of the captured architecture (train.py:114 then runs unchanged), and
marks the graph for a recapture at its next run otherwise.  ``recapture()`` by hand is only for weights swapped by
assignment.
"""


def _as_i64(x: int) -> int:
    """a u64 bit pattern as the int64 a torch tensor holds"""
    x &= 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >= (1 << 63) else x


class _CapturedOpponent:
    """The opponent side of a captured rollout: the sampler's Philox key and position as device words (the kernels read
    ``*seed_dev`` instead of the baked key and add ``*step_dev`` to the baked step), and the in-place swap."""

    def __init__(self, wrapper, dev):
        self.wrapper, self.dev = wrapper, dev
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.sampler = None

    def attach(self):
        """point the current opponent's sampler at the device words (call before warm-up / capture); returns the sampler
        (None for policies that draw by themselves) with ``calls`` = 0: the position lives in ``step_dev`` from here on"""
        opp = self.wrapper.opponent_policy
        # (RandomPolicy folds into the step kernel and draws on the wrapper's own OPP stream: nothing to attach)
        self.sampler = None if getattr(opp, "fused_uniform_random", False) else getattr(opp, "_sampler", None)
        if self.sampler is not None:
            if self.sampler.step_dev is not self.step_dev:  # first capture with this sampler: take over where it stands
                self.step_dev.fill_(self.sampler.calls)
                self.seed_dev.fill_(_as_i64(self.sampler.seed))
            self.sampler.step_dev, self.sampler.seed_dev = self.step_dev, self.seed_dev
            self.sampler.calls = 0
        return self.sampler

    def set_weights(self, source, seed=None):
        """``source``: a module of the captured opponent's architecture or its ``state_dict()``"""
        opp = self.wrapper.opponent_policy
        model = getattr(opp, "model", None)
        if model is None or self.sampler is None:
            raise RuntimeError("the captured opponent is not a FusedNNPolicy: set_opponent(...) + recapture() instead")
        state = source.state_dict() if hasattr(source, "state_dict") else source
        with torch.no_grad():
            model.load_state_dict(state)  # copy_ into the captured module's own parameters and buffers (BN statistics too)
        model.eval()
        from selfplay.policy import default_key

        key = default_key(seed)  # unseeded: a key of its own, as a freshly built FusedNNPolicy would get
        self.sampler.seed = key
        self.seed_dev.fill_(_as_i64(key))
        self.step_dev.zero_()    # a fresh policy starts at call 0
        return key

    def adopt(self, policy) -> bool:
        """``wrapper.set_opponent(policy)`` on a captured wrapper (the reference's train.py:106-114 does that before every
        ``learn``): when ``policy`` carries a network of the captured opponent's architecture -- an ``NNPolicy`` or a
        ``FusedNNPolicy`` around a ``deepcopy`` of the agent, a pool entry -- its weights go into the captured opponent in
        place and the sampler is re-keyed (the policy's own key when it has one), the captured policy object stays
        installed and True is returned; anything else (another kind of policy, another architecture) returns False: the
        owner installs it and captures again at its next run."""
        mine = getattr(self.wrapper.opponent_policy, "model", None)
        theirs = getattr(policy, "model", None)
        if self.sampler is None or mine is None or theirs is None:
            return False
        a, b = mine.state_dict(), theirs.state_dict()
        if a.keys() != b.keys() or any(a[k].shape != b[k].shape or a[k].dtype != b[k].dtype for k in a):
            return False
        theirs_sampler = getattr(policy, "_sampler", None)
        self.set_weights(theirs, seed=theirs_sampler.seed if theirs_sampler is not None else None)
        return True

    def state(self):
        if self.sampler is None:
            return None
        return {"seed": int(self.sampler.seed), "step": int(self.step_dev.item())}

    def load_state(self, state):
        if state is None or self.sampler is None:
            return
        self.sampler.seed = int(state["seed"])
        self.seed_dev.fill_(_as_i64(self.sampler.seed))
        self.step_dev.fill_(int(state["step"]))

import torch

import mnk_hip


def _prepare_board_kernels(env) -> None:
    """Between warm-up and capture: on a board without a built-in kernel variant the library compiles the board's own
    variant of an API kernel once that kernel is hot -- but never under a capture, so a graph captured earlier would
    replay the generic kernels for good.  The warm-up has launched exactly the kernels the capture will: compile those
    now (about a second each, once per board and process; nothing to do on 3x3x3, 9x9x5, 13x13x5, 15x15x5, 19x19x5)."""
    env.specialise_kernels()


class GraphedAgentStep:
    def __init__(self, wrapper, net, seed=None):
        self.wrapper, self.net = wrapper, net
        self.dev = wrapper._dev
        from selfplay.policy import default_key

        self.seed = default_key(seed)  # unseeded: a key of its own, never the opponent sampler's
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.dev)
        env, n = wrapper.env, wrapper.num_envs
        # two slots of (observation, mask); slot `self.cur` holds the observation the next step acts on
        self.obs = [torch.empty((n, 2, env.m, env.n), dtype=env.obs_dtype, device=self.dev) for _ in range(2)]
        self.mask = [torch.empty((n, env.max_moves), dtype=torch.bool, device=self.dev) for _ in range(2)]
        self.cur = 0
        wrapper.reset(out={"observation": self.obs[0], "action_mask": self.mask[0]})
        self._opp = _CapturedOpponent(wrapper, self.dev)
        self.graphs = [None, None]
        self.outs = [None, None]
        self._stale = False
        self.recapture()
        wrapper._captured_by = self

    def _body(self, src: int):
        """One agent-step acting on slot ``src`` and leaving the next observation in slot ``1 - src`` (runs eagerly
        during warm-up, recorded during capture)."""
        w = self.wrapper
        n = w.num_envs
        obs, mask = self.obs[src], self.mask[src]
        with torch.no_grad():
            dist, values = self.net(obs, None)
            logits = dist.logits.contiguous()
            if logits.dtype not in (torch.float32, torch.bfloat16):
                logits = logits.to(torch.float32)
        actions = torch.empty(n, dtype=torch.long, device=self.dev)
        logp = torch.empty(n, dtype=torch.float32, device=self.dev)
        # the agent's masked draw happens inside the step kernel (mnk_selfplay_pre_logits / _step_random_logits)
        draw = (mnk_hip.ptr(logits), mnk_hip.LOGITS_BF16 if logits.dtype == torch.bfloat16 else mnk_hip.LOGITS_F32,
                mnk_hip.ptr(mask), self.seed, None, 0, mnk_hip.ptr(self.step_dev), w.env_id0, 0, mnk_hip.ptr(actions),
                mnk_hip.ptr(logp))
        _, rewards, term, trunc, _ = w._advance(None, w._forced_sides,
                                                {"observation": self.obs[1 - src], "action_mask": self.mask[1 - src]},
                                                agent_draw=draw)
        self.step_dev.add_(1)
        if self._opp.sampler is not None:
            self._opp.step_dev.add_(1)
        return {"obs": obs, "mask": mask, "actions": actions, "log_probs": logp, "values": values,
                "rewards": rewards, "terminated": term, "dones": term}  # truncated is all-False (wrapper:66)

    def recapture(self):
        w = self.wrapper
        w.step_dev = self.step_dev
        self._opp.attach()
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):  # warm-up outside capture, as torch.cuda.graph requires: an even number of
            for _ in range(4):         # steps, so the current observation is back in slot `cur`
                self._body(self.cur)
                self.cur ^= 1
        torch.cuda.current_stream(self.dev).wait_stream(side)
        _prepare_board_kernels(w.env)
        for src in (0, 1):
            self.graphs[src] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graphs[src]):
                self.outs[src] = self._body(src)
        self._stale = False
        # the two captures did not execute: the current observation is still in slot `cur`

    def set_opponent_weights(self, source, seed=None):
        """A new opponent WITHOUT a new capture: ``source`` (a module of the captured opponent's architecture, or its
        ``state_dict()``) is copied into the captured opponent's parameters and buffers and its sampler starts over
        under a new Philox key -- what ``wrapper.set_opponent(FusedNNPolicy(deepcopy(source), seed=seed))`` +
        ``recapture()`` would leave behind, minus the warm-up steps and the capture.  Returns the key."""
        return self._opp.set_weights(source, seed)

    def adopt_opponent(self, policy) -> bool:
        """called by ``wrapper.set_opponent``: see ``_CapturedOpponent.adopt``"""
        if self._opp.adopt(policy):
            return True
        self._stale = True  # another kind of opponent: capture again before the next step
        return False

    def step(self):
        if self._stale:
            self.recapture()
        src = self.cur
        self.graphs[src].replay()
        self.cur ^= 1
        return self.outs[src]

    def current_obs(self):
        """The observation the next step() will act on (static buffers)."""
        return {"observation": self.obs[self.cur], "action_mask": self.mask[self.cur]}


class GraphedRollout:
    """A whole rollout of ``buffer.n_steps`` agent-steps captured as ONE hipGraph whose nodes write straight into the
    rows of the rollout buffer -- the loop of ``PPOAgent.learn`` (alg/ppo.py:93-122) with nothing left on the host.

    A captured kernel's pointers are frozen, but a rollout fills the same buffer rows every time, so each step's node
    can carry its own row pointers: step t reads observation / mask from row t, the step kernel -- the agent's masked draw
    folded in -- writes actions, log-probabilities, rewards and terminated flags into row t and the next observation / mask
    into row t+1 (the spill row after the last step).  The first node copies the spill row -- the observation carried
    over from the previous rollout -- into row 0: the only copy of an observation in the whole rollout.

        roll = GraphedRollout(wrapper, buffer, net)       # wrapper.reset() is done inside
        for _ in range(iterations):
            roll.run()                                    # buffer.ptr == n_steps afterwards
            nxt = roll.next_obs()                         # the observation after the last step (PPOAgent._last_obs)
            _, last_values = net(nxt["observation"], nxt["action_mask"])
            buffer.compute_advantages_and_returns(last_values, gamma, lam); ...update...; buffer.reset()

    ``net(obs, None) -> (dist, values)`` as every reference architecture; ``net=None`` is the uniformly random agent
    (``RandomPolicy``: the draw reads only the mask, values / log-probabilities of the uniform policy are written).
    ``buffer``: ``alg.rollout_buffer.RolloutBuffer`` (dense rows) -- a ``PackedRolloutBuffer`` works with ``net=None``
    or a net that is fed from ``obs_scratch`` (the dense observation then ping-pongs between two scratch slots).
    Same kernels and random streams as the eager loop with ``FusedNNPolicy`` / ``RandomPolicy`` (seeded alike).

    The wrapper (and its opponent's sampler) belong to the graph from here on: their Philox step counters live in the
    graph's device counter, so eager ``wrapper.step`` calls in between would repeat random numbers -- use ``run()`` only,
    or build a fresh wrapper for eager work.  A new opponent before a rollout (train.py:106-114): ``set_opponent_weights``
    (in place, no capture); ``recapture()`` only after ``set_opponent`` with another kind of policy.  ``recapture()`` plays
    ONE real rollout (the warm-up ``torch.cuda.graph`` asks for): the buffer holds it afterwards, like after ``run()``.
    ``state_dict()`` / ``load_state_dict()``: env + wrapper state and every Philox position (the device counters included)."""

    def __init__(self, wrapper, buffer, net=None, seed=None):
        from selfplay.policy import default_key

        self.wrapper, self.buffer, self.net = wrapper, buffer, net
        self.dev = wrapper._dev
        self.seed = default_key(seed)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.dense = hasattr(buffer, "observations")
        env, n = wrapper.env, wrapper.num_envs
        if not self.dense:  # packed buffer: the dense observation lives in two scratch slots
            self.obs_scratch = [torch.empty((n, 2, env.m, env.n), dtype=env.obs_dtype, device=self.dev) for _ in range(2)]
            self.mask_scratch = [torch.empty((n, env.max_moves), dtype=torch.bool, device=self.dev) for _ in range(2)]
        if buffer.n_steps < 2:
            raise ValueError("GraphedRollout needs a buffer of at least 2 steps (a one-step buffer alternates its spill "
                             "rows from rollout to rollout; a captured graph's row pointers are frozen)")
        buffer.keep_storage = True  # the graph's nodes hold row pointers: reset() must zero in place
        buffer.reset()
        steps = buffer.n_steps
        buffer._live = steps  # the carried-over observation lives in the spill row
        spill = dict(buffer.row(steps))
        if not self.dense:
            spill.update(observation=self.obs_scratch[steps & 1], action_mask=self.mask_scratch[steps & 1])
        wrapper.reset(out=spill)  # the first observation arrives where every later rollout finds its carried-over one
        self._opp = _CapturedOpponent(wrapper, self.dev)
        self.graph = None
        self._stale = False
        self.recapture()
        wrapper._captured_by = self

    def _obs_of(self, t):
        if self.dense:
            r = self.buffer.row(t)
            return r["observation"], r["action_mask"]
        return self.obs_scratch[t & 1], self.mask_scratch[t & 1]

    def _body(self):
        w, buf = self.wrapper, self.buffer
        n, steps = w.num_envs, buf.n_steps
        src = buf.row(steps)
        dst = buf.row(0)
        for key in (("observation", "action_mask") if self.dense else ("packed",)):
            dst[key].copy_(src[key])
        if not self.dense and (steps & 1):  # the carried-over dense observation sits in slot steps & 1; step 0 reads slot 0
            self.obs_scratch[0].copy_(self.obs_scratch[1])
            self.mask_scratch[0].copy_(self.mask_scratch[1])
        for t in range(steps):
            obs, mask = self._obs_of(t)
            row = buf.row(t)
            logits = None
            if self.net is not None:
                with torch.no_grad():
                    dist, values = self.net(obs, None)
                    logits = dist.logits.contiguous()
                    if logits.dtype not in (torch.float32, torch.bfloat16):
                        logits = logits.to(torch.float32)
                row["values"].copy_(values.reshape(-1))
            out = {"rewards": row["rewards"], "terminated": row["dones"]}
            nobs, nmask = self._obs_of(t + 1)
            out.update(observation=nobs, action_mask=nmask)
            if not self.dense:
                out["packed"] = buf.row(t + 1)["packed"]
            # the Philox step of this node: the part that differs between the nodes of one rollout is baked in, the part
            # that advances from rollout to rollout is read from step_dev
            w.step_count = self._step0 + t
            if self._opp.sampler is not None:
                self._opp.sampler.calls = t
            # the agent's masked draw -- actions and log-probabilities straight into row t -- inside the step kernel
            draw = (mnk_hip.ptr(logits),
                    mnk_hip.LOGITS_BF16 if logits is not None and logits.dtype == torch.bfloat16 else mnk_hip.LOGITS_F32,
                    mnk_hip.ptr(mask), self.seed, None, t, mnk_hip.ptr(self.step_dev), w.env_id0, 0,
                    mnk_hip.ptr(row["actions"]), mnk_hip.ptr(row["log_probs"]))
            w._advance(None, w._forced_sides, out, agent_draw=draw)
        self.step_dev.add_(steps)
        if self._opp.sampler is not None:
            self._opp.step_dev.add_(steps)

    def recapture(self):
        w = self.wrapper
        w.step_dev = self.step_dev
        self._step0 = w.step_count
        self._opp.attach()
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):  # one real rollout as the warm-up torch.cuda.graph asks for
            self._body()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        _prepare_board_kernels(w.env)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._body()
        w.step_count = self._step0
        if self._opp.sampler is not None:
            self._opp.sampler.calls = 0
        self.buffer.ptr = self.buffer.n_steps  # the warm-up rollout filled the buffer
        self._stale = False

    def adopt_opponent(self, policy) -> bool:
        """called by ``wrapper.set_opponent``: see ``_CapturedOpponent.adopt``"""
        if self._opp.adopt(policy):
            return True
        self._stale = True  # another kind of opponent: the next run() is a recapture (which plays that rollout itself)
        return False

    def set_opponent_weights(self, source, seed=None):
        """A new opponent before the next rollout WITHOUT a new capture (the reference does
        ``set_opponent(NNPolicy(deepcopy(agent.network)))`` before every ``learn()``, train.py:106-114): ``source`` -- a
        module of the captured opponent's architecture or its ``state_dict()`` -- is copied into the captured
        opponent's own parameters and buffers, and its sampler starts over under a new Philox key (``seed``; unseeded:
        a key of its own) through the device words the captured kernels read.  The rollouts that follow equal the eager
        loop after ``set_opponent(FusedNNPolicy(deepcopy(source), seed=seed))`` bit for bit.  Returns the key."""
        return self._opp.set_weights(source, seed)

    def state_dict(self) -> dict:
        """Everything a resumed run needs to continue this one's streams: the wrapper's state (its Philox step counter
        INCLUDES the part that lives in the graph's device counter), the agent sampler's key and position, the captured
        opponent's key and position, the carried-over observation (spill row)."""
        spill = self.buffer.row(self.buffer.n_steps)
        return {"wrapper": self.wrapper.state_dict(), "seed": self.seed, "steps_done": int(self.step_dev.item()),
                "opponent": self._opp.state(),
                "spill": {k: v.cpu() for k, v in spill.items() if k in ("observation", "action_mask", "packed")},
                "scratch": None if self.dense else [(o.cpu(), m.cpu()) for o, m in zip(self.obs_scratch, self.mask_scratch)]}

    def load_state_dict(self, state: dict) -> None:
        """Into a GraphedRollout built the same way (env shape, buffer, networks, ``seed``): no new capture.  The agent
        sampler's key and the wrapper's step at capture time are baked into the graph's nodes, so both must match."""
        if int(state["seed"]) != int(self.seed):
            raise ValueError("the agent sampler's Philox key is baked into the captured graph: build the GraphedRollout "
                             f"with seed={int(state['seed'])} to resume this run")
        done = int(state["steps_done"])
        if int(state["wrapper"]["step_count"]) - done != self._step0:
            raise ValueError(f"the saved run was captured at wrapper step {int(state['wrapper']['step_count']) - done}, this "
                             f"one at {self._step0}: build both from a wrapper in the same state")
        self.wrapper.load_state_dict(state["wrapper"])  # (the wrapper moves the advancing part into step_dev itself)
        assert int(self.step_dev.item()) == done and self.wrapper.step_count == self._step0
        self._opp.load_state(state["opponent"])
        spill = self.buffer.row(self.buffer.n_steps)
        for k, v in state["spill"].items():
            spill[k].copy_(v)
        if state.get("scratch") is not None:
            for (o, m), (so, sm) in zip(zip(self.obs_scratch, self.mask_scratch), state["scratch"]):
                o.copy_(so)
                m.copy_(sm)

    def run(self):
        if self._stale:
            self.recapture()  # (its warm-up IS this rollout)
            return
        self.graph.replay()
        self.buffer.ptr = self.buffer.n_steps

    def next_obs(self):
        """observation / mask that follow the last step (``PPOAgent._last_obs``): feed the net for ``last_values``"""
        obs, mask = self._obs_of(self.buffer.n_steps)
        return {"observation": obs, "action_mask": mask}

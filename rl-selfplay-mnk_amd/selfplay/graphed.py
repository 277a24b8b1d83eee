"""One agent-step of self-play captured as a HIP graph.

At the reference's training scale (a few hundred to a few thousand envs) a rollout step is launch-bound: the
network forward, the masked draw and the two env kernels are ~20-40 launches of a few microseconds each, issued
from Python (``alg/ppo.py:93-108``: ~150 us of host time per step against ~40 us of GPU work).  ``GraphedAgentStep``
captures that sequence with ``torch.cuda.graph`` (hipGraph underneath) and replays it:

    net(obs, None) -> mnk_sample_logits (mask + softmax + draw + log-prob) -> wrapper.step kernels -> next obs

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
capture-safe (the built-in policies are: ``RandomPolicy`` folds into the step kernel, ``FusedNNPolicy`` reads the
same device step counter).  Call ``recapture()`` after ``set_opponent`` or after swapping network weights
by assignment (in-place weight updates, e.g. an optimizer step, need no recapture).
"""
import torch

import mnk_hip


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
        self.graphs = [None, None]
        self.outs = [None, None]
        self.recapture()

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
        mnk_hip.call("mnk_sample_logits", mnk_hip.ptr(logits),
                     mnk_hip.LOGITS_BF16 if logits.dtype == torch.bfloat16 else mnk_hip.LOGITS_F32,
                     mnk_hip.ptr(mask), n, logits.shape[1],
                     self.seed, 0, mnk_hip.ptr(self.step_dev), w.env_id0, 0, mnk_hip.ptr(actions), mnk_hip.ptr(logp),
                     mnk_hip.stream_ptr(self.dev))
        _, rewards, term, trunc, _ = w._advance(actions, w._forced_sides,
                                                {"observation": self.obs[1 - src], "action_mask": self.mask[1 - src]})
        self.step_dev.add_(1)
        return {"obs": obs, "mask": mask, "actions": actions, "log_probs": logp, "values": values,
                "rewards": rewards, "terminated": term, "dones": term}  # truncated is all-False (wrapper:66)

    def recapture(self):
        w = self.wrapper
        w.step_dev = self.step_dev
        opp_sampler = getattr(w.opponent_policy, "_sampler", None)
        if opp_sampler is not None:
            opp_sampler.step_dev = self.step_dev
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):  # warm-up outside capture, as torch.cuda.graph requires: an even number of
            for _ in range(4):         # steps, so the current observation is back in slot `cur`
                self._body(self.cur)
                self.cur ^= 1
        torch.cuda.current_stream(self.dev).wait_stream(side)
        for src in (0, 1):
            self.graphs[src] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graphs[src]):
                self.outs[src] = self._body(src)
        # the two captures did not execute: the current observation is still in slot `cur`

    def step(self):
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
    can carry its own row pointers: step t reads observation / mask from row t, ``mnk_sample_logits`` writes actions and
    log-probabilities into row t, the step kernels write rewards / terminated into row t and the next observation / mask
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
    or build a fresh wrapper for eager work.  ``recapture()`` after ``set_opponent`` or a weight swap by assignment.
    """

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
        buffer.reset()
        steps = buffer.n_steps
        spill = dict(buffer.row(steps))
        if not self.dense:
            spill.update(observation=self.obs_scratch[steps & 1], action_mask=self.mask_scratch[steps & 1])
        wrapper.reset(out=spill)  # the first observation arrives where every later rollout finds its carried-over one
        self.graph = None
        self.recapture()

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
            mnk_hip.call("mnk_sample_logits", mnk_hip.ptr(logits),
                         mnk_hip.LOGITS_BF16 if logits is not None and logits.dtype == torch.bfloat16 else mnk_hip.LOGITS_F32,
                         mnk_hip.ptr(mask), n, mask.shape[1], self.seed, t, mnk_hip.ptr(self.step_dev), w.env_id0, 0,
                         mnk_hip.ptr(row["actions"]), mnk_hip.ptr(row["log_probs"]), mnk_hip.stream_ptr(self.dev))
            out = {"rewards": row["rewards"], "terminated": row["dones"]}
            nobs, nmask = self._obs_of(t + 1)
            out.update(observation=nobs, action_mask=nmask)
            if not self.dense:
                out["packed"] = buf.row(t + 1)["packed"]
            # the Philox step of this node: the part that differs between the nodes of one rollout is baked in, the part
            # that advances from rollout to rollout is read from step_dev
            w.step_count = self._step0 + t
            if self._opp_sampler is not None:
                self._opp_sampler.calls = self._opp_calls0 + t
            w._advance(row["actions"], w._forced_sides, out)
        self.step_dev.add_(steps)

    def recapture(self):
        w = self.wrapper
        w.step_dev = self.step_dev
        self._step0 = w.step_count
        self._opp_sampler = getattr(w.opponent_policy, "_sampler", None)
        if self._opp_sampler is not None:
            self._opp_sampler.step_dev = self.step_dev
            self._opp_calls0 = self._opp_sampler.calls
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):  # one real rollout as the warm-up torch.cuda.graph asks for
            self._body()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._body()
        w.step_count = self._step0
        if self._opp_sampler is not None:
            self._opp_sampler.calls = self._opp_calls0
        self.buffer.ptr = self.buffer.n_steps  # the warm-up rollout filled the buffer

    def run(self):
        self.graph.replay()
        self.buffer.ptr = self.buffer.n_steps

    def next_obs(self):
        """observation / mask that follow the last step (``PPOAgent._last_obs``): feed the net for ``last_values``"""
        obs, mask = self._obs_of(self.buffer.n_steps)
        return {"observation": obs, "action_mask": mask}

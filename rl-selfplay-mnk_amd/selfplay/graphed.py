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

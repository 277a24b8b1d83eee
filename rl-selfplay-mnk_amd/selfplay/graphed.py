"""One agent-step of self-play captured as a HIP graph.

At the reference's training scale (a few hundred to a few thousand envs) a rollout step is launch-bound: the
network forward, the masked draw and the two env kernels are ~20-40 launches of a few microseconds each, issued
from Python (``alg/ppo.py:93-108``: ~150 us of host time per step against ~40 us of GPU work).  ``GraphedAgentStep``
captures that sequence once with ``torch.cuda.graph`` (hipGraph underneath) and replays it:

    net(obs, None) -> mnk_sample_logits (mask + softmax + draw + log-prob) -> wrapper.step kernels -> next obs

Everything a replay needs lives in static device buffers, including the advancing part of the Philox step
counter (``step_dev`` of the C ABI: a captured kernel's arguments are frozen, so the counter is read from
memory and bumped by one more node of the graph).  Semantics are those of the eager path with
``FusedNNPolicy`` as the agent: same kernels, same random stream.

    collector = GraphedAgentStep(wrapper, net)      # wrapper.reset() is done inside
    for _ in range(n_steps):
        out = collector.step()                      # dict of static tensors, overwritten by the next step()
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
        obs, _ = wrapper.reset()
        self.cur_obs = obs["observation"].clone()
        self.cur_mask = obs["action_mask"].clone()
        self.graph = None
        self.out = None
        self.recapture()

    def _body(self):
        """One agent-step on the static buffers (runs eagerly during warm-up, recorded during capture)."""
        w = self.wrapper
        n = w.num_envs
        prev_obs, prev_mask = self.cur_obs.clone(), self.cur_mask.clone()
        with torch.no_grad():
            dist, values = self.net(self.cur_obs, None)
            logits = dist.logits.contiguous()
            if logits.dtype not in (torch.float32, torch.bfloat16):
                logits = logits.to(torch.float32)
        actions = torch.empty(n, dtype=torch.long, device=self.dev)
        logp = torch.empty(n, dtype=torch.float32, device=self.dev)
        mnk_hip.call("mnk_sample_logits", mnk_hip.ptr(logits),
                     mnk_hip.LOGITS_BF16 if logits.dtype == torch.bfloat16 else mnk_hip.LOGITS_F32,
                     mnk_hip.ptr(self.cur_mask), n, logits.shape[1],
                     self.seed, 0, mnk_hip.ptr(self.step_dev), w.env_id0, 0, mnk_hip.ptr(actions), mnk_hip.ptr(logp),
                     mnk_hip.stream_ptr(self.dev))
        nxt, rewards, term, trunc, _ = w._advance(actions, w._forced_sides)
        self.cur_obs.copy_(nxt["observation"])
        self.cur_mask.copy_(nxt["action_mask"])
        self.step_dev.add_(1)
        return {"obs": prev_obs, "mask": prev_mask, "actions": actions, "log_probs": logp, "values": values,
                "rewards": rewards, "terminated": term, "dones": term | trunc}

    def recapture(self):
        w = self.wrapper
        w.step_dev = self.step_dev
        opp_sampler = getattr(w.opponent_policy, "_sampler", None)
        if opp_sampler is not None:
            opp_sampler.step_dev = self.step_dev
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):  # warm-up outside capture, as torch.cuda.graph requires
            for _ in range(3):
                self._body()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._body()

    def step(self):
        self.graph.replay()
        return self.out

    def current_obs(self):
        """The observation the next step() will act on (static buffers)."""
        return {"observation": self.cur_obs, "action_mask": self.cur_mask}

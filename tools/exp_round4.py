"""Developer tool (GPU box): the round-4 measurements that are not part of bench.py.  Run bare for timings, or under
rocprofv3 (tools/profile_script.sh <tag> tools/exp_round4.py <mode>) and condense with tools/summarize_trace.py.

    python tools/exp_round4.py fused [m n k envs]     the step kernels with the masked draw folded in against the two
                                                      launches they replace (mnk_sample_logits + plain kernel), 50 launches
                                                      back to back in a hipGraph: pre / post / step_random x f32 / bf16 / no logits
    python tools/exp_round4.py cadence [envs ...]     a rollout of 64 agent-steps at the reference's cadence -- a NEW
                                                      opponent (deepcopy of the agent) before every rollout, train.py:106-114:
                                                      eager loop with set_opponent(FusedNNPolicy(deepcopy)) against ONE captured
                                                      GraphedRollout with set_opponent_weights, and against recapture per rollout
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch  # noqa: E402

import mnk_hip  # noqa: E402
from env.torch_vector_mnk_env import TorchVectorMnkEnv  # noqa: E402
from selfplay.policy import HipSampler, RandomPolicy  # noqa: E402
from selfplay.random_rollout import RandomRollout  # noqa: E402
from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper  # noqa: E402

DEV = "cuda:0"


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps  # us


def graphed(body, launches=50):
    side = torch.cuda.Stream(DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    torch.cuda.current_stream(DEV).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(launches):
            body()
    return timeit(g.replay) / launches


class FixedLogits:
    """an opponent whose "network" returns a fixed logits tensor: the env side of a network opponent without its forward"""
    fused_logits = True

    def __init__(self, logits, seed):
        self._logits, self._sampler = logits, HipSampler(seed)

    def logits(self, obs):
        return self._logits

    def act(self, obs, deterministic=False):
        return self._sampler.draw(self._logits, obs["action_mask"], deterministic)


def fused(m=9, n=9, k=5, nenv=65536):
    c = m * n
    g = torch.Generator().manual_seed(0)
    f32 = (torch.randn(nenv, c, generator=g) * 2).to(DEV)
    forms = (("f32", f32, 4), ("bf16", f32.to(torch.bfloat16), 2), ("none", None, 0))
    state_rt = 2 * (16 * mnk_hip.state_words(m, n) + 4)
    obs_b = 8 * c + c
    print(f"# {m}x{n}x{k} x {nenv} envs: us per launch, 50 launches back to back in a hipGraph "
          f"(algorithmic MB per agent-step side in brackets)", flush=True)
    for name, logits, eb in forms:
        for kind in ("step_random", "pre+post"):
            rows = {}
            for fold in (False, True):
                env = TorchVectorMnkEnv(m, n, k, nenv, device=DEV)
                RandomRollout(env, seed=0).run(2 * c // 3, record=False)  # mid-game boards
                w = TorchSelfPlayWrapper(env, seed=1)
                w.set_opponent(RandomPolicy(c, seed=2) if kind == "step_random" else FixedLogits(logits, 2))
                w.fuse_opponent_draw = fold
                obs, _ = w.reset()
                out = {"observation": torch.empty_like(obs["observation"]), "action_mask": torch.empty_like(obs["action_mask"]),
                       "rewards": torch.empty(nenv, device=DEV), "terminated": torch.empty(nenv, dtype=torch.bool, device=DEV)}
                mask = obs["action_mask"]
                sampler = HipSampler(seed=3)
                acts = torch.empty(nenv, dtype=torch.long, device=DEV)
                logp = torch.empty(nenv, device=DEV)

                def body():
                    if fold:
                        w.step_logits(logits, mask, sampler, out=out, actions_out=acts, logp_out=logp)
                    else:
                        a, _ = sampler.draw(logits, mask, False, want_logp=True)
                        w.step(a, out=out)

                rows[fold] = graphed(body)
            # agent side: logits + mask in, action + logp out; state round trip(s); next observation + mask out
            agent = eb * c + c + 12
            if kind == "step_random":
                alg = agent + state_rt + obs_b + 4 + 1 + 8 + 2
                launches = "2 -> 1"
            else:
                alg = 2 * agent + 2 * state_rt + 2 * obs_b + 4 + 1 + 8 + 3
                launches = "4 -> 2"
            print(f"logits {name:5s} {kind:12s} launches {launches}: separate {rows[False]:7.2f} us  folded {rows[True]:7.2f} us  "
                  f"x{rows[False] / rows[True]:.2f}   [{alg * nenv / 1e6:.1f} MB: {alg * nenv / rows[True] / 1e3:6.0f} GB/s folded, "
                  f"{alg * nenv / rows[True] / 1e3 / 8000:.2f} of 8 TB/s]", flush=True)


from bench import train_cadence as cadence_times  # noqa: E402  (the driver-run line carries the 384-env row of this)


def cadence(*sizes):
    for nenv in (sizes or (384, 1024, 4096)):
        t_eager, t_swap, t_recap = cadence_times(nenv)
        print(f"new opponent before every rollout of 64 steps, N={nenv:5d}: eager loop {t_eager:8.1f} us/agent-step "
              f"({nenv / t_eager * 1e6:.3e} agent-steps/s)   one graph + set_opponent_weights {t_swap:8.1f} "
              f"({nenv / t_swap * 1e6:.3e}, x{t_eager / t_swap:.2f})   recapture per rollout {t_recap:8.1f} "
              f"(x{t_eager / t_recap:.2f})", flush=True)


def gae_depth(nenv=65536, T=256):
    """mnk_gae by prefetch depth (MNK_GAE_DEPTH): 17 B per (t, env)"""
    from selfplay.random_rollout import gae

    g = torch.Generator().manual_seed(0)
    r = torch.randn(T, nenv, generator=g).to(DEV)
    v = torch.randn(T, nenv, generator=g).to(DEV)
    d = (torch.rand(T, nenv, generator=g) < 0.02).to(DEV)
    last = torch.randn(nenv, generator=g).to(DEV)
    ref = None
    for depth in ("8", "16", "32"):
        os.environ["MNK_GAE_DEPTH"] = depth
        mnk_hip.reload_config()
        adv, ret = gae(r, v, d, last)
        if ref is None:
            ref = (adv.clone(), ret.clone())
        assert torch.equal(adv, ref[0]) and torch.equal(ret, ref[1])
        adv = torch.empty_like(r)
        ret = torch.empty_like(r)

        def launch():
            mnk_hip.call("mnk_gae", mnk_hip.ptr(r), mnk_hip.ptr(v), mnk_hip.ptr(d), mnk_hip.ptr(last), nenv, T, 0.99, 0.99 * 0.95,
                         mnk_hip.ptr(adv), mnk_hip.ptr(ret), mnk_hip.stream_ptr(torch.device(DEV)))

        us = timeit(launch, reps=30, warm=5)
        nbytes = 17 * T * nenv
        print(f"mnk_gae {T} x {nenv} depth {depth:>2s}: {us:7.2f} us  {nbytes / us / 1e3:6.0f} GB/s  ({nbytes / us / 1e3 / 8000:.2f} of 8 TB/s)",
              flush=True)
    os.environ.pop("MNK_GAE_DEPTH", None)
    mnk_hip.reload_config()


def pair_jit(*boards):
    """the run-time specialised rollout kernel of boards without a built-in variant: one lane per env against two lanes per
    env (MNK_ROLLOUT_PAIR=0 / 1), us per 256 plies by batch size"""
    boards = boards or ("12x12x5", "11x11x5", "7x9x7", "10x10x5", "25x25x5")
    os.environ["MNK_JIT"] = "1"
    for board in boards:
        m, n, k = (int(v) for v in board.split("x"))
        for nenv in (8192, 16384, 32768, 49152):
            row = {}
            for pair in ("0", "1"):
                os.environ["MNK_ROLLOUT_PAIR"] = pair
                mnk_hip.reload_config()
                env = TorchVectorMnkEnv(m, n, k, nenv, device=DEV)
                roll = RandomRollout(env, seed=1)
                rec = roll.alloc(256)
                for _ in range(6):
                    roll.run(256, out=rec)
                row[pair] = timeit(lambda: roll.run(256, out=rec), reps=20, warm=3)
            rows = mnk_hip.record_words(m, n)
            print(f"{board:9s} x {nenv:6d} envs: one lane {row['0']:7.1f} us  two lanes {row['1']:7.1f} us  x{row['0'] / row['1']:.2f}   "
                  f"({nenv * 256 / min(row.values()) * 1e6:.3e} env-steps/s, {nenv * 256 * (8 * rows + 4) / min(row.values()) / 1e3:.0f} GB/s)",
                  flush=True)
    os.environ.pop("MNK_ROLLOUT_PAIR", None)
    os.environ.pop("MNK_JIT", None)
    mnk_hip.reload_config()


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
    if mode == "pairjit":
        pair_jit(*sys.argv[2:])
        sys.exit(0)
    {"fused": fused, "cadence": cadence, "gae": gae_depth}[mode](*[int(v) for v in sys.argv[2:]])

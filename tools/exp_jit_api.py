"""Generic against run-time specialised API kernels on boards without a built-in variant (developer tool, GPU box).

Each case: 50 launches captured into a hipGraph (no host gaps), us per call, with MNK_JIT_API=0 (generic kernels: run-time
shift amounts, table write-out, the folded draw as two launches) and with the board's own variants prepared before the
capture (mnk_jit_prepare).  A compiled board of similar size runs beside them for scale."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch

import mnk_hip
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.policy import HipSampler, RandomPolicy
from selfplay.random_rollout import RandomRollout
from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper

DEV = "cuda:0"


def graph_time(fn, n=50, reps=20):
    side = torch.cuda.Stream(DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream(DEV).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


class Fixed:
    """an opponent policy that costs nothing: the same (legal or not -- occupied cells are accepted) actions every call"""

    def __init__(self, acts):
        self.acts = acts

    def act(self, obs):
        return self.acts


def cases(m, n, k, N):
    c = m * n
    env = TorchVectorMnkEnv(m, n, k, N, device=DEV)
    RandomRollout(env, seed=0).run(min(150, c), record=False)  # a stationary mix of positions
    out = {"observation": torch.empty((N, 2, m, n), dtype=torch.float32, device=DEV),
           "action_mask": torch.empty((N, c), dtype=torch.bool, device=DEV),
           "rewards": torch.empty(N, dtype=torch.float32, device=DEV),
           "terminated": torch.empty(N, dtype=torch.bool, device=DEV)}
    acts = torch.zeros(N, dtype=torch.long, device=DEV)
    env.sample_legal_into(acts, seed=1, step=0)
    logits = torch.randn(N, c, device=DEV)
    mask = torch.ones((N, c), dtype=torch.bool, device=DEV)
    sampler = HipSampler(seed=3)
    a_out = torch.empty(N, dtype=torch.long, device=DEV)
    lp_out = torch.empty(N, dtype=torch.float32, device=DEV)
    res = {}

    w = TorchSelfPlayWrapper(env, seed=1)
    w.set_opponent(RandomPolicy(c, seed=2))
    w.reset()
    res["selfplay_step_random"] = graph_time(lambda: w._advance(acts, None, out=out))
    res["step_random_logits f32"] = graph_time(
        lambda: w.step_logits(logits, mask, sampler, out=out, actions_out=a_out, logp_out=lp_out))
    w2 = TorchSelfPlayWrapper(env, seed=1)
    w2.set_opponent(Fixed(acts))
    w2.reset()
    res["selfplay pre + post"] = graph_time(lambda: w2._advance(acts, None, out=out))
    rew, done = out["rewards"], out["terminated"]
    res["step + mask + obs"] = graph_time(lambda: env.step_into(acts, rew, done, out["action_mask"], out["observation"])) \
        if hasattr(env, "step_into") else float("nan")
    res["step_random + mask"] = graph_time(lambda: env.step_random_into(rew, done, out["action_mask"], seed=5, step=0))
    res["observe"] = graph_time(lambda: env.observe_into(out["observation"], out["action_mask"]))
    return res


def main():
    mnk_hip.load()
    boards = [(12, 12, 5), (11, 11, 5), (10, 10, 5), (7, 7, 4), (6, 7, 4)]
    sizes = [65536, 4096]
    all_kinds = list(range(mnk_hip.JIT_API_COUNT))
    for N in sizes:
        for (m, n, k) in boards:
            os.environ["MNK_JIT_API"] = "0"
            mnk_hip.reload_config()
            gen = cases(m, n, k, N)
            os.environ["MNK_JIT_API"] = "1"
            mnk_hip.reload_config()
            mnk_hip.jit_prepare(m, n, k, all_kinds)
            jit = cases(m, n, k, N)
            print(f"{m}x{n}x{k} N={N}:")
            for key in gen:
                print(f"    {key:28s} generic {gen[key]:7.2f} us   specialised {jit[key]:7.2f} us   {gen[key] / jit[key]:.2f}x", flush=True)
        for (m, n, k) in [(13, 13, 5), (9, 9, 5)]:
            ref = cases(m, n, k, N)
            print(f"{m}x{n}x{k} N={N} (built-in variant): " + ", ".join(f"{key} {v:.2f}" for key, v in ref.items()), flush=True)


def profile(m, n, k, N, launches=200):
    """for rocprofv3 (tools/profile_script.sh): the same launches on the generic and on the specialised kernels, eager, so
    that the trace holds both kernel names side by side with their FETCH_SIZE / WRITE_SIZE"""
    mnk_hip.load()
    c = m * n
    for mode in ("0", "1"):
        os.environ["MNK_JIT_API"] = mode
        mnk_hip.reload_config()
        env = TorchVectorMnkEnv(m, n, k, N, device=DEV)
        RandomRollout(env, seed=0).run(min(150, c), record=False)
        out = {"observation": torch.empty((N, 2, m, n), dtype=torch.float32, device=DEV),
               "action_mask": torch.empty((N, c), dtype=torch.bool, device=DEV),
               "rewards": torch.empty(N, dtype=torch.float32, device=DEV),
               "terminated": torch.empty(N, dtype=torch.bool, device=DEV)}
        acts = torch.zeros(N, dtype=torch.long, device=DEV)
        env.sample_legal_into(acts, seed=1, step=0)
        logits = torch.randn(N, c, device=DEV)
        mask = torch.ones((N, c), dtype=torch.bool, device=DEV)
        sampler = HipSampler(seed=3)
        a_out = torch.empty(N, dtype=torch.long, device=DEV)
        lp_out = torch.empty(N, dtype=torch.float32, device=DEV)
        w = TorchSelfPlayWrapper(env, seed=1)
        w.set_opponent(RandomPolicy(c, seed=2))
        w.reset()
        w2 = TorchSelfPlayWrapper(env, seed=1)
        w2.set_opponent(Fixed(acts))
        w2.reset()
        for _ in range(launches):
            w._advance(acts, None, out=out)
            w.step_logits(logits, mask, sampler, out=out, actions_out=a_out, logp_out=lp_out)
            w2._advance(acts, None, out=out)
            env.step_into(acts, out["rewards"], out["terminated"], out["action_mask"], out["observation"])
            env.step_random_into(out["rewards"], out["terminated"], out["action_mask"], seed=5, step=0)
            env.observe_into(out["observation"], out["action_mask"])
        torch.cuda.synchronize()
    print(f"{m}x{n}x{k} x {N} envs: {launches} launches of each call, first on the generic kernels (NW rounded up, CN = CK = 0), "
          f"then on the board's own (<{(m * (n + 1) + 31) // 32}, {n}, {k}>); algorithmic bytes per self-play step "
          f"(state both ways + f32 observation + mask + scalars): {N * (2 * (16 * env.words + 8) + 9 * c + 8 + 4 + 1 + 2 + 16) / 1e6:.1f} MB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "profile":
        profile(*(int(v) for v in sys.argv[2:6]))
    else:
        main()

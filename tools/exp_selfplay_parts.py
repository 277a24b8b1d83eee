"""Where the fused self-play step's time goes (developer tool, GPU box): the same kernel with and without its
write-out, and the write-out alone, each as 50 launches captured into a hipGraph (no host gaps), us per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch
import mnk_hip
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.random_rollout import RandomRollout

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


def main(m, n, k, N):
    env = TorchVectorMnkEnv(m, n, k, N, device=DEV)
    RandomRollout(env, seed=0).run(150, record=False)
    C = m * n
    acts = torch.zeros(N, dtype=torch.long, device=DEV)
    pend = torch.zeros(N, dtype=torch.bool, device=DEV)
    side_t = torch.zeros(N, dtype=torch.long, device=DEV)
    rew = torch.empty(N, dtype=torch.float32, device=DEV)
    term = torch.empty(N, dtype=torch.bool, device=DEV)
    obs = torch.empty((N, 2, m, n), dtype=torch.float32, device=DEV)
    mask = torch.empty((N, C), dtype=torch.bool, device=DEV)
    env.sample_legal_into(acts, seed=1, step=0)
    p = mnk_hip.ptr

    def step(o, mk):
        mnk_hip.call("mnk_selfplay_step_random", p(env._planes), p(env._meta), N, m, n, k, p(acts), p(pend), p(side_t),
                     None, 5, 7, None, 0, p(rew), p(term), p(o), mnk_hip.OBS_F32, p(mk), None, p(env._err), None, None, None,
                     0, env._stream())

    full = graph_time(lambda: step(obs, mask))
    logic = graph_time(lambda: step(None, None))
    mask_only = graph_time(lambda: step(None, mask))
    emit = graph_time(lambda: env.observe_into(obs, mask, flip_side=side_t, fix_empty_mask=True))
    emit_obs = graph_time(lambda: env.observe_into(obs, None))
    emit_mask = graph_time(lambda: env.observe_into(None, mask))
    nb = N * (2 * (16 * env.words + 4 + 8 * env.words + 4) + 9 * C + 8 + 4 + 1 + 2 + 16)
    print(f"{m}x{n}x{k} N={N}: fused step {full:.1f} us ({nb / full / 1e3:.0f} GB/s) | game logic only {logic:.1f} | "
          f"logic + mask {mask_only:.1f} | write-out alone: obs+mask {emit:.1f} ({N * 9 * C / emit / 1e3:.0f} GB/s), "
          f"obs {emit_obs:.1f}, mask {emit_mask:.1f}", flush=True)


if __name__ == "__main__":
    mnk_hip.load()
    main(9, 9, 5, 65536)
    main(9, 9, 5, 262144)
    main(19, 19, 5, 32768)
    main(19, 19, 5, 65536)

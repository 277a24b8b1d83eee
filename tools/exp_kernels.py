"""Per-kernel timings of the API-level kernels at a given shape (developer tool, GPU box).
Prints achieved algorithmic GB/s per SURVEY.md section 8d byte counts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch
import mnk_hip
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.random_rollout import RandomRollout, unpack_records
from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper
from selfplay.policy import RandomPolicy

DEV = "cuda:0"

def timeit(fn, reps=50, warm=5):
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

def report(name, us, nbytes, n):
    print(f"{name:34s} {us:8.1f} us  {nbytes/us/1e3:8.1f} GB/s  ({nbytes/n:6.0f} B/env)  {n/us*1e6:.3e} env/s", flush=True)

def main(m=9, n=9, k=5, N=65536):
    C, W = m * n, mnk_hip.state_words(m, n)
    S = 16 * W + 4
    env = TorchVectorMnkEnv(m, n, k, N, device=DEV)
    RandomRollout(env, seed=0).run(150, record=False)  # stationary position
    acts = torch.empty(N, dtype=torch.long, device=DEV)
    rew = torch.empty(N, dtype=torch.float32, device=DEV)
    done = torch.empty(N, dtype=torch.bool, device=DEV)
    mask = torch.empty((N, C), dtype=torch.bool, device=DEV)
    obs = torch.empty((N, 2, m, n), dtype=torch.float32, device=DEV)
    print(f"--- {m}x{n}x{k}, N={N}, W={W}")
    step = [0]
    def sample():
        env.sample_legal_into(acts, seed=1, step=step[0]); step[0] += 1
    report("sample_legal", timeit(sample), N * (S - 4 + 8), N)
    planes_bak, meta_bak = env._planes.clone(), env._meta.clone()
    def restore():
        env._planes.copy_(planes_bak); env._meta.copy_(meta_bak)
    def st(maskbuf, obsbuf):
        def f():
            env.step_into(acts, rew, done, maskbuf, obsbuf)
        return f
    sample()
    us_copy = timeit(restore)
    def with_restore(f):
        def g():
            restore(); f()
        return g
    report("step (no outputs)", timeit(with_restore(st(None, None))) - us_copy, N * (8 + S + 8 * W + 4 + 5), N)
    report("step + mask  [B_step]", timeit(with_restore(st(mask, None))) - us_copy, N * (8 + S + 8 * W + 4 + 5 + C), N)
    report("step + mask + obs", timeit(with_restore(st(mask, obs))) - us_copy, N * (8 + S + 8 * W + 4 + 5 + C + 8 * C), N)
    report("step_random + mask (one launch per ply)", timeit(with_restore(lambda: env.step_random_into(rew, done, mask, seed=1, step=7))) - us_copy,
           N * (S + 8 * W + 4 + 5 + C), N)
    report("observe (obs+mask)", timeit(lambda: env.observe_into(obs, mask)), N * (S - 4 + 9 * C), N)
    report("observe (mask only)", timeit(lambda: env.observe_into(None, mask)), N * (S - 4 + C), N)
    report("reset_mask", timeit(lambda: env.reset_mask_(done)), N * 1, N)
    # self-play wrapper
    wrap = TorchSelfPlayWrapper(env, seed=3)
    wrap.set_opponent(RandomPolicy(C))
    o, _ = wrap.reset()
    def agent_step():
        env.sample_legal_into(acts, seed=2, step=step[0]); step[0] += 1
        wrap.step(acts)
    us_s = timeit(sample)
    report("wrapper.step fused random opp", timeit(agent_step) - us_s, N * (2 * (S + 8 * W + 4) + 9 * C + 8 + 4 + 1 + 2 + 16), N)
    class Lowest:
        def act(self, o):
            return torch.argmax(o["action_mask"].to(torch.uint8), dim=1)
    wrap.set_opponent(Lowest())
    report("wrapper.step pre+argmax+post", timeit(agent_step) - us_s, N * (2 * (S + 8 * W + 4) + 17 * C + 15 + 32), N)
    # records -> rollout buffer
    T = 32
    rec = RandomRollout(env, seed=5).run(T)
    report(f"unpack_records T={T}", timeit(lambda: unpack_records(rec, env), reps=10), N * T * (S + 9 * C + 8 + 4 + 1), N * T)
    from selfplay.random_rollout import gae
    v = torch.randn(256, N, device=DEV); r = torch.randn(256, N, device=DEV); d = torch.rand(256, N, device=DEV) < 0.02
    lv = torch.randn(N, device=DEV)
    report("gae T=256", timeit(lambda: gae(r, v, d, lv), reps=10), N * 256 * (4 + 4 + 1 + 4 + 4), N * 256)
    lg = torch.randn(N, C, device=DEV)
    sm = RandomPolicy(C)._sampler
    report("sample_logits", timeit(lambda: sm.draw(lg, mask, False)), N * (5 * C + 8), N)
    lgb = lg.to(torch.bfloat16)
    report("sample_logits bf16", timeit(lambda: sm.draw(lgb, mask, False)), N * (3 * C + 8), N)
    report("sample uniform (mask only)", timeit(lambda: sm.draw(None, mask, False)), N * (C + 8), N)

if __name__ == "__main__":
    if len(sys.argv) > 1:  # exp_kernels.py 9x9x5 262144 ...
        for board, nenv in zip(sys.argv[1::2], sys.argv[2::2]):
            main(*(int(v) for v in board.split("x")), int(nenv))
    else:
        main()
        main(19, 19, 5, 32768)

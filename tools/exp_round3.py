"""Developer tool (GPU box): the round-3 measurements that are not part of bench.py.  Run bare for timings, or under
rocprofv3 (--kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE in separate passes) and condense with
tools/summarize_trace.py.

    python tools/exp_round3.py sink [envs] [steps]     the reference's rollout loop (policy -> wrapper.step -> buffer.add,
                                                       alg/ppo.py:93-108) eagerly, with the wrapper attached to the buffer
                                                       (`nosink`: without): which kernels run between the step kernels
    python tools/exp_round3.py gather [envs] [T] [B]   minibatch gather from a PackedRolloutBuffer (mnk_gather_obs)
    python tools/exp_round3.py narrow [envs]           the step / self-play kernels writing f32, bf16, u8 observations
    python tools/exp_round3.py replay [envs]           mnk_replay_actions of a 256-ply log (7-bit and byte formats)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch  # noqa: E402

import mnk_hip  # noqa: E402
from alg.packed_rollout_buffer import PackedRolloutBuffer  # noqa: E402
from alg.rollout_buffer import RolloutBuffer  # noqa: E402
from env.torch_vector_mnk_env import TorchVectorMnkEnv  # noqa: E402
from selfplay.policy import RandomPolicy  # noqa: E402
from selfplay.random_rollout import (ACT_BITS7, ACT_U8, GatheredLogs, RandomRollout, gather_start_state,  # noqa: E402
                                     replay_shard)
from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper  # noqa: E402

DEV = "cuda:0"
M, N_, K = 9, 9, 5
C = M * N_


def timeit(fn, reps=20, warm=3):
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


def sink(nenv=65536, steps=64, modes=(True,)):
    for attach in modes:
        env = TorchVectorMnkEnv(M, N_, K, nenv, device=DEV)
        wrap = TorchSelfPlayWrapper(env, seed=1)
        wrap.set_opponent(RandomPolicy(C, seed=2))
        buf = RolloutBuffer(steps, nenv, (2, M, N_), C, device=DEV)
        if attach:
            wrap.attach_sink(buf)
        agent = RandomPolicy(C, seed=3)
        values = torch.zeros(nenv, 1, device=DEV)
        logp = torch.zeros(nenv, device=DEV)
        obs, _ = wrap.reset()
        state = {"obs": obs}

        def rollout():
            buf.ptr = 0  # (not buffer.reset(): its in-place zero fill is once per learn(), not part of the step loop)
            obs = state["obs"]
            for _ in range(steps):
                actions = agent.act(obs)
                nxt, rew, term, trunc, _ = wrap.step(actions)
                buf.add(obs["observation"], actions, rew, values, logp, term | trunc, obs["action_mask"])
                obs = nxt
            state["obs"] = obs

        rollout()
        before = buf.copied_bytes
        us = timeit(rollout, reps=3, warm=1)
        per = (buf.copied_bytes - before) / (4 * steps * nenv)
        print(f"sink attached={attach}: {us / steps:8.2f} us per agent-step (eager, host-bound), buffer.add copied "
              f"{per:.1f} B per agent-step", flush=True)


def gather(nenv=65536, T=256, B=16384):
    buf = PackedRolloutBuffer(T, nenv, M, N_, device=DEV)
    env = TorchVectorMnkEnv(M, N_, K, nenv, device=DEV)
    roll = RandomRollout(env, seed=4)
    for t in range(T):  # any mid-game positions: the state planes of a running random rollout
        if t % 16 == 0:
            roll.run(5, record=False)
        buf.planes[t].copy_(env._planes)
    buf.ptr = T
    g = torch.Generator(device=DEV).manual_seed(0)
    for label, idx in (("random permutation (what get_data_loader draws)", torch.randperm(T * nenv, device=DEV, generator=g)[:B]),
                       ("the same samples sorted by (t, i)", None), ("contiguous samples", torch.arange(B, device=DEV))):
        if idx is None:
            idx = torch.sort(torch.randperm(T * nenv, device=DEV, generator=torch.Generator(device=DEV).manual_seed(0))[:B]).values
        for dt in (torch.float32, torch.bfloat16, torch.uint8):
            us = timeit(lambda: buf.gather(idx, obs_dtype=dt), reps=50, warm=5)
            out_bytes = B * (2 * C * torch.empty((), dtype=dt).element_size() + C)
            alg = B * (8 + 16 * buf.words) + out_bytes
            print(f"gather_obs B={B} from {T}x{nenv}, {label}, obs {str(dt)[6:]}: {us:7.2f} us  "
                  f"{alg / us / 1e3:7.0f} GB/s algorithmic ({alg / 1e6:.1f} MB)", flush=True)


def narrow(nenv=65536):
    for dt in (torch.float32, torch.bfloat16, torch.uint8):
        env = TorchVectorMnkEnv(M, N_, K, nenv, device=DEV, obs_dtype=dt)
        RandomRollout(env, seed=0).run(150, record=False)
        es = torch.empty((), dtype=dt).element_size()
        wrap = TorchSelfPlayWrapper(env, seed=1)
        wrap.set_opponent(RandomPolicy(C, seed=2))
        obs, _ = wrap.reset()
        acts = RandomPolicy(C, seed=3).act(obs)
        out = {"observation": torch.empty((nenv, 2, M, N_), dtype=dt, device=DEV),
               "action_mask": torch.empty((nenv, C), dtype=torch.bool, device=DEV),
               "rewards": torch.empty(nenv, device=DEV), "terminated": torch.empty(nenv, dtype=torch.bool, device=DEV)}
        # back-to-back launches replayed as one graph: the kernel's own time without host gaps
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(DEV)
        side.wait_stream(torch.cuda.current_stream(DEV))
        with torch.cuda.stream(side):
            for _ in range(3):
                wrap.step(acts, out=out)
        torch.cuda.current_stream(DEV).wait_stream(side)
        with torch.cuda.graph(g):
            for _ in range(50):
                wrap.step(acts, out=out)
        us = timeit(g.replay, reps=10, warm=2) / 50
        nbytes = nenv * (2 * 36 + 2 * C * es + C + 8 + 4 + 1 + 2 + 16)
        print(f"k_selfplay_step_random obs {str(dt)[6:]:8s}: {us:6.2f} us back to back  {nbytes / 1e6:5.1f} MB  "
              f"{nbytes / us / 1e3:6.0f} GB/s  ({nbytes / us / 1e3 / 8000:.2f} of 8 TB/s)", flush=True)
        obs_t = out["observation"]
        mask_t = out["action_mask"]
        rew, done = out["rewards"], out["terminated"]
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            env.step_into(acts, rew, done, mask_t, obs_t, autoreset=True)
        torch.cuda.current_stream(DEV).wait_stream(side)
        with torch.cuda.graph(g2):
            for _ in range(50):
                env.step_into(acts, rew, done, mask_t, obs_t, autoreset=True)
        us = timeit(g2.replay, reps=10, warm=2) / 50
        nbytes = nenv * (8 + 36 + 20 + 5 + C + 2 * C * es)
        print(f"k_step_full + mask + obs {str(dt)[6:]:8s}: {us:6.2f} us back to back  {nbytes / 1e6:5.1f} MB  "
              f"{nbytes / us / 1e3:6.0f} GB/s  ({nbytes / us / 1e3 / 8000:.2f} of 8 TB/s)", flush=True)


def replay(nenv=65536, T=256):
    for fmt in (ACT_BITS7, ACT_U8):
        env = TorchVectorMnkEnv(M, N_, K, nenv, device=DEV)
        roll = RandomRollout(env, seed=5)
        roll.run(256, record=False)
        state = gather_start_state(env)
        rec = roll.alloc(T, log_actions=fmt, with_state=False)
        roll.run(T, out=rec)
        logs = GatheredLogs.empty(1, 0, nenv, T, C, DEV, fmt=fmt, with_state=False)
        logs.msg.copy_(rec.msg.unsqueeze(0))
        out = replay_shard(logs, 0, M, N_, K, state=state)
        assert torch.equal(out.planes, rec.planes)
        err = torch.zeros(2, dtype=torch.int32, device=DEV)
        for record in (True, False):
            us = timeit(lambda: replay_shard(logs, 0, M, N_, K, err=err, out=out if record else None, state=state,
                                             record=record), reps=30, warm=5)
            nbytes = nenv * (T * ((28 if record else 0) + rec.msg.numel() * 8 / (nenv * T)) + 72)
            print(f"replay_actions fmt={fmt} records={record}: {us:7.2f} us per {T} plies  {nenv * T / us * 1e6:.3e} env-steps/s  "
                  f"{nbytes / us / 1e3:6.0f} GB/s", flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "sink"
    args = [int(v) for v in sys.argv[2:]]
    {"sink": sink, "nosink": lambda *a: sink(*a, modes=(False,)), "gather": gather, "narrow": narrow,
     "replay": replay}[mode](*args)

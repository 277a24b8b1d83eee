#!/usr/bin/env python3
"""bench.py -- random-policy rollout throughput of the MNK self-play path on MI355X.

Metric (BASELINE.json): env-steps/sec, 9x9x5, 65 536 parallel envs per GPU, random-policy
rollout.  A bench "step" is one pass of the hot path over the batch = ONE launch of the fused
kernel ``mnk_rollout_random``: ``--chunk`` plies (default 256 = the reference's n_steps) on every
env -- per ply: uniform legal move (RandomPolicy), place stone, 4-direction win scan, reward/done,
restart of finished games, packed record written to HBM.  ``value`` is env-steps (plies x envs)
per second.  With --gpus N > 1 the env axis is sharded (65 536 envs per
rank, global env ids key the RNG) and every chunk is all-gathered over RCCL on a side stream
while the next chunk runs -- by default as a keyframed action log: 7 bits per action at 9x9 (0.875 B
per env-step) plus, in every 8th chunk's message (--keyframe), the chunk-start state (36 B per env):
0.893 B per env-step; mnk_replay_actions rebuilds any chunk's full records on demand from the last
keyframe + the logs since.  --keyframe 1 makes every message self-contained, --keyframe 0 sends the
state once before the first chunk (receivers must then replay every chunk to keep it current);
--gather records all-gathers the 28 B packed records themselves.

    python bench.py                       # 1 GPU, defaults finish in well under a minute
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8

Rank 0 prints ONE JSON line (driver contract) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# multi-process GPU work on this pool needs dmabuf IPC; must be in the environment before HIP initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

# HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/
# (FETCH_SIZE x 1024 x 2 [gfx950 correction] + WRITE_SIZE x 1024), keyed by (board, envs/GPU, chunk).
# bench.py cannot run rocprofv3 on itself; configurations without a committed profile report null.
PMC_TRAFFIC = {
    ("9x9x5", 65536, 256): (474.71e6, "profiles/r04_rollout_9x9x5.md"),
    ("19x19x5", 32768, 256): (845.87e6, "profiles/r03_rollout_19x19x5_32768.md"),   # two lanes per env, words split
    ("12x12x5", 65536, 256): (745.25e6, "profiles/r02_rollout_12x12x5_jit.md"),     # run-time specialised kernel
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64, help="timed launches of the rollout kernel (chunk plies each)")
    ap.add_argument("--warmup", type=int, default=64, help="untimed launches right before the timed region")
    ap.add_argument("--settle", type=int, default=512,
                    help="untimed set-up launches before the warm-up: the boards reach their stationary fill after ~1 "
                         "launch, but the GPU reaches its sustained clock only after ~50 ms of load (measured: the first "
                         "timed region reads 6 %% low after 64 launches = 8 ms, and level after 512)")
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--board", type=str, default="9x9x5")
    ap.add_argument("--chunk", type=int, default=256,
                    help="plies per launch of the fused rollout kernel (256 = the reference's n_steps, train.py:246)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="0 = every CPU this process may use (affinity mask and cgroup quota)")
    ap.add_argument("--no-api-path", action="store_true", help="skip the extra per-step-launch measurement")
    ap.add_argument("--mode", choices=("rollout", "selfplay"), default="rollout",
                    help="rollout = the BASELINE.json metric (default); selfplay = BASELINE config 3: the wrapper loop "
                         "with a small conv policy as agent and an opponent pool (agent-steps/s, NN time dominates)")
    ap.add_argument("--backend", default="nccl", help="process-group backend; nccl = RCCL.  gloo is for rehearsing "
                    "the multi-rank path on a one-GPU box (all ranks then share device 0)")
    ap.add_argument("--gather", choices=("actions", "records", "none"), default="actions",
                    help="what ranks all-gather per chunk when --gpus > 1 (ignored on one GPU): the action log (see "
                         "--keyframe) or the packed records")
    ap.add_argument("--keyframe", type=int, default=8,
                    help="--gather actions: every K-th chunk's message also carries the chunk-start state (a keyframe), so "
                         "any chunk's records can be rebuilt on demand from at most K messages; 1 = every message is "
                         "self-contained, 0 = the state is gathered once before the first chunk and never again")
    ap.add_argument("--allgather", choices=("auto", "rccl", "direct"), default="auto",
                    help="form of the exchange step through the C ABI: rccl = ncclAllGather (RCCL picks the algorithm), "
                         "direct = one grouped ncclSend + ncclRecv per peer (mnk_allgather_records_direct): every message "
                         "once over each of the rank's own xGMI links.  Whichever runs in the timed loop, the line also "
                         "times both forms alone (exchange.alone).  auto (default): the timed loop runs with ncclAllGather; if "
                         "the direct form alone is at least 10 %% faster on the slowest rank, warm-up and timed region are "
                         "run again with it and the better of the two is `value` (the other one stays in the line)")
    ap.add_argument("--rehearse-exchange", action="store_true",
                    help="on ONE GPU: run the multi-GPU code path anyway -- a process group of one rank, the C-ABI "
                         "communicator, the exchange step on the side stream, both exchange forms alone -- so that every "
                         "line of it has run on hardware before the driver's multi-GPU run (its numbers mean nothing)")
    ap.add_argument("--exchange-every", type=int, default=1, metavar="J",
                    help="--gpus > 1: ONE all-gather per J chunks carrying the J chunks' messages end to end (J times the "
                         "bytes per call): amortises RCCL's per-call cost and moves the message up the xGMI size curve; "
                         "two runs with different J give the inbound rate at two message sizes.  --steps, --warmup and "
                         "--settle must be multiples of J")
    ap.add_argument("--alone-limit", type=float, default=60.0,
                    help="seconds an exchange form timed alone (exchange.alone) may take before it is declared hung: the line "
                         "is printed with `exchange_hung` and every rank leaves with exit code 4")
    ap.add_argument("--phase-timeout", type=float, default=300.0,
                    help="multi-rank runs: a phase of the bench (set-up, warm-up, timed region, ...) that takes longer than "
                         "this many seconds is taken to be a hung collective: the rank says which phase on stderr and "
                         "leaves with exit code 5 instead of sitting there until the driver's limit (0 = off)")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="if the C-ABI RCCL communicator (mnk_comm_*) cannot be created, run the same all-gather through "
                         "torch.distributed instead of exiting non-zero")
    args = ap.parse_args()
    if args.exchange_every < 1:
        ap.error("--exchange-every must be at least 1")
    if args.exchange_every > 1 and any(v % args.exchange_every for v in (args.steps, args.warmup, args.settle)):
        ap.error("--steps, --warmup and --settle must be multiples of --exchange-every")
    return args


EXIT_COMM, EXIT_HUNG, EXIT_PHASE = 3, 4, 5  # no C-ABI communicator on every rank / an exchange form never finished / a phase timed out


class PhaseWatchdog:
    """A thread that ends the process (exit code EXIT_PHASE, the phase named on stderr) when one phase of a multi-rank
    run outlasts ``limit_s``: a collective that never completes blocks its rank inside a HIP or RCCL call where no
    Python-level timeout can reach it -- ctypes and torch release the GIL there, so this thread still runs."""

    def __init__(self, limit_s, rank, on_fire=None):
        import threading

        self.limit_s, self.rank, self.on_fire = limit_s, rank, on_fire
        self.phase, self.since = "start", time.monotonic()
        self._stop = threading.Event()
        if limit_s > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def enter(self, phase):
        self.phase, self.since = phase, time.monotonic()

    def stop(self):
        self._stop.set()

    def _run(self):
        while not self._stop.wait(1.0):
            waited = time.monotonic() - self.since
            if waited > self.limit_s:
                print(f"[bench rank {self.rank}] phase '{self.phase}' has not finished after {waited:.0f} s: taken to be a "
                      f"hung collective; leaving with exit code {EXIT_PHASE}", file=sys.stderr, flush=True)
                if self.on_fire is not None:
                    try:
                        self.on_fire(self.phase)
                    except Exception:  # noqa: BLE001 -- nothing may keep the process alive here
                        pass
                os._exit(EXIT_PHASE)


def agree_any(store, tag, rank, world, mine, wait_s):
    """True when ANY rank says ``mine``: the ranks' common verdict WITHOUT a GPU collective (the process group's TCP store),
    for decisions taken exactly when the GPU collectives cannot be trusted -- a rank that timed out on an exchange form
    must not be waited for in an all-reduce by a rank that finished just inside the limit.  A rank that does not post its
    verdict within ``wait_s`` counts as hung."""
    from datetime import timedelta

    store.set(f"mnk_bench/{tag}/{rank}", "1" if mine else "0")
    verdict = bool(mine)
    for r in range(world):
        key = f"mnk_bench/{tag}/{r}"
        try:
            store.wait([key], timedelta(seconds=wait_s))
            verdict = verdict or store.get(key) == b"1"
        except Exception:  # noqa: BLE001 -- timeout (RuntimeError / DistStoreError by version): that rank never got there
            verdict = True
    return verdict


def record_bytes(rows):
    """Algorithmic HBM bytes of one env-step inside the fused rollout:
    the packed record R = 8 * rows + 4 that is written for every ply (rows = mnk_record_words: the 32-bit
    words of both planes, interleaved, no padding); the state itself lives in registers for the whole launch
    (its load/store is amortised over the chunk)."""
    return 8 * rows + 4


def state_bytes(words):
    return 16 * words + 4


def host_cpu_identity():
    """CPU model, logical CPUs of the host, and the CPUs this process may actually use (affinity mask and
    cgroup quota: a GPU box hands one GPU's job a share of the host, not all of it)."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    logical = os.cpu_count() or 1
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else logical
    try:  # cgroup v2 quota: "max 100000" or "<quota> <period>"
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
            if quota != "max":
                usable = max(1, min(usable, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return model, logical, usable


def cpu_baseline(env, seconds, threads=0):
    """The oracle's torch-eager restatement of the reference env (conv2d win scan) and of
    RandomPolicy (multinomial), timed on the host cores from the GPU env's current (stationary)
    position.  Test infrastructure used as the reported CPU baseline -- never the product path."""
    from oracle.env_torch import OracleVectorEnv
    from oracle.policies import OracleRandomPolicy

    n = env.num_envs
    model, logical, usable = host_cpu_identity()
    threads = threads or usable  # every core this process may use
    torch.set_num_threads(threads)
    ora = OracleVectorEnv(env.m, env.n, env.k, n)
    ora.boards.copy_(env.boards.cpu())
    ora.current_player.copy_(env.current_player.cpu())
    ora.move_counts.copy_(env.move_counts.cpu())
    pol = OracleRandomPolicy(env.m * env.n)
    torch.manual_seed(0)
    obs = ora.observe()

    def one(obs):
        a = pol.act(obs)
        obs, _, d = ora.step(a)
        if bool(d.any()):
            ora.reset(torch.nonzero(d).squeeze(1))
            obs = ora.observe()
        return obs

    for _ in range(2):
        obs = one(obs)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        obs = one(obs)
        steps += 1
    dt = time.perf_counter() - t0
    # one-thread figure, a short sample
    torch.set_num_threads(1)
    steps1, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < min(6.0, seconds):
        obs = one(obs)
        steps1 += 1
    dt1 = time.perf_counter() - t1
    torch.set_num_threads(threads)
    return {
        "value": steps * n / dt,
        "single_thread_value": steps1 * n / dt1,
        "unit": "env-steps/s",
        "cores": torch.get_num_threads(),
        "cpu_model": model,
        "host_logical_cpus": logical,
        "usable_cpus": usable,
        "kind": "port",
        "sample": f"{steps} plies x {n} envs ({env.m}x{env.n}x{env.k}) from the GPU env's stationary position, "
                  f"{dt:.1f} s; oracle/env_torch.py (torch eager, conv2d win scan) + multinomial RandomPolicy",
    }


def api_path_rate(env, seed, steps=200):
    """The same workload through the API-level kernels, one launch each per ply:
    mnk_sample_legal -> mnk_step (rewards, dones, legal mask) -> mnk_reset_mask(done).
    Launch-latency bound at this batch size; reported for transparency only."""
    n, dev = env.num_envs, env._dev
    acts = torch.empty(n, dtype=torch.long, device=dev)
    rew = torch.empty(n, dtype=torch.float32, device=dev)
    done = torch.empty(n, dtype=torch.bool, device=dev)
    mask = torch.empty((n, env.max_moves), dtype=torch.bool, device=dev)

    def ply(t):
        env.sample_legal_into(acts, seed=seed, step=t)
        env.step_into(acts, rew, done, mask)
        env.reset_mask_(done)

    for t in range(20):
        ply(t)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for t in range(steps):
        ply(1000 + t)
    torch.cuda.synchronize(dev)
    return steps * n / (time.perf_counter() - t0)


def api_path_graphed_rate(env, seed, plies_per_graph=64, replays=8, fused_reset=False):
    """BASELINE config 2 as the API would be driven in production: the same three launches per ply
    (mnk_sample_legal -> mnk_step with the legal mask -> mnk_reset_mask), ``plies_per_graph`` plies captured
    once into a hipGraph (torch.cuda.graph) and replayed -- the Philox step counter advances in device memory
    (``step_dev``), so every replay plays new plies.  No host launch cost per ply: what is left is the kernels.
    ``fused_reset``: two launches per ply, the reset of finished games folded into mnk_step (MNK_STEP_AUTORESET)."""
    n, dev = env.num_envs, env._dev
    acts = torch.empty(n, dtype=torch.long, device=dev)
    rew = torch.empty(n, dtype=torch.float32, device=dev)
    done = torch.empty(n, dtype=torch.bool, device=dev)
    mask = torch.empty((n, env.max_moves), dtype=torch.bool, device=dev)
    step_dev = torch.full((1,), 1 << 20, dtype=torch.int64, device=dev)  # away from the eager path's counters

    def body():
        for t in range(plies_per_graph):
            env.sample_legal_into(acts, seed=seed, step=t, step_dev=step_dev)
            env.step_into(acts, rew, done, mask, autoreset=fused_reset)
            if not fused_reset:
                env.reset_mask_(done)
        step_dev.add_(plies_per_graph)

    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream(dev).wait_stream(side)
    env.specialise_kernels()  # a board without a built-in variant: its own kernels go into the graph (nothing compiles under a capture)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        body()
    graph.replay()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(replays):
        graph.replay()
    torch.cuda.synchronize(dev)
    return replays * plies_per_graph * n / (time.perf_counter() - t0)


def api_path_one_launch_rate(env, seed, plies_per_graph=64, replays=8):
    """BASELINE config 2 with ONE launch per ply: mnk_step_random = the lane's own Philox draw + place stone + win scan
    + reward / done + restart of a finished game + legal mask of the position that follows (policy.py:18-29 ->
    env:55-84 -> env:34-44 -> env:46-53), ``plies_per_graph`` plies captured into a hipGraph and replayed.  Also
    returns the per-ply time of the same graph on a 64-env batch: the floor one dependent launch costs here."""
    dev = env._dev

    def graphed(e):
        n = e.num_envs
        rew = torch.empty(n, dtype=torch.float32, device=dev)
        done = torch.empty(n, dtype=torch.bool, device=dev)
        mask = torch.empty((n, e.max_moves), dtype=torch.bool, device=dev)
        step_dev = torch.full((1,), 1 << 22, dtype=torch.int64, device=dev)

        def body():
            for t in range(plies_per_graph):
                e.step_random_into(rew, done, mask, seed=seed, step=t, step_dev=step_dev, autoreset=True)
            step_dev.add_(plies_per_graph)

        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            body()
        torch.cuda.current_stream(dev).wait_stream(side)
        e.specialise_kernels()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body()
        graph.replay()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(replays):
            graph.replay()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / (replays * plies_per_graph)

    per_ply = graphed(env)
    from env.torch_vector_mnk_env import TorchVectorMnkEnv

    floor = graphed(TorchVectorMnkEnv(env.m, env.n, env.k, 64, device=str(dev)))
    return env.num_envs / per_ply, per_ply * 1e6, floor * 1e6


def write_ceiling_GBps(dev, nenv, chunk, rows, seconds=0.05):
    """What the device sustains for the store pattern the rollout kernel is bound by: a write-only kernel that fills
    rec[t][row][N] the way mnk_rollout_random does (mnk_probe_record_writes), timed with HIP events for ~50 ms."""
    import mnk_hip

    rec = torch.empty((chunk, rows, nenv), dtype=torch.int64, device=dev)
    stream = mnk_hip.stream_ptr(dev)

    def launch(k):
        for _ in range(k):
            mnk_hip.call("mnk_probe_record_writes", mnk_hip.ptr(rec), nenv, chunk, rows, stream)

    launch(8)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch(4)
    e1.record()
    torch.cuda.synchronize(dev)
    per = e0.elapsed_time(e1) * 1e-3 / 4
    k = max(4, int(seconds / max(per, 1e-6)))
    e0.record()
    launch(k)
    e1.record()
    torch.cuda.synchronize(dev)
    return rec.numel() * 8 * k / (e0.elapsed_time(e1) * 1e-3) / 1e9


def gpu_identity(dev):
    """Identity of the device the numbers were taken on: torch's device properties plus what `rocminfo` (a child
    process) says about the first gfx950 agent -- marketing name (empty on this pool's image), max clock, CUs."""
    import re
    import subprocess

    p = torch.cuda.get_device_properties(dev)
    out = {"name": p.name, "arch": getattr(p, "gcnArchName", None), "compute_units": p.multi_processor_count,
           "memory_GiB": round(p.total_memory / 2 ** 30, 1), "hip": torch.version.hip, "rocminfo": None}
    try:
        text = subprocess.run(["rocminfo"], capture_output=True, text=True, timeout=20).stdout
        for block in text.split("*******")[1:]:
            if re.search(r"^\s*Name:\s+gfx950", block, flags=re.M):
                def field(label):
                    m = re.search(r"^[ \t]*" + label + r":[ \t]*(.*?)[ \t]*$", block, flags=re.M)
                    return m.group(1) if m else None
                out["rocminfo"] = {"name": field("Name"), "marketing_name": field("Marketing Name"),
                                   "max_clock_MHz": field(r"Max Clock Freq\. \(MHz\)"), "compute_units": field("Compute Unit")}
                break
    except Exception as e:  # noqa: BLE001 -- identity is best effort
        out["rocminfo"] = f"unavailable: {e}"
    return out


def selfplay_object(m, n, k, nenv, seed, dev, steps=64):
    """BASELINE config 3's env side in the driver-run line: the TorchSelfPlayWrapper loop with the uniformly random
    agent and the built-in random opponent, every step written straight into a RolloutBuffer (the fused sink).
    Eager = the reference's loop shape from Python (alg/ppo.py:93-108: policy, step, buffer.add); graphed = the same
    ``steps`` agent-steps as one hipGraph whose nodes write the buffer's rows (selfplay/graphed.py GraphedRollout)."""
    from alg.rollout_buffer import RolloutBuffer
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay.graphed import GraphedRollout
    from selfplay.policy import RandomPolicy
    from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper

    c = m * n
    out = {"what": f"{m}x{n}x{k}, {nenv} envs, TorchSelfPlayWrapper loop, uniformly random agent and opponent, "
                   f"{steps} agent-steps per rollout written in place into a RolloutBuffer (observation f32 + mask + "
                   "rewards + dones + actions + log-probs)"}
    for sink in (True, False):
        env = TorchVectorMnkEnv(m, n, k, nenv, device=str(dev))
        wrap = TorchSelfPlayWrapper(env, seed=seed)
        wrap.set_opponent(RandomPolicy(c, seed=seed + 1))
        buf = RolloutBuffer(steps, nenv, (2, m, n), c, device=str(dev))
        if sink:
            wrap.attach_sink(buf)
        agent = RandomPolicy(c, seed=seed + 2)
        zeros = torch.zeros(nenv, device=dev)
        obs, _ = wrap.reset()

        def rollout():
            nonlocal obs
            buf.reset()
            for _ in range(steps):
                actions = agent.act(obs)
                nxt, rew, term, trunc, _ = wrap.step(actions)
                buf.add(obs["observation"], actions, rew, zeros, zeros, term | trunc, obs["action_mask"])
                obs = nxt

        rollout()
        torch.cuda.synchronize(dev)
        before = buf.copied_bytes
        t0 = time.perf_counter()
        for _ in range(3):
            rollout()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / 3
        key = "eager_sink" if sink else "eager_copying"
        out[key + "_agent_steps_per_s"] = nenv * steps / dt
        out[key + "_add_copied_bytes_per_agent_step"] = (buf.copied_bytes - before) / (3 * steps * nenv)
        del buf, wrap, env
    env = TorchVectorMnkEnv(m, n, k, nenv, device=str(dev))
    wrap = TorchSelfPlayWrapper(env, seed=seed)
    wrap.set_opponent(RandomPolicy(c, seed=seed + 1))
    buf = RolloutBuffer(steps, nenv, (2, m, n), c, device=str(dev))
    roll = GraphedRollout(wrap, buf, None, seed=seed + 2)
    roll.run()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 5
    for _ in range(reps):
        roll.run()
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / (reps * steps)
    # algorithmic bytes of one agent-step written in place: state round trip, next observation f32 (8C) + mask (C),
    # the mask read by the draw (C), action 8 + log-prob 4 + reward 4 + done 1 + side / pending / flags
    words = env.words
    alg = 2 * (16 * words + 4) + 8 * c + c + c + 8 + 4 + 4 + 1 + 8 + 1
    out.update({
        "graphed_agent_steps_per_s": nenv / (ms * 1e-3),
        "graphed_env_side_us_per_agent_step": ms * 1e3,
        "launches_per_agent_step": 1,  # round 4: the agent's draw is folded into the step kernel (mnk_selfplay_step_random_logits)
        "alg_bytes_per_agent_step": alg,
        "achieved_GBps": alg * nenv / (ms * 1e-3) / 1e9,
        "frac_of_hbm_peak": alg * nenv / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "survey_B_agent_bytes": 2 * (16 * words + 4 + 8 * words + 4) + 17 * c + 15,
    })
    return out


def train_cadence(nenv, steps=64, rollouts=12, m=9, n=9, k=5, dev="cuda:0"):
    """The rollout loop at the REFERENCE's cadence: a new opponent -- a deepcopy of the agent's network -- before every
    rollout (train.py:106-114), `steps` agent-steps per rollout, network agent against network opponent (a small
    BatchNorm-free conv policy: 3 x conv3x3(32) + 1x1 heads).  us per agent-step of
      (a) the reference-shaped eager loop: net -> Categorical.sample / log_prob -> wrapper.step -> buffer.add
          (alg/ppo.py:93-108), set_opponent(FusedNNPolicy(deepcopy(net))) before each rollout, the sink attached;
      (b) ONE captured GraphedRollout whose opponent is swapped in place (set_opponent_weights: no capture);
      (c) a recapture before every rollout (what round 3 needed: an eager warm-up rollout + a capture)."""
    import copy

    import torch.nn as nn

    from alg.rollout_buffer import RolloutBuffer
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay.graphed import GraphedRollout
    from selfplay.policy import FusedNNPolicy
    from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper

    c = m * n

    class ConvNet(nn.Module):
        def __init__(self, width=32):
            super().__init__()
            self.body = nn.Sequential(nn.Conv2d(2, width, 3, padding=1), nn.ReLU(), nn.Conv2d(width, width, 3, padding=1),
                                      nn.ReLU(), nn.Conv2d(width, width, 3, padding=1), nn.ReLU())
            self.pi = nn.Sequential(nn.Conv2d(width, 2, 1), nn.Flatten(), nn.Linear(2 * c, c))
            self.v = nn.Sequential(nn.Conv2d(width, 1, 1), nn.Flatten(), nn.Linear(c, 1), nn.Tanh())

        def forward(self, obs, action_mask=None):
            f = self.body(obs)
            logits = self.pi(f)
            if action_mask is not None:
                logits = torch.where(action_mask.bool(), logits, torch.full_like(logits, -torch.inf))
            return torch.distributions.Categorical(logits=logits, validate_args=False), self.v(f)

    torch.manual_seed(0)
    net = ConvNet().to(dev).eval()

    def make():
        w = TorchSelfPlayWrapper(TorchVectorMnkEnv(m, n, k, nenv, device=dev), seed=1)
        w.set_opponent(FusedNNPolicy(copy.deepcopy(net), seed=2))
        return w

    w = make()
    buf = RolloutBuffer(steps, nenv, (2, m, n), c, device=dev)
    w.attach_sink(buf)
    obs, _ = w.reset()
    state = {"obs": obs}

    def eager_rollout(j):
        w.set_opponent(FusedNNPolicy(copy.deepcopy(net), seed=100 + j))  # train.py:110-114
        for _ in range(steps):
            o = state["obs"]
            with torch.no_grad():
                dist, values = net(o["observation"], o["action_mask"])
                a = dist.sample()
                lp = dist.log_prob(a)
            nxt, r, term, trunc, _ = w.step(a)
            buf.add(o["observation"], a, r, values, lp, term | trunc, o["action_mask"])
            state["obs"] = nxt
        buf.reset()  # (after the rollout, as ppo.py:148: the carried-over observation sits in the spill row)

    def wall(fn):
        fn(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(rollouts):
            fn(1 + j)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (rollouts * steps) * 1e6

    t_eager = wall(eager_rollout)
    buf2 = RolloutBuffer(steps, nenv, (2, m, n), c, device=dev)
    roll = GraphedRollout(make(), buf2, net, seed=3)
    graph = roll.graph

    def swap_rollout(j):
        roll.set_opponent_weights(net, seed=100 + j)
        roll.run()
        buf2.reset()

    t_swap = wall(swap_rollout)
    assert roll.graph is graph  # no capture happened

    def recapture_rollout(j):
        roll.wrapper.set_opponent(FusedNNPolicy(copy.deepcopy(net), seed=100 + j))
        roll.recapture()  # (one real rollout + a capture)
        buf2.reset()

    t_recap = wall(recapture_rollout)
    return t_eager, t_swap, t_recap


def with_action_log_rate(roll, chunk, steps):
    """The kernel variant the multi-GPU runs use by default (--gather actions: records AND the action log written),
    timed on one GPU like the headline (K launches, HIP events): the driver's N = 1 scaling point runs the
    records-only variant (no exchange on one GPU), this is the per-GPU compute rate its N > 1 points start from."""
    chunk = max(4, chunk - chunk % 4)
    rec = roll.alloc(chunk, log_actions=True, with_state=False)
    roll.step += (-roll.step) % 4  # a log starts on a multiple of four plies (the Philox block boundary)
    for _ in range(512):  # as the headline's --settle: the device reaches its sustained clock after ~50 ms of load
        roll.run(chunk, out=rec)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        roll.run(chunk, out=rec)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    n = roll.env.num_envs
    return {"value": n * chunk / (ms * 1e-3), "unit": "env-steps/s", "avg_launch_us": ms * 1e3,
            "kernel_variant": "records + action log " + {1: "(one byte per action)", 2: "(two bytes per action)",
                                                         3: "(7-bit stream)", 4: "(a byte and a bit per action)"}[rec.fmt],
            "log_bytes_per_env_step": rec.msg.numel() * 8 / (n * chunk)}


def replay_rate(env, roll, chunk, reps=8):
    """Receiving side of the multi-GPU exchange: rebuild one shard's full records from its
    chunk-start state + action log (mnk_replay_actions); reported for transparency."""
    from selfplay.random_rollout import GatheredLogs, gather_start_state, replay_shard

    state = gather_start_state(env)  # one rank: a copy of the env state, advanced by the replays below
    rec = roll.alloc(chunk, log_actions=True, with_state=False)  # the log alone, in the board's most compact format
    roll.run(chunk, out=rec)
    logs = GatheredLogs.empty(1, 0, env.num_envs, chunk, env.max_moves, env._dev, fmt=rec.fmt, with_state=False)
    logs.msg.copy_(rec.msg.unsqueeze(0))
    out = replay_shard(logs, 0, env.m, env.n, env.k, state=state)
    assert torch.equal(out.planes, rec.planes) and torch.equal(out.meta, rec.meta), "replay != records"
    assert torch.equal(state.planes[0], env._planes), "replay state != sender's state"
    err = torch.zeros(2, dtype=torch.int32, device=env._dev)
    # the timed replays re-play the same log from wherever the state stands: the same work per ply
    replay_shard(logs, 0, env.m, env.n, env.k, err=err, out=out, state=state)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        replay_shard(logs, 0, env.m, env.n, env.k, err=err, out=out, state=state)
    torch.cuda.synchronize()
    return reps * chunk * env.num_envs / (time.perf_counter() - t0)


def selfplay_mode(args):
    print(json.dumps(selfplay_config3(args, min(args.steps * 4, 256), 64)), flush=True)


def selfplay_config3(args, steps, warm):
    """BASELINE.json config 3: full self-play rollout through TorchSelfPlayWrapper with a small conv
    policy (the shape of the reference's cnn_b_s: 4 x [conv3x3(56) + BN + ReLU], 1x1-conv heads,
    alg/architectures/configs.py:49-56) as agent and a pool of 4 frozen opponents rotated every 64 steps.
    The networks are the caller's side of the boundary (PyTorch-ROCm / MIOpen, bf16 autocast, eval mode);
    what this build contributes are the two env kernels and the fused mask+softmax+draw per step.  Every agent-step is
    stored: the wrapper is attached to a RolloutBuffer of 64 rows (the fused sink) and observations leave the kernels as
    bf16 -- the dtype the networks' first convolution computes in under autocast -- so nothing is copied or cast on the
    way from the step kernel to the buffer and to the network."""
    import torch.nn as nn

    import mnk_hip
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay.opponent_pool import OpponentPool
    from selfplay.policy import FusedNNPolicy
    from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper

    mnk_hip.load()
    dev = torch.device("cuda", 0)
    m, n, k = (int(v) for v in args.board.split("x"))
    c, nenv = m * n, args.envs

    class ConvPolicy(nn.Module):
        def __init__(self, width=56, hidden=128):
            super().__init__()
            layers, cin = [], 2
            for _ in range(4):
                layers += [nn.Conv2d(cin, width, 3, padding=1), nn.BatchNorm2d(width), nn.ReLU()]
                cin = width
            self.body = nn.Sequential(*layers)
            self.folded = False

            def head(ch, out, squash):
                mods = [nn.Conv2d(width, ch, 1), nn.Flatten(), nn.LayerNorm(ch * c), nn.ReLU(),
                        nn.Linear(ch * c, hidden), nn.LayerNorm(hidden), nn.ReLU(), nn.Linear(hidden, out)]
                return nn.Sequential(*(mods + ([nn.Tanh()] if squash else [])))

            self.actor, self.critic = head(2, c, False), head(1, 1, True)

        def fold_batchnorm(self):
            """eval-mode inference: BatchNorm folded into the preceding conv (MIOpen's NHWC bf16 BatchNorm
            inference kernel alone takes 135 ms at this batch -- 17x the four convolutions)"""
            mods = list(self.body)
            out = []
            for a, b in zip(mods, mods[1:] + [None]):
                if isinstance(a, nn.Conv2d) and isinstance(b, nn.BatchNorm2d):
                    out.append(nn.utils.fusion.fuse_conv_bn_eval(a, b))
                elif not isinstance(a, nn.BatchNorm2d):
                    out.append(a)
            self.body = nn.Sequential(*out)
            self.folded = True
            return self

        def forward(self, obs, action_mask=None):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                f = self.body(obs.contiguous(memory_format=torch.channels_last))
                logits, value = self.actor(f).float(), self.critic(f).float()
            if action_mask is not None:
                logits = torch.where(action_mask, logits, torch.full_like(logits, -torch.inf))
            return torch.distributions.Categorical(logits=logits, validate_args=False), value

    torch.manual_seed(args.seed)
    nets = [ConvPolicy().to(dev).eval().fold_batchnorm().to(memory_format=torch.channels_last) for _ in range(5)]
    agent = FusedNNPolicy(nets[0], seed=args.seed)
    pool = OpponentPool(max_size=4)
    for net in nets[1:]:
        pool.add_opponent(FusedNNPolicy(net, seed=args.seed + 1))
    from alg.rollout_buffer import RolloutBuffer

    env = TorchVectorMnkEnv(m, n, k, nenv, device=str(dev), obs_dtype=torch.bfloat16)
    wrap = TorchSelfPlayWrapper(env, seed=args.seed)
    opponents = list(pool.pool)
    wrap.set_opponent(opponents[0])
    buf = RolloutBuffer(64, nenv, (2, m, n), c, device=str(dev), obs_dtype=torch.bfloat16)
    wrap.attach_sink(buf)
    obs, _ = wrap.reset()
    state = {"obs": obs, "plies": torch.zeros((), dtype=torch.long, device=dev)}
    zeros = torch.zeros(nenv, device=dev)

    def step(t):
        if t % 64 == 0:
            wrap.set_opponent(opponents[(t // 64) % len(opponents)])
        if buf.ptr == buf.n_steps:
            buf.ptr = 0  # the next rollout overwrites the rows (values / advantages are not part of this measurement)
        before = env._meta >> 1
        prev = state["obs"]
        # the agent's forward is the caller's; its mask + softmax + draw happen inside the step kernel, and so do the
        # opponent's (FusedNNPolicy): two env-side launches per agent-step
        state["obs"], rew, term, trunc, info = wrap.step_logits(agent.logits(prev), prev["action_mask"], agent._sampler)
        buf.add(prev["observation"], info["actions"], rew, zeros, info["log_probs"], term | trunc, prev["action_mask"])
        # plies played this step = growth of the move counters (resets restart them at 0 or 1)
        state["plies"] += torch.clamp((env._meta >> 1) - before, min=0).sum()

    # (steps / warm: agent-steps here, not kernel launches)
    for t in range(warm):
        step(t)
    state["plies"].zero_()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for t in range(steps):
        step(warm + t)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    # time of the env-side work alone: the same kernels with a constant-action opponent and agent
    class Const:
        def act(self, o):
            return torch.zeros(nenv, dtype=torch.long, device=dev)
    wrap.set_opponent(Const())
    wrap.attach_sink(None)
    acts = torch.zeros(nenv, dtype=torch.long, device=dev)
    for _ in range(5):
        wrap.step(acts)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(50):
        wrap.step(acts)
    torch.cuda.synchronize(dev)
    env_ms = (time.perf_counter() - t1) * 1e3 / 50
    out = {
        "metric": f"agent-steps/sec {m}x{n}x{k}, {nenv} envs, self-play rollout with conv policy + opponent pool",
        "value": nenv * steps / dt, "unit": "agent-steps/s", "n_gpus": 1, "steps": steps, "warmup": warm,
        "ms_per_step": dt * 1e3 / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64 env state; bf16 autocast policy (caller side)", "data": "synthetic",
        "config": {"workload": f"{m}x{n}x{k}, {nenv} envs, TorchSelfPlayWrapper loop, agent = opponent pool of 4 = "
                               "random-init 4x conv3x3(56) policies, fused mask+softmax+draw, bf16 observations written "
                               "straight into a RolloutBuffer (fused sink)", "envs_per_gpu": nenv},
        "buffer_add_copied_bytes_per_agent_step": buf.copied_bytes / ((steps + warm) * nenv),
        "env_steps_per_s": float(state["plies"].item()) / dt,
        "env_side_ms_per_step": env_ms,
        "nn_and_sampling_ms_per_step": dt * 1e3 / steps - env_ms,
        "roofline": {"note": "not a kernel of this path: 99.7 % of the step is the caller-side network forward "
                             "(PyTorch-ROCm / MIOpen), which SURVEY.md section 8 puts out of scope; the env-side "
                             "kernels of this step are timed in profiles/ (api kernels)", "bound": None,
                     "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None},
        "cpu_baseline": {"note": "the CPU baseline is reported on the headline workload (default --mode rollout)",
                         "value": None, "unit": "agent-steps/s", "cores": None, "kind": None, "sample": None},
    }
    return out


@contextlib.contextmanager
def c_stdout_to_stderr():
    """RCCL prints a version banner on the process's stdout when its first communicator comes up; stdout is for the ONE
    JSON line.  File descriptor 1 points at stderr while communicators are created (C-level writes included)."""
    import ctypes

    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def wait_ranks(procs, grace_s=30.0, poll_s=0.05):
    """Waits for ALL rank processes at once.  When one exits non-zero (a rank whose communicator failed leaves with 3
    while its peers may already sit in a collective that can never complete) the others get ``grace_s`` seconds to
    finish on their own, are then terminated -- and killed if they ignore that -- so the launcher never blocks on a
    rank that waits for a dead peer.  Returns the exit codes (negative = ended by that signal)."""
    deadline = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            return codes
        if deadline is None and any(c not in (None, 0) for c in codes):
            deadline = time.monotonic() + grace_s
            print(f"bench.py: a rank exited non-zero ({codes}); the others have {grace_s:.0f} s to finish", file=sys.stderr)
        if deadline is not None and time.monotonic() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.monotonic() + 10.0
            while any(p.poll() is None for p in procs) and time.monotonic() < t_kill:
                time.sleep(poll_s)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            return [p.wait() for p in procs]
        time.sleep(poll_s)


def spawn_ranks(n, argv=None, grace_s=None):
    """``python bench.py --gpus N`` without a launcher: start N fresh rank processes (one per GPU, the env
    variables torch.distributed.run would set), relay their output -- rank 0 prints the JSON line -- and exit
    non-zero if any of them does (``wait_ranks``: no rank is waited for forever).  Children are new interpreters started
    BEFORE this process has made any GPU call; nothing is re-exec'ed.  ``argv``: the children's command line (default:
    this script with this process's arguments)."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    if argv is None:
        argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    if grace_s is None:
        grace_s = float(os.environ.get("MNK_BENCH_GRACE_S", "30"))
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(argv, env=env))
    codes = wait_ranks(procs, grace_s)
    if any(codes):
        sys.exit(f"bench.py: rank exit codes {codes}")


def main():
    args = parse()
    if args.mode == "selfplay":
        return selfplay_mode(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            return spawn_ranks(args.gpus)  # nothing in this process has touched the GPU yet
        args.gpus = world

    import mnk_hip
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay.exchange import RecordExchange
    from selfplay.random_rollout import GatheredLogs, RandomRollout, gather_action_logs, gather_start_state

    mnk_hip.load()
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist = None
    multi = world > 1 or args.rehearse_exchange  # the code path with a process group and an exchange step
    if multi:
        import torch.distributed as dist

        if world == 1:  # rehearsal: no launcher set these
            import socket

            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

        with c_stdout_to_stderr():
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(args.backend, rank=rank, world_size=world)
            dist.barrier()  # brings the backend's communicator up now (a lazily initialising backend would do it later)

    m, n, k = (int(v) for v in args.board.split("x"))
    nenv, chunk = args.envs, args.chunk
    env = TorchVectorMnkEnv(m, n, k, nenv, device=str(dev))
    env.reset()
    roll = RandomRollout(env, seed=args.seed, env_id0=rank * nenv)
    mode = args.gather if multi else "none"
    logging = mode == "actions"
    if logging:  # a group of the log holds plies 4q..4q+3: chunks (and so the warm-up) end on multiples of 4
        chunk = args.chunk = max(4, chunk - chunk % 4)
    keyframe = max(0, args.keyframe)
    every = args.exchange_every if multi else 1  # chunks per exchange step (one all-gather carries `every` messages)
    # two slots (a group of `every` chunks is gathered while the next one computes); the chunks of a slot share one set
    # of record rows (nothing reads them here; a consumer would take them before the slot comes round again)
    bufs = [roll.alloc(chunk) for _ in range(2)]
    from selfplay.random_rollout import LogGroup, RolloutRecords, _msg_words, action_log_format

    fmt = action_log_format(env.max_moves) if logging else 0
    side = exchange = start_state = None
    groups = [{}, {}]   # --gather actions: per slot, state pattern of a group -> LogGroup (send + gathered buffers)
    rec_groups = None   # --gather records: per slot (planes [J, T, R, N], meta [J, T, N], gathered planes, gathered meta)
    chunk_no = [0]  # chunks played so far (keyframes are counted from the first chunk of the run)

    def is_keyframe(c):
        return keyframe == 1 or (keyframe > 1 and c % keyframe == 0)

    def pattern_of(c0):
        """which of the chunks c0 .. c0 + every - 1 carry their chunk-start state"""
        return tuple(is_keyframe(c) for c in range(c0, c0 + every))
    if mode != "none":
        side = torch.cuda.Stream(dev)
        if args.backend == "nccl":  # the collective goes through the C ABI (mnk_allgather_records), RCCL over xGMI
            failure = None
            try:
                with c_stdout_to_stderr():
                    exchange = RecordExchange.from_process_group()
            except Exception as e:  # noqa: BLE001
                failure = e
                exchange = None
            # every rank must take the same path
            ok = torch.tensor([1 if exchange is not None else 0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if exchange is not None:
                    exchange.close()
                    exchange = None
                if not args.allow_fallback:  # a run that quietly measures another transport is worse than no run
                    print(f"[bench rank {rank}] the C-ABI RCCL communicator (mnk_comm_init) could not be created on every "
                          f"rank ({failure}); --allow-fallback runs the all-gather through torch.distributed instead",
                          file=sys.stderr)
                    dist.destroy_process_group()
                    sys.exit(EXIT_COMM)
                print(f"[bench rank {rank}] C-ABI communicator unavailable ({failure}); using torch.distributed",
                      file=sys.stderr)
        if mode == "records":
            rec_groups = []
            for b in bufs:
                planes = torch.empty((every,) + tuple(b.planes.shape), dtype=torch.int64, device=dev)
                meta = torch.empty((every,) + tuple(b.meta.shape), dtype=torch.int32, device=dev)
                rec_groups.append((planes, meta, torch.empty((world,) + tuple(planes.shape), dtype=torch.int64, device=dev),
                                   torch.empty((world,) + tuple(meta.shape), dtype=torch.int32, device=dev)))
        else:
            # every state pattern the keyframe cadence produces (its period in groups divides `keyframe`)
            for g in range(max(keyframe, 1)):
                pat = pattern_of(g * every)
                for slot in (0, 1):
                    if pat not in groups[slot]:
                        groups[slot][pat] = LogGroup(world, env.words, nenv, chunk, fmt, pat, dev, planes=bufs[slot].planes,
                                                     meta=bufs[slot].meta)
    if exchange is not None:
        exchange.direct = args.allgather == "direct"
    main_stream = torch.cuda.current_stream(dev)
    gather_done = [None, None]
    kernel_events = []  # (start, end) HIP events around every rollout launch of the timed region
    gather_events = []  # (start, end) HIP events around every all-gather of the timed region, on the side stream

    timing = [False]

    def gather_group(slot, group):
        """the exchange step of one group of `every` chunks, enqueued on the side stream"""
        if logging:
            group.gather(exchange=exchange, stream=side)
        else:
            planes, meta, gp, gm = rec_groups[slot]
            if exchange is not None:
                exchange.all_gather(planes.view(-1), gp.view(-1), side)
                exchange.all_gather(meta.view(-1), gm.view(-1), side)
            else:
                dist.all_gather_into_tensor(gp.view(-1), planes.view(-1))
                dist.all_gather_into_tensor(gm.view(-1), meta.view(-1))

    def run_steps(total):
        """`total` plies per env: total/chunk launches; every `every` chunks their messages are all-gathered (ONE
        collective) on the side stream while the next group computes.  Returns the number of launches."""
        launches, done, c = 0, 0, 0
        assert total % (chunk * every) == 0  # a bench step is a whole chunk, an exchange a whole group
        group = None
        while done < total:
            slot, j = (c // every) & 1, c % every
            if j == 0:
                if gather_done[slot] is not None:
                    main_stream.wait_event(gather_done[slot])  # the slot's buffers are free again
                if logging:
                    group = groups[slot][pattern_of(chunk_no[0])]
            chunk_no[0] += 1
            if logging:
                out = group.records(j)
            elif mode == "records":
                out = RolloutRecords(planes=rec_groups[slot][0][j], meta=rec_groups[slot][1][j])
            else:
                out = bufs[slot]
            # kernel time: with an exchange step in the loop every launch is bracketed by its own pair of HIP
            # events (the stream also waits for gathers); on one GPU two events around the whole timed region
            # do (per-launch pairs cost ~5 % throughput)
            if timing[0] and mode != "none":
                ks, ke = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ks.record(main_stream)
            roll.run(chunk, out=out)
            if timing[0] and mode != "none":
                ke.record(main_stream)
                kernel_events.append((ks, ke))
            launches += 1
            if mode != "none" and j == every - 1:
                ready = torch.cuda.Event()
                ready.record(main_stream)
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    if timing[0]:
                        gs = torch.cuda.Event(enable_timing=True)
                        gs.record(side)
                    gather_group(slot, group)
                    ev = torch.cuda.Event(enable_timing=timing[0])
                    ev.record(side)
                    gather_done[slot] = ev
                    if timing[0]:
                        gather_events.append((gs, ev))
            done += chunk
            c += 1
        if mode != "none":
            main_stream.wait_stream(side)
        return launches

    def exchange_alone_ms(reps=8, limit_s=60.0):
        """`reps` exchange steps of the every-group message back to back on the side stream with nothing else running,
        by HIP events (this rank's mean); None if they do not finish within `limit_s` -- the caller then agrees with the
        other ranks (over the TCP store, not the GPU) that the form hung, prints its line and leaves with a non-zero
        exit code without touching a collective again."""
        group = groups[0][min(groups[0], key=sum)] if logging else None  # the group with the fewest keyframes in it
        events = []
        barrier()
        with torch.cuda.stream(side):
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(side)
                gather_group(0, group)
                b.record(side)
                events.append((a, b))
        t_start = time.perf_counter()
        while not events[-1][1].query():
            if time.perf_counter() - t_start > limit_s:
                return None
            time.sleep(0.001)
        return sum(a.elapsed_time(b) for a, b in events) / reps

    def barrier():
        if multi:
            # drain this rank's streams first: the side stream's all-gathers run on the C ABI's communicator, the
            # barrier on torch.distributed's -- two RCCL communicators are never in flight at the same time
            torch.cuda.synchronize(dev)
            dist.barrier()
        torch.cuda.synchronize(dev)

    if logging and keyframe == 0:
        # the log travels alone from the first chunk on: every rank takes every shard's start state once (36 B per
        # env), outside the timed region; keeping it current is the receivers' job (replay_shard(..., state=...))
        start_state = gather_start_state(env, exchange=exchange, stream=main_stream)
        torch.cuda.synchronize(dev)
    # multi-rank runs: no phase may block for ever (a collective that never completes cannot be timed out from Python)
    line = [None]  # the JSON line as far as it has been assembled: what a watchdog exit still prints on rank 0

    def print_partial(phase):
        if rank == 0 and line[0] is not None:
            line[0]["aborted_in_phase"] = phase
            print(json.dumps(line[0]), flush=True)
    watchdog = PhaseWatchdog(args.phase_timeout if multi else 0.0, rank, print_partial)
    watchdog.enter("set-up launches")
    run_steps(args.settle * chunk)
    barrier()
    watchdog.enter("warm-up")
    run_steps(args.warmup * chunk)
    barrier()
    watchdog.enter("timed region")
    timing[0] = True
    region0, region1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    region0.record(main_stream)
    launches = run_steps(args.steps * chunk)
    region1.record(main_stream)
    assert launches == args.steps
    barrier()
    dt = time.perf_counter() - t0
    # time spent in the rollout kernel (HIP events on the stream it is launched on)
    dev_ms = sum(a.elapsed_time(b) for a, b in kernel_events) if kernel_events else region0.elapsed_time(region1)
    if multi:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    plies = args.steps * chunk
    value = world * nenv * plies / dt
    gather_ms = [a.elapsed_time(b) for a, b in gather_events]
    # SURVEY.md section 8d asks for repetitions: four more timed regions of the same K steps (not part of `value`)
    reps = [value]
    timing[0] = False
    watchdog.enter("repetitions")
    for _ in range(4):
        barrier()
        r0 = time.perf_counter()
        run_steps(args.steps * chunk)
        barrier()
        rdt = time.perf_counter() - r0
        if multi:
            rmax = torch.tensor([rdt], dtype=torch.float64, device=dev)
            dist.all_reduce(rmax, op=dist.ReduceOp.MAX)
            rdt = float(rmax.item())
        reps.append(world * nenv * plies / rdt)
    # for the decomposition: the same K launches with the exchange step switched off (compute only)
    compute_only = None
    if mode != "none":
        def run_compute_only(total):
            done = 0
            while done < total:
                roll.run(chunk, out=(groups[(done // chunk) & 1][min(groups[0], key=sum)].records(0) if logging
                                     else bufs[(done // chunk) & 1]))
                done += chunk

        watchdog.enter("compute-only run")
        barrier()
        c0 = time.perf_counter()
        run_compute_only(args.steps * chunk)
        barrier()
        cdt = time.perf_counter() - c0
        cmax = torch.tensor([cdt], dtype=torch.float64, device=dev)
        dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
        compute_only = world * nenv * plies / float(cmax.item())
    words = env.words
    # roofline of the dominant kernel (mnk_rollout_random): algorithmic bytes per launch / avg launch time
    plies_per_launch = chunk
    rows = mnk_hip.record_words(m, n)
    alg_bytes = nenv * (plies_per_launch * record_bytes(rows) + 2 * state_bytes(words))
    launch_s = dev_ms * 1e-3 / launches   # HIP events on the kernel's stream over the timed region
    achieved = alg_bytes / launch_s / 1e9
    achieved_wall = alg_bytes / (dt / launches) / 1e9  # the interval `value` is computed from (slowest rank's wall clock)
    survey_b_roll = state_bytes(words) + 8 * words + 4 + state_bytes(words)  # 92 B at 9x9
    out = {
        "metric": f"env-steps/sec {m}x{n}x{k}, {nenv} parallel envs/GPU, random-policy rollout",
        "value": value,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "setup_launches": args.settle,
        "ms_per_step": dt * 1e3 / args.steps,
        "value_without_exchange": compute_only,
        "repetitions": {"values": reps, "median": sorted(reps)[len(reps) // 2], "min": min(reps), "max": max(reps)},
        "step_definition": f"one launch of the fused rollout kernel = {chunk} plies on each of {nenv} envs per GPU",
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": f"{m}x{n}x{k}, {nenv} envs/GPU, fused random rollout (sample+step+win-scan+reset+record), "
                        f"{chunk} plies/launch" + (f", per-chunk RCCL all-gather of {mode} across {world} GPUs"
                                                   if mode != "none" else ""),
            "envs_per_gpu": nenv, "chunk": chunk, "board": args.board, "seed": args.seed, "gather": mode,
        },
        # what exactly ran (the driver compares its N = 1 point with the one-GPU line): the exchange mode, the kernel
        # variant that mode implies, and the transport of the collective
        "gather": mode,
        "kernel_variant": "records + action log " + {1: "(one byte per action)", 2: "(two bytes per action)", 3: "(7-bit stream)",
                                                     4: "(a byte and a bit per action)"}[fmt] if logging
                          else "records only (no action log)",
        "transport": None,
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            # the same fraction on both clocks of this line: `frac` = frac_kernel_events (HIP events around the timed
            # region on the kernel's stream, what the contract asks for); frac_wall = from `ms_per_step`, the barrier-to-
            # barrier wall clock `value` is computed from (it contains the last launch's drain and, with an exchange
            # step, the waits for it): frac_wall <= frac_kernel_events by construction
            "frac_kernel_events": achieved / HBM_PEAK_GBS,
            "frac_wall": achieved_wall / HBM_PEAK_GBS,
            "achieved_wall": achieved_wall,
            "traffic": PMC_TRAFFIC.get((args.board, nenv, chunk), (None, None))[0],
            "traffic_unit": "bytes per launch",
            "traffic_source": PMC_TRAFFIC.get((args.board, nenv, chunk), (None, None))[1],
            "alg_bytes_per_launch": alg_bytes,
            "kernel": "k_rollout_random (the launcher's form for this board and batch: one lane per env, two lanes per env, "
                      "or the run-time specialised variant)",
            "launches": launches,
            "avg_launch_us": launch_s * 1e6,
            "alg_bytes_per_env_step": record_bytes(rows) + 2 * state_bytes(words) / plies_per_launch,
            "achieved_at_survey_B_roll": survey_b_roll * nenv * plies_per_launch / launch_s / 1e9,
        },
    }
    if mode != "none":
        if logging:
            key_bytes = _msg_words(env.words, nenv, chunk, fmt, True) * 8
            log_bytes = _msg_words(env.words, nenv, chunk, fmt, False) * 8
            share = 1.0 if keyframe == 1 else (0.0 if keyframe == 0 else 1.0 / keyframe)
            msg_bytes = share * key_bytes + (1.0 - share) * log_bytes  # average per chunk
        else:
            msg_bytes = bufs[0].planes.numel() * 8 + bufs[0].meta.numel() * 4
        mean_ms = sum(gather_ms) / max(len(gather_ms), 1)  # per exchange step = per `every` chunks
        worst = torch.tensor([mean_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        mean_ms = float(worst.item())
        step_ms = dt * 1e3 / args.steps
        compute_ms = world * nenv * plies / compute_only * 1e3 / args.steps
        exposed_ms = max(0.0, step_ms - compute_ms)
        out["exchange"] = {
            "what": {"actions": "the action log (7-bit stream up to 128 cells, a byte up to 256, a byte and a bit beyond)" +
                                (f"; every {keyframe}th chunk's message also carries the chunk-start state (keyframe), "
                                 "records of any chunk are rebuilt on demand from the last keyframe + the logs since"
                                 if keyframe > 1 else "; every message carries the chunk-start state (self-contained)"
                                 if keyframe == 1 else "; the state was gathered once before the first chunk: receivers "
                                 "must replay every chunk to keep it current"),
                     "records": "packed records (rows + meta word per ply)"}[mode],
            "keyframe_every_chunks": keyframe if logging else None,
            "start_state_bytes_per_rank": start_state.msg.shape[1] * 8 if start_state is not None else 0,
            "transport": ("mnk_allgather_records_direct" if exchange.direct else "mnk_allgather_records") + " (C ABI, RCCL)"
                         if exchange is not None
                         else f"torch.distributed {args.backend}" + (" (FALLBACK: the C-ABI communicator failed)"
                                                                     if args.backend == "nccl" else ""),
            "bytes_per_rank_per_chunk": msg_bytes,
            "chunks_per_exchange": every,                 # --exchange-every: one all-gather carries this many chunks' messages
            "bytes_per_rank_per_exchange": msg_bytes * every,
            "bytes_per_env_step": msg_bytes / (nenv * chunk),
            "allgather_ms": mean_ms,                      # per exchange step; slowest rank's mean, HIP events on the side stream
            "recv_GBps_per_rank": (world - 1) * msg_bytes * every / (mean_ms * 1e-3) / 1e9 if mean_ms else None,
            # every peer's message reaches this rank over that peer's own xGMI link (fully connected mesh): the
            # per-link rate is one message per all-gather time
            "per_link_GBps": msg_bytes * every / (mean_ms * 1e-3) / 1e9 if mean_ms else None,
            "xgmi_link_peak_GBps": 153.0,
            "compute_ms_per_chunk": compute_ms,
            "exposed_ms_per_chunk": exposed_ms,           # step time minus the compute-only step time
            "overlap_fraction": max(0.0, 1.0 - exposed_ms * every / mean_ms) if mean_ms else None,
        }
    if args.rehearse_exchange and world == 1:
        out["rehearsal"] = "the multi-GPU code path on one GPU (a one-rank communicator): not a measurement"
    if mode != "none":
        out["transport"] = out["exchange"]["transport"]
        out["exchange"]["algorithm"] = ("one grouped ncclSend + ncclRecv per peer (mnk_allgather_records_direct)"
                                        if exchange is not None and exchange.direct else
                                        "ncclAllGather" if exchange is not None else "all_gather_into_tensor")
    # both forms of the exchange step alone (rank 0's clock): the one of the timed loop first, the other one last and
    # under a watchdog, because the driver's multi-GPU run is the first time it meets more than one GPU
    alone = None
    exchange_hung = None  # name of the exchange form that never finished, on ANY rank
    line[0] = out
    if exchange is not None:
        alone = {}
        store = dist.distributed_c10d._get_default_store()
        for form in ((False, True) if not exchange.direct else (True, False)):
            name = "direct_sendrecv" if form else "ncclAllGather"
            # the host side of the enqueue can block too (RCCL's group end): the watchdog covers that
            watchdog.enter(f"exchange form alone: {name}")
            exchange.direct = form
            ms = exchange_alone_ms(limit_s=args.alone_limit)
            if os.environ.get("MNK_BENCH_FAKE_HANG") == name:  # tests: take this form to have hung on this rank
                ms = None
            alone[name + "_ms"] = ms
            # one verdict for all ranks, over the TCP store: a rank that finished must not wait in a GPU collective for
            # one that did not
            if agree_any(store, f"alone/{name}", rank, world, ms is None, args.alone_limit + 30.0):
                exchange_hung = name
                print(f"[bench rank {rank}] exchange form {name} alone did not finish within {args.alone_limit:.0f} s on "
                      f"{'this rank' if ms is None else 'another rank'}; leaving with exit code {EXIT_HUNG}", file=sys.stderr)
                break
        exchange.direct = args.allgather == "direct"
    if exchange_hung:
        out["exchange_hung"] = exchange_hung
    watchdog.enter("re-run with the direct form" if exchange is not None else "report")
    if alone is not None and not exchange_hung and args.allgather == "auto":
        # every rank decides on the same numbers: the slowest rank's
        both = torch.tensor([alone["ncclAllGather_ms"], alone["direct_sendrecv_ms"]], dtype=torch.float64, device=dev)
        dist.all_reduce(both, op=dist.ReduceOp.MAX)
        alone["slowest_rank"] = {"ncclAllGather_ms": float(both[0].item()), "direct_sendrecv_ms": float(both[1].item())}
        if float(both[1].item()) < 0.9 * float(both[0].item()) or os.environ.get("MNK_BENCH_FORCE_DIRECT") == "1":
            exchange.direct = True
            del kernel_events[:], gather_events[:]
            run_steps(args.warmup * chunk)
            barrier()
            timing[0] = True
            d0 = time.perf_counter()
            assert run_steps(args.steps * chunk) == args.steps
            barrier()
            ddt = time.perf_counter() - d0
            timing[0] = False
            dmean = sum(a.elapsed_time(b) for a, b in gather_events) / max(len(gather_events), 1)
            second = torch.tensor([ddt, dmean * 1e-3], dtype=torch.float64, device=dev)
            dist.all_reduce(second, op=dist.ReduceOp.MAX)
            ddt, dmean = float(second[0].item()), float(second[1].item()) * 1e3
            dvalue = world * nenv * plies / ddt
            ex = out["exchange"]
            tried = {"value": dvalue, "ms_per_step": ddt * 1e3 / args.steps, "allgather_ms": dmean}
            if dvalue > out["value"]:
                ex["with_ncclAllGather"] = {"value": out["value"], "ms_per_step": out["ms_per_step"],
                                            **{key: ex[key] for key in ("allgather_ms", "recv_GBps_per_rank", "per_link_GBps",
                                                                        "exposed_ms_per_chunk", "overlap_fraction")}}
                out["value"], out["ms_per_step"] = dvalue, tried["ms_per_step"]
                exposed = max(0.0, tried["ms_per_step"] - ex["compute_ms_per_chunk"])
                ex.update(allgather_ms=dmean, exposed_ms_per_chunk=exposed,
                          recv_GBps_per_rank=(world - 1) * ex["bytes_per_rank_per_chunk"] / (dmean * 1e-3) / 1e9 if dmean else None,
                          per_link_GBps=ex["bytes_per_rank_per_chunk"] / (dmean * 1e-3) / 1e9 if dmean else None,
                          overlap_fraction=max(0.0, 1.0 - exposed / dmean) if dmean else None,
                          algorithm="one grouped ncclSend + ncclRecv per peer (mnk_allgather_records_direct)",
                          transport="mnk_allgather_records_direct (C ABI, RCCL)")
                out["transport"] = ex["transport"]
                out["repetitions"]["note"] = "the repetitions ran with ncclAllGather; `value` is the re-run with the direct form"
                ex["auto"] = "switched to the direct form: faster alone and in the loop"
            else:
                ex["with_direct_sendrecv"] = tried
                ex["auto"] = "kept ncclAllGather: the direct form was faster alone but not in the loop"
                exchange.direct = False
        else:
            out["exchange"]["auto"] = "kept ncclAllGather: the direct form alone was not 10 % faster on the slowest rank"
    if alone is not None:
        alone["what"] = (f"the exchange step ({every} chunk message{'s' if every > 1 else ''} per all-gather, the group with the "
                         "fewest keyframes) alone: 8 back to back, nothing else running, this rank's HIP events; null = did "
                         f"not finish within {args.alone_limit:.0f} s (the run then leaves with exit code {EXIT_HUNG})")
        out["exchange"]["alone"] = alone
    if rank == 0:
        if not exchange_hung:  # (a device read: not with a collective stuck on the GPU)
            stats = roll.stats.tolist()
            out["rollout_stats"] = {"episodes": stats[0], "mean_plies": stats[4] / max(stats[0], 1),
                                    "draw_rate": stats[3] / max(stats[0], 1)}
        out["gpu"] = gpu_identity(dev)
        jit = mnk_hip.jit_stats()
        if jit["compiled"] or jit["cache_hits"] or jit["failed"]:  # a board without a built-in variant: what hiprtc did for it
            out["jit"] = jit
        if not multi and not args.no_api_path:
            out["roofline"]["measured_write_ceiling_GBps"] = write_ceiling_GBps(dev, nenv, chunk, rows)
            out["roofline"]["frac_of_measured_write_ceiling"] = achieved / out["roofline"]["measured_write_ceiling_GBps"]
            out["api_path_env_steps_per_s"] = api_path_rate(env, args.seed)
            out["api_path_graphed_env_steps_per_s"] = api_path_graphed_rate(env, args.seed)
            out["api_path_graphed_fused_reset_env_steps_per_s"] = api_path_graphed_rate(env, args.seed, fused_reset=True)
            one, per_ply_us, floor_us = api_path_one_launch_rate(env, args.seed)
            out["api_path_graphed_one_launch_env_steps_per_s"] = one
            out["api_path_one_launch"] = {
                "us_per_ply": per_ply_us, "launch_floor_us": floor_us,
                "what": "mnk_step_random: draw + step + win scan + reset + legal mask in one launch per ply, 64 plies per "
                        "hipGraph; launch_floor_us = the same graph on 64 envs (what one dependent launch costs)",
                # SURVEY.md section 8d's B_step without the action read: state in, mover's plane + meta out, reward, done, mask
                "alg_bytes_per_env_step": state_bytes(words) + 8 * words + 4 + 5 + m * n,
                "frac_of_hbm_peak": (state_bytes(words) + 8 * words + 4 + 5 + m * n) * one / 1e9 / HBM_PEAK_GBS,
            }
            out["replay_env_steps_per_s"] = replay_rate(env, roll, chunk)
            out["with_action_log"] = with_action_log_rate(roll, chunk, args.steps)
            out["selfplay"] = selfplay_object(m, n, k, nenv, args.seed, dev)
            # the reference trains at 384 envs with a NEW opponent before every rollout (train.py:106-114, :246)
            te, ts, tr = train_cadence(384, rollouts=6, m=m, n=n, k=k, dev=str(dev))
            out["selfplay"]["train_cadence_384_envs"] = {
                "what": "64 agent-steps per rollout, network agent vs network opponent (3 x conv3x3(32)), a new opponent "
                        "(deepcopy of the agent) before every rollout: us per agent-step",
                "eager_reference_loop_us": te, "one_graph_inplace_opponent_swap_us": ts, "recapture_per_rollout_us": tr,
                "speedup_swap_vs_eager": te / ts, "agent_steps_per_s_swap": 384 / ts * 1e6,
                "env_side_launches_per_agent_step": 2}
            # BASELINE config 3 proper (conv policy as agent, pool of 4 network opponents), a short run of what
            # `--mode selfplay` measures at length: the caller-side networks are >99 % of it
            torch.cuda.empty_cache()
            c3 = selfplay_config3(args, 48, 16)
            out["selfplay"]["config3"] = {key: c3[key] for key in ("metric", "value", "unit", "ms_per_step", "env_steps_per_s",
                                                                   "env_side_ms_per_step", "nn_and_sampling_ms_per_step",
                                                                   "buffer_add_copied_bytes_per_agent_step")}
            out["selfplay"]["config3"]["workload"] = c3["config"]["workload"]
        if not multi and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(env, args.cpu_seconds, args.cpu_threads)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if exchange_hung:  # an exchange form never finished: no synchronisation or collective can be trusted any more --
        sys.stdout.flush()  # the line is out; leave NON-ZERO (the driver's rc must say that this run hung) and at once
        sys.stderr.flush()
        os._exit(EXIT_HUNG)
    watchdog.enter("teardown")
    if exchange is not None:
        torch.cuda.synchronize(dev)
        exchange.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    watchdog.stop()


if __name__ == "__main__":
    main()

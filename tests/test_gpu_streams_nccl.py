"""GPU: the kernels follow torch's current stream, and the exchange step works on the RCCL backend."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hip():
    import __graft_entry__ as entry

    entry.build_hip()
    entry._ensure_path()
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay.random_rollout import RandomRollout

    return TorchVectorMnkEnv, RandomRollout


def test_kernels_run_on_the_current_stream():
    """Work enqueued under torch.cuda.stream(side) is ordered with that stream's other work: a rollout on a
    side stream, consumed after wait_stream, equals the same rollout on the default stream."""
    Env, Rollout = _hip()
    ref = Rollout(Env(9, 9, 5, 4096, device=DEV), seed=3).run(64)
    side = torch.cuda.Stream(DEV)
    env = Env(9, 9, 5, 4096, device=DEV)
    roll = Rollout(env, seed=3)
    torch.cuda.current_stream(DEV).synchronize()
    with torch.cuda.stream(side):
        big = torch.randn(4096, 4096, device=DEV)
        for _ in range(10):  # keep the side stream busy ahead of the rollout
            big = big @ big.t() * 1e-4
        rec = roll.run(64)
        obs = env.observe()
    torch.cuda.current_stream(DEV).wait_stream(side)
    assert torch.equal(rec.planes, ref.planes) and torch.equal(rec.meta, ref.meta)
    assert obs["observation"].shape == (4096, 2, 9, 9)


_NCCL_CHILD = r"""
import os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "rl-selfplay-mnk_amd")]
import torch, torch.distributed as dist
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.random_rollout import RandomRollout, gather_action_logs, gather_records, replay_shard
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
env = TorchVectorMnkEnv(9, 9, 5, 2048, device="cuda:0")
roll = RandomRollout(env, seed=1)
rec = roll.alloc(64, log_actions=True)
roll.run(64, out=rec)
side = torch.cuda.Stream(dev)
done = torch.cuda.Event(); done.record()
with torch.cuda.stream(side):
    side.wait_event(done)
    logs = gather_action_logs(rec)          # RCCL all-gather of the int64 message
    full = gather_records(rec)              # world size 1: identity
torch.cuda.current_stream().wait_stream(side)
again = replay_shard(logs, 0, 9, 9, 5)
assert torch.equal(again.planes, rec.planes) and torch.equal(again.meta, rec.meta)
assert full is rec
t = torch.ones(4, device=dev); dist.all_reduce(t); assert float(t.sum()) == 4.0
# the same exchange through the C ABI (mnk_comm_* / mnk_allgather_records): id from rank 0 over the process group,
# communicator on this rank's device, all-gather enqueued on the side stream
from selfplay.exchange import RecordExchange
import mnk_hip
assert mnk_hip.load().mnk_comm_version() >= 20000
ex = RecordExchange.from_process_group()
assert (ex.rank, ex.world) == (0, 1)
roll.run(64, out=rec)
done.record()
with torch.cuda.stream(side):
    side.wait_event(done)
    logs2 = gather_action_logs(rec, exchange=ex, stream=side)
torch.cuda.current_stream().wait_stream(side)
assert torch.equal(logs2.msg[0], rec.msg)
again = replay_shard(logs2, 0, 9, 9, 5)
assert torch.equal(again.planes, rec.planes) and torch.equal(again.meta, rec.meta)
# the per-peer send / receive form of the exchange (mnk_allgather_records_direct): ncclGroupStart / ncclSend / ncclRecv /
# ncclGroupEnd -- on one rank the message travels to "peer 0", i.e. through RCCL's send-to-self
ex.direct = True
roll.run(64, out=rec)
done.record()
with torch.cuda.stream(side):
    side.wait_event(done)
    logs3 = gather_action_logs(rec, exchange=ex, stream=side)
torch.cuda.current_stream().wait_stream(side)
assert torch.equal(logs3.msg[0], rec.msg)
# the exchange step of the self-play wrapper's rollout buffer through the same communicator, on the side stream, in both
# forms (PackedRolloutBuffer.all_gather: planes + actions + log-probabilities + values + advantages; a bool field too)
from alg.packed_rollout_buffer import ALL_FIELDS, PackedRolloutBuffer
buf = PackedRolloutBuffer(6, 300, 9, 9, device="cuda:0")
gen = torch.Generator(device=dev).manual_seed(2)
buf.planes.random_(generator=gen); buf.actions.random_(0, 81, generator=gen)
for f in (buf.log_probs, buf.values, buf.advantages, buf.rewards):
    f.normal_(generator=gen)
buf.returns.copy_(buf.advantages + buf.values); buf.dones.copy_(torch.rand(6, 300, device=dev, generator=gen) < 0.2)
buf.ptr = 6
for direct in (False, True):
    ex.direct = direct
    done.record()
    with torch.cuda.stream(side):
        side.wait_event(done)
        got_buf = buf.all_gather(exchange=ex, stream=side, fields=ALL_FIELDS if direct else ("planes", "actions", "log_probs", "values", "advantages"))
    torch.cuda.current_stream().wait_stream(side)
    for f in ("planes", "actions", "log_probs", "values", "advantages", "returns") + (("rewards", "dones") if direct else ()):
        assert torch.equal(getattr(got_buf, f), getattr(buf, f)), (direct, f)
odd = torch.arange(1, 1 + 13, dtype=torch.uint8, device=dev)   # a message that is not a multiple of 8 bytes
got = torch.zeros(13, dtype=torch.uint8, device=dev)
ex.all_gather(odd, got)
torch.cuda.synchronize()
assert torch.equal(got, odd)
ex.direct = False
try:
    ex.all_gather(rec.msg, torch.empty(3, dtype=torch.int64, device=dev))
    raise SystemExit("a short receive buffer was accepted")
except ValueError:
    pass
torch.cuda.synchronize()
ex.close()
dist.barrier(); dist.destroy_process_group()
print("NCCL_OK")
"""


def test_exchange_step_on_the_rccl_backend():
    """One rank, backend nccl (= RCCL): process-group init on the GPU, the all-gather of the action-log message
    on a side stream -- once through torch.distributed, once through the C ABI's own communicator
    (mnk_comm_init / mnk_allgather_records) -- and replay of the gathered shard.  (More ranks need more GPUs:
    the driver's multi-GPU run.)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", _NCCL_CHILD % {"root": ROOT}], env=env, capture_output=True,
                         text=True, timeout=300)
    assert "NCCL_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


_TWO_RANK_CHILD = r"""
import os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "rl-selfplay-mnk_amd")]
import torch, torch.distributed as dist
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.random_rollout import KeyframedLogs, RandomRollout, gather_action_logs, gather_records
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("gloo", rank=rank, world_size=world)
m, n, k, nenv, chunk, every = %(m)d, %(n)d, %(k)d, 320, 24, 3
env = TorchVectorMnkEnv(m, n, k, nenv, device="cuda:0")
roll = RandomRollout(env, seed=9, env_id0=rank * nenv)          # the shard's global env ids key the RNG
history = KeyframedLogs(m, n, k)
truth = []
for c in range(7):
    key = c %% every == 0
    rec = roll.alloc(chunk, log_actions=True, with_state=key)
    roll.run(chunk, out=rec)
    history.push(gather_action_logs(rec))                       # the exchange step: ONE all-gather of the message
    full = gather_records(rec)                                   # ground truth: every shard's records themselves
    truth = [full] if key else truth + [full]
    for shard in range(world):                                   # rebuild EVERY shard's records of EVERY held chunk
        for j, want in enumerate(truth):
            got = history.rebuild(shard, j)
            cols = slice(shard * nenv, (shard + 1) * nenv)
            assert torch.equal(got.planes, want.planes[:, :, cols]) and torch.equal(got.meta, want.meta[:, cols]), (c, shard, j)
# and the sharded run as a whole equals one process with all the envs
if rank == 0:
    whole = RandomRollout(TorchVectorMnkEnv(m, n, k, world * nenv, device="cuda:0"), seed=9)
    for c in range(7):
        last = whole.run(chunk)
    assert torch.equal(last.planes, truth[-1].planes) and torch.equal(last.meta, truth[-1].meta)
dist.barrier()
dist.destroy_process_group()
print("RANK_OK", rank)
"""


@pytest.mark.parametrize("m,n,k", [(9, 9, 5), (19, 19, 5)])
def test_two_ranks_exchange_keyframed_logs_and_rebuild_each_others_records(m, n, k):
    """Two rank processes (gloo rendezvous and transport; both on this box's one GPU), each rolling out its own env
    shard: every chunk's message is all-gathered (a keyframe every third chunk, the log alone otherwise; the 7-bit
    stream at 9x9, the byte + bit log at 19x19), and every rank rebuilds EVERY shard's records of every chunk held
    since the last keyframe -- equal to the records that shard wrote itself (all-gathered as ground truth), and the
    two shards together equal a single process that holds all the envs."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", _TWO_RANK_CHILD % {"root": ROOT, "m": m, "n": n, "k": k}], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for rank, (out, err) in enumerate(outs):
        assert f"RANK_OK {rank}" in out, out[-2000:] + err[-4000:]


_TWO_RANK_BUFFER_CHILD = r"""
import copy, os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "rl-selfplay-mnk_amd")]
import torch, torch.distributed as dist
from alg.packed_rollout_buffer import ALL_FIELDS, UPDATE_FIELDS, PackedRolloutBuffer
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.policy import FusedNNPolicy, HipSampler
from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("gloo", rank=rank, world_size=world)
m, n, k, nenv, steps = 9, 9, 5, 160, 12
c = m * n


class RowNet(torch.nn.Module):
    # a "network" made of element-wise operations and exact small-integer sums only: the logits of a row are the same
    # bits whatever the batch size (a conv / GEMM may pick another algorithm for 160 rows than for 320)
    def __init__(self, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w = torch.nn.Parameter(torch.randn(2, c, generator=g))
        self.b = torch.nn.Parameter(torch.randn(c, generator=g))

    def forward(self, obs, action_mask=None):
        x = obs.float().flatten(2)                                  # [B, 2, C] of 0 / 1
        lead = (x[:, 0].sum(1, keepdim=True) - x[:, 1].sum(1, keepdim=True))  # exact
        logits = x[:, 0] * self.w[0] + x[:, 1] * self.w[1] + self.b + 0.25 * lead

        class Dist:
            pass

        d = Dist()
        d.logits = logits
        return d, torch.tanh(0.125 * lead)


agent, opp_net = RowNet(1).to(dev).eval(), RowNet(2).to(dev).eval()


def rollout(count, id0):
    # the shard's global env ids key every random stream: the wrapper's sides, the agent's and the opponent's draws
    wrap = TorchSelfPlayWrapper(TorchVectorMnkEnv(m, n, k, count, device="cuda:0"), seed=11)
    wrap.env_id0 = id0
    opp = FusedNNPolicy(copy.deepcopy(opp_net), seed=5)
    opp._sampler.env_id0 = id0
    wrap.set_opponent(opp)
    sampler = HipSampler(seed=7)
    sampler.env_id0 = id0
    buf = PackedRolloutBuffer(steps, count, m, n, device="cuda:0")
    wrap.attach_sink(buf)
    obs, _ = wrap.reset()
    for t in range(steps):
        with torch.no_grad():
            d, values = agent(obs["observation"], None)
        cur = buf.row(t)["packed"]                                   # where reset() / the previous step put the planes
        nxt, rew, term, trunc, info = wrap.step_logits(d.logits, obs["action_mask"], sampler)
        buf.add(cur, info["actions"], rew, values, info["log_probs"], term | trunc)
        obs = nxt
    with torch.no_grad():
        _, last = agent(obs["observation"], None)
    buf.compute_advantages_and_returns(last.reshape(-1), 0.99, 0.95)
    wrap.env.check_errors()
    return buf


mine = rollout(nenv, rank * nenv)
full = mine.all_gather()                                             # the exchange step (gloo here, RCCL on the GPUs of a node)
everything = mine.all_gather(fields=ALL_FIELDS)
assert full.n_steps == world * steps and full.ptr == full.n_steps and full.num_envs == nenv
for f in ("planes", "actions", "log_probs", "values", "advantages", "returns"):   # returns: recomputed == gathered
    assert torch.equal(getattr(full, f), getattr(everything, f)), f
    assert torch.equal(getattr(full, f)[rank * steps:(rank + 1) * steps], getattr(mine, f)), f
assert not full.rewards.any() and torch.equal(everything.dones[rank * steps:(rank + 1) * steps], mine.dones)
again = mine.all_gather(out=full)
assert again is full
if rank == 0:   # the shards together are what one process with all the envs plays and stores
    whole = rollout(world * nenv, 0)
    for r in range(world):
        rows, cols = slice(r * steps, (r + 1) * steps), slice(r * nenv, (r + 1) * nenv)
        assert torch.equal(everything.planes[rows], whole.planes[..., cols]), r
        for f in ("actions", "log_probs", "values", "advantages", "returns", "rewards", "dones"):
            assert torch.equal(getattr(everything, f)[rows], getattr(whole, f)[:, cols]), (r, f)
    # a minibatch drawn from the gathered buffer = the same samples of the single-process buffer
    g = torch.Generator().manual_seed(4)
    pick = torch.randperm(world * steps * nenv, generator=g)[:700].to(dev)
    r_, t_, i_ = pick // (steps * nenv), (pick // nenv) %% steps, pick %% nenv
    obs_a, mask_a = full.gather(pick)
    obs_b, mask_b = whole.gather(t_ * (world * nenv) + r_ * nenv + i_)
    assert torch.equal(obs_a, obs_b) and torch.equal(mask_a, mask_b)
    batches = list(full.get_data_loader(4096))
    assert sum(b[0].shape[0] for b in batches) == world * steps * nenv
dist.barrier()
dist.destroy_process_group()
print("RANK_OK", rank)
"""


def test_two_ranks_all_gather_their_self_play_rollout_buffers():
    """SURVEY 8e for the self-play wrapper's path: two rank processes (gloo; both on this box's GPU), each playing a
    rollout of network agent against network opponent on its own block of envs (global env ids key the sides and both
    samplers) straight into a ``PackedRolloutBuffer`` (the sink), GAE per shard, then ``buffer.all_gather()`` -- the
    packed planes + actions + log-probabilities + values + advantages of every rank, 52 B per agent-step at 9x9.  The
    gathered buffer holds every shard's steps, its recomputed returns are the gathered ones bit for bit, and the two
    shards together are what a single process with all the envs plays, stores and draws minibatches from."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", _TWO_RANK_BUFFER_CHILD % {"root": ROOT}], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for rank, (out, err) in enumerate(outs):
        assert f"RANK_OK {rank}" in out, out[-2000:] + err[-4000:]


def test_sharded_self_play_ppo_example_keeps_its_replicas_identical():
    """examples/selfplay_ppo_sharded.py: two ranks (gloo; both on this box's GPU) play their own env shards, all-gather
    their rollout buffers and run the same PPO update on the union -- no gradient communication -- and their weights stay
    bit-identical while the policy learns; then one rank on the RCCL backend, the all-gather through the C ABI's
    communicator."""
    import re

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for var in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(var, None)
    script = os.path.join(ROOT, "examples", "selfplay_ppo_sharded.py")
    out = subprocess.run([sys.executable, script, "--ranks", "2", "--backend", "gloo", "--envs", "512", "--iters", "20"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("iter")]
    assert len(lines) == 2 and all("replicas identical: True" in ln for ln in lines), out.stdout[-2000:]
    assert "2 ranks x 512 envs, 16384 samples per update (32 gathered steps)" in lines[-1]
    assert float(re.search(r"score vs random ([0-9.]+)", lines[-1]).group(1)) > 0.6, lines[-1]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, script, "--backend", "nccl", "--envs", "512", "--iters", "3"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "1 ranks x 512 envs" in out.stdout and "replicas identical: True" in out.stdout, out.stdout[-2000:]


@pytest.mark.parametrize("extra", [[], ["--allgather", "direct", "--keyframe", "1"], ["--gather", "records"], ["FORCE_SWITCH"],
                                   ["--exchange-every", "2"], ["--exchange-every", "2", "--gather", "records"]])
def test_bench_multi_gpu_code_path_rehearsed_on_one_gpu(extra):
    """``bench.py --rehearse-exchange``: the code path the driver's multi-GPU run takes -- process group (RCCL), the C ABI's
    communicator, the exchange step on a side stream overlapping the next chunk, both exchange forms timed alone under a
    watchdog -- with one rank on this box's GPU.  stdout must hold the ONE JSON line and nothing else (RCCL's version
    banner goes to stderr)."""
    import json

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for var in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(var, None)
    force = extra == ["FORCE_SWITCH"]
    if force:  # --allgather auto's second timed region (taken when the direct form is faster alone), forced
        env["MNK_BENCH_FORCE_DIRECT"] = "1"
        extra = []
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-exchange", "--steps", "6", "--warmup", "2",
                          "--settle", "8", "--envs", "8192"] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and "rehearsal" in d and d["gather"] == ("records" if "records" in extra else "actions")
    ex = d["exchange"]
    assert d["transport"] == ex["transport"] and "C ABI, RCCL" in ex["transport"]
    if not force:
        assert ("direct" in ex["transport"]) == ("direct" in extra)
    assert ex["alone"]["ncclAllGather_ms"] > 0 and ex["alone"]["direct_sendrecv_ms"] > 0
    assert d["value"] > 0 and d["value_without_exchange"] > 0 and ex["allgather_ms"] > 0
    every = 2 if "--exchange-every" in extra else 1
    assert ex["chunks_per_exchange"] == every and ex["bytes_per_rank_per_exchange"] == every * ex["bytes_per_rank_per_chunk"]
    r = d["roofline"]
    assert r["frac"] == r["frac_kernel_events"] and 0 < r["frac_wall"] <= 1.02 * r["frac_kernel_events"]
    if "--allgather" not in extra:
        assert ex["alone"]["slowest_rank"]["ncclAllGather_ms"] > 0 and "auto" in ex
        if force:
            assert ("with_ncclAllGather" in ex) != ("with_direct_sendrecv" in ex)
            assert ("direct" in ex["transport"]) == ("with_ncclAllGather" in ex)


@pytest.mark.parametrize("form", ["ncclAllGather", "direct_sendrecv"])
def test_bench_leaves_non_zero_when_an_exchange_form_hangs(form):
    """The branch a hung exchange form takes (``exchange_alone_ms`` -> None, here faked for one form): the ranks agree over
    the TCP store, rank 0 still prints its line -- with ``exchange_hung`` naming the form -- and the process leaves with
    exit code 4, never 0: the driver's rc must say that the run did not end well."""
    import json

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MNK_BENCH_FAKE_HANG=form)
    for var in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(var, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-exchange", "--steps", "4", "--warmup", "2",
                          "--settle", "4", "--envs", "4096"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 4, (out.returncode, out.stderr[-3000:])
    assert f"exchange form {form} alone did not finish" in out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[:2000]
    d = json.loads(lines[0])
    assert d["exchange_hung"] == form and d["exchange"]["alone"][form + "_ms"] is None and d["value"] > 0


def test_bench_two_gloo_ranks_exchange_every_two_chunks():
    """``bench.py --gpus 2 --exchange-every 2`` with self-spawned ranks (gloo; both on this box's GPU): one all-gather per two
    chunks carrying both messages, the launcher relaying rank 0's line and the ranks' exit codes."""
    import json

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for var in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(var, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--exchange-every",
                          "2", "--keyframe", "4", "--steps", "8", "--warmup", "2", "--settle", "4", "--envs", "4096"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    ex = d["exchange"]
    assert d["n_gpus"] == 2 and ex["chunks_per_exchange"] == 2 and ex["keyframe_every_chunks"] == 4
    assert ex["bytes_per_env_step"] == pytest.approx(0.875 + 36 / (256 * 4), abs=2e-3)

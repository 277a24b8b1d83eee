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

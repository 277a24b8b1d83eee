"""Self-play PPO with the env axis sharded over ranks -- one process per GPU -- and the rollout buffers all-gathered
(north_star: "env batches shard naturally across the GPUs of one node with RCCL all-gather of
rollout buffers over xGMI").

Every rank owns a block of envs (``env_id0 = rank * envs`` keys the wrapper's side draws and both samplers, so the ranks
together play exactly the games ONE process with all the envs would play), rolls out into its own
``PackedRolloutBuffer`` through the sink, computes its advantages, and then

    full = buf.all_gather(exchange=ex)        # packed planes + action + log-prob + value + advantage: 52 B per agent-step at 9x9

hands every rank everybody's samples.  Each rank then runs the SAME PPO update on the union -- same initial weights, same
shuffle (the device generator is seeded alike), same arithmetic -- so the replicas stay bit-identical WITHOUT any gradient
communication: the all-gather of the rollout buffers is the only collective.  At the end the ranks compare a digest of
their weights.

    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 examples/selfplay_ppo_sharded.py     # 8 GPUs: RCCL through the C ABI
    python examples/selfplay_ppo_sharded.py --ranks 2 --backend gloo                         # rehearsal: two ranks on ONE GPU
"""
import argparse
import copy
import hashlib
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", default="3x3x3")
    ap.add_argument("--envs", type=int, default=1024, help="envs PER RANK")
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"))
    ap.add_argument("--ranks", type=int, default=0, help="start this many rank processes here (no launcher needed)")
    return ap.parse_args()


def spawn(args):
    """``--ranks R`` without a launcher: R fresh processes (before anything here touches the GPU), rank 0's output relayed"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = [a for a in sys.argv[1:]]
    i = argv.index("--ranks")
    del argv[i:i + 2]
    procs = []
    for rank in range(args.ranks):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                   WORLD_SIZE=str(args.ranks), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    codes = [p.wait() for p in procs]
    sys.exit(max(codes))


def main():
    args = parse()
    if args.ranks and "RANK" not in os.environ:
        return spawn(args)
    import torch
    import torch.distributed as dist
    import torch.nn as nn

    import __graft_entry__ as entry

    entry.build()
    from alg.packed_rollout_buffer import PackedRolloutBuffer
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay.exchange import RecordExchange
    from selfplay.policy import FusedNNPolicy, HipSampler, NNPolicy, RandomPolicy
    from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper
    from selfplay.validation import validate_gpu

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if args.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        exchange = RecordExchange.from_process_group()   # mnk_comm_unique_id on rank 0 -> broadcast -> mnk_comm_init
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        exchange = None                                   # the same all-gather through torch.distributed

    class ActorCritic(nn.Module):
        def __init__(self, cells):
            super().__init__()
            self.body = nn.Sequential(nn.Flatten(), nn.Linear(2 * cells, 256), nn.Tanh(), nn.Linear(256, 256), nn.Tanh())
            self.pi, self.v = nn.Linear(256, cells), nn.Linear(256, 1)

        def forward(self, obs, action_mask=None):
            h = self.body(obs.float())
            logits = self.pi(h)
            if action_mask is not None:
                logits = torch.where(action_mask.bool(), logits, torch.full_like(logits, -torch.inf))
            return torch.distributions.Categorical(logits=logits, validate_args=False), torch.tanh(self.v(h))

    m, n, k = (int(v) for v in args.board.split("x"))
    cells, nenv, id0 = m * n, args.envs, rank * args.envs
    torch.manual_seed(0)                      # the same initial weights and the same device generator on every rank
    torch.cuda.manual_seed(0)
    net = ActorCritic(cells).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=2e-3)
    wrap = TorchSelfPlayWrapper(TorchVectorMnkEnv(m, n, k, nenv, device=str(dev)), seed=1)
    wrap.env_id0 = id0                        # global env ids key every random stream of the shard
    wrap.track_episodes()
    sampler = HipSampler(seed=2)
    sampler.env_id0 = id0
    buf = PackedRolloutBuffer(args.steps, nenv, m, n, device=str(dev))
    wrap.attach_sink(buf)
    full = None
    obs = None
    for it in range(args.iters):
        opponent = FusedNNPolicy(copy.deepcopy(net), seed=100 + it)   # train.py:106-114: a fresh copy of the agent
        opponent._sampler.env_id0 = id0
        wrap.set_opponent(opponent)
        if obs is None:
            obs, _ = wrap.reset()
        packed = buf.row(0 if it == 0 else args.steps)["packed"]   # reset() wrote row 0; later rollouts start from the spill row
        for t in range(args.steps):
            with torch.no_grad():
                dist_, values = net(obs["observation"], None)
            obs, rewards, term, trunc, info = wrap.step_logits(dist_.logits, obs["action_mask"], sampler)
            buf.add(packed, info["actions"], rewards, values, info["log_probs"], term | trunc)
            packed = buf.row(t + 1)["packed"]
        with torch.no_grad():
            _, last = net(obs["observation"], None)
        buf.compute_advantages_and_returns(last.reshape(-1), 0.99, 0.95)
        full = buf.all_gather(exchange=exchange, out=full)        # THE exchange step: everybody's rollout on every rank
        for _ in range(4):
            for b_obs, b_act, b_logp, b_ret, b_adv, b_mask, _ in full.get_data_loader(8192):
                dist_, value = net(b_obs, b_mask)
                ratio = torch.exp(dist_.log_prob(b_act) - b_logp)
                surrogate = torch.min(ratio * b_adv, torch.clamp(ratio, 0.8, 1.2) * b_adv).mean()
                loss = -surrogate + 0.5 * (value.reshape(-1) - b_ret).pow(2).mean() - 0.01 * dist_.entropy().mean()
                opt.zero_grad()
                loss.backward()
                opt.step()
        buf.reset()
        if it % 10 == 9 or it == args.iters - 1:
            stats = wrap.pop_episode_stats()
            digest = hashlib.sha256(b"".join(p.detach().cpu().numpy().tobytes() for p in net.parameters())).hexdigest()
            digests = [None] * world
            dist.all_gather_object(digests, digest)
            if rank == 0:
                rng = torch.cuda.get_rng_state(dev)   # whatever the validation draws must not move rank 0's shuffle
                res = validate_gpu(NNPolicy(net), RandomPolicy(cells), (m, n, k), n_episodes=4096, device=str(dev))
                torch.cuda.set_rng_state(rng, dev)
                net.train()
                print(f"iter {it + 1:3d}: {world} ranks x {nenv} envs, {world * args.steps * nenv} samples per update "
                      f"({full.n_steps} gathered steps); rank 0 played {stats['episodes']} games; score vs random "
                      f"{res['validation/vs_benchmark/score_rate']:.3f}; replicas identical: {len(set(digests)) == 1}", flush=True)
    wrap.env.check_errors()
    if exchange is not None:
        torch.cuda.synchronize(dev)
        exchange.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Generate the golden fixtures in this directory from the imported reference.

Run in the build container only (``/root/reference`` is mounted there):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The fixtures are data only -- inputs (actions, index subsets, sides, poked
positions, logits) and the outputs the reference produced for them -- stored
bit-packed in the layout of ``oracle/packing.py``.  No reference source is stored.

Files written
  env_<m>x<n>x<k>_s<seed>.npz   G1+G2: op-log of TorchVectorMnkEnv.step / step_subset / reset(idx)
  selfplay_<..>_<opp>_s<seed>.npz  G3: TorchSelfPlayWrapper trace with a row-local deterministic opponent
  edges.npz                     G4: the hand-written scenarios of tests/scenarios.py
  masked_logits.npz             G6 (epilogue part): masked-categorical head of the reference nets
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.packing import pack_boards, pack_cells  # noqa: E402
from oracle.pin_against_reference import import_reference  # noqa: E402
from oracle.policies import HighestLegalPolicy, LowestLegalPolicy, MaskHashPolicy  # noqa: E402
from scenarios import SCENARIOS  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

ENV_CASES = [
    # m, n, k, N, T, seeds
    (3, 3, 3, 64, 40, (0, 1, 2)),
    (4, 6, 3, 32, 40, (0, 1)),
    (9, 9, 5, 256, 160, (0, 1, 2)),
    (13, 13, 5, 64, 200, (0,)),
    (19, 19, 5, 32, 300, (0,)),
    (7, 9, 7, 16, 100, (0,)),
]

SELFPLAY_CASES = [
    # m, n, k, N, T, opponent, seed
    (3, 3, 3, 64, 40, "lowest", 0),
    (3, 3, 3, 64, 40, "hash", 1),
    (9, 9, 5, 128, 150, "hash", 0),
    (9, 9, 5, 128, 150, "highest", 1),
    (4, 6, 3, 32, 50, "hash", 0),
    (13, 13, 5, 32, 220, "hash", 0),
    (19, 19, 5, 16, 320, "hash", 0),
]

OPPONENTS = {"lowest": LowestLegalPolicy, "highest": HighestLegalPolicy, "hash": MaskHashPolicy}

# Round 4 (``--more-boards``): boards WITHOUT a built-in kernel variant -- their kernels are compiled at run time
# (csrc/mnk_jit.hip) -- under a prefix of their own, so the indices the older tests address the files above by stay put:
#   boards_env_<m>x<n>x<k>_s<seed>.npz, boards_selfplay_<m>x<n>x<k>_<opp>_s<seed>.npz
# 1 / 2 / 5 register words per plane, a connect-four-shaped board, rows wider than 31 cells (table write-out)
MORE_ENV_CASES = [(5, 5, 4, 48, 60, (0,)), (6, 7, 4, 64, 90, (0,)), (12, 12, 5, 48, 260, (0,)), (10, 10, 5, 40, 200, (0,)),
                  (3, 33, 3, 24, 120, (0,))]
MORE_SELFPLAY_CASES = [(6, 7, 4, 64, 90, "hash", 0), (12, 12, 5, 40, 220, "highest", 0), (5, 5, 4, 48, 60, "lowest", 0),
                       (10, 10, 5, 32, 180, "hash", 1)]


def _state(env, m, n):
    return (
        pack_boards(env.boards.numpy(), m, n),
        env.current_player.numpy().astype(np.uint8),
        env.move_counts.numpy().astype(np.int32),
    )


def make_env_log(RefEnv, m, n, k, nenv, steps, seed):
    rng = np.random.default_rng(1000 + seed)
    env = RefEnv(m, n, k, nenv, device="cpu")
    env.reset()
    c = m * n
    log = {key: [] for key in ("actions", "active", "reset", "planes", "cp", "mc", "rewards", "dones", "mask")}
    for t in range(steps):
        mask = env.observe()["action_mask"].numpy()
        # mostly legal moves, some arbitrary cells (occupied / negative): the reference accepts both
        acts = np.zeros(nenv, dtype=np.int64)
        for i in range(nenv):
            legal = np.nonzero(mask[i])[0]
            if len(legal) and rng.random() > 0.1:
                acts[i] = rng.choice(legal)
            else:
                acts[i] = rng.integers(-c, c)
        active = np.ones(nenv, dtype=bool)
        if t % 3 == 1:  # G2: subset step over an ascending index list
            active = rng.random(nenv) < 0.6
            if not active.any():
                active[0] = True
        idx = torch.from_numpy(np.nonzero(active)[0])
        if active.all():
            obs, rew, done = env.step(torch.from_numpy(acts))
        else:
            obs, rew, done = env.step_subset(torch.from_numpy(acts[active]), idx)
        planes, cp, mc = _state(env, m, n)
        log["actions"].append(acts.astype(np.int32))
        log["active"].append(active)
        log["planes"].append(planes)
        log["cp"].append(cp)
        log["mc"].append(mc)
        log["rewards"].append(rew.numpy().astype(np.int8))
        assert set(np.unique(rew.numpy())) <= {0.0, 1.0}
        log["dones"].append(done.numpy())
        log["mask"].append(pack_cells(obs["action_mask"].numpy(), m, n))
        # reset most of the finished envs; let a few run on past the end of their game
        reset = done.numpy() & (rng.random(nenv) < 0.85)
        log["reset"].append(reset)
        env.reset(torch.from_numpy(np.nonzero(reset)[0]))
    out = {key: np.stack(val) for key, val in log.items()}
    out["geom"] = np.array([m, n, k, nenv, steps], dtype=np.int64)
    return out


def make_selfplay_trace(RefEnv, RefWrap, m, n, k, nenv, steps, opp, seed):
    torch.manual_seed(seed)
    rng = np.random.default_rng(2000 + seed)
    env = RefEnv(m, n, k, nenv, device="cpu")
    wrap = RefWrap(env)
    wrap.set_opponent(OPPONENTS[opp]())
    c = m * n
    log = {key: [] for key in ("agent_actions", "sides", "obs_planes", "obs_mask", "rewards", "terminated",
                               "pending", "planes", "cp", "mc")}

    def snap(obs):
        log["sides"].append(wrap.agent_side.numpy().astype(np.uint8))
        log["obs_planes"].append(pack_boards(obs["observation"].numpy(), m, n))
        log["obs_mask"].append(pack_cells(obs["action_mask"].numpy(), m, n))
        planes, cp, mc = _state(env, m, n)
        log["planes"].append(planes)
        log["cp"].append(cp)
        log["mc"].append(mc)

    obs, _ = wrap.reset()
    snap(obs)
    for t in range(steps):
        mask = obs["action_mask"].numpy()
        acts = np.zeros(nenv, dtype=np.int64)
        for i in range(nenv):
            legal = np.nonzero(mask[i])[0]
            acts[i] = rng.choice(legal) if rng.random() > 0.03 else rng.integers(0, c)
        obs, rew, term, trunc, _ = wrap.step(torch.from_numpy(acts))
        assert not bool(trunc.any())
        assert set(np.unique(rew.numpy())) <= {-1.0, 0.0, 1.0}
        log["agent_actions"].append(acts.astype(np.int32))
        log["rewards"].append(rew.numpy().astype(np.int8))
        log["terminated"].append(term.numpy())
        log["pending"].append(wrap.pending_resets.numpy().copy())
        snap(obs)
    out = {key: np.stack(val) for key, val in log.items()}
    out["geom"] = np.array([m, n, k, nenv, steps], dtype=np.int64)
    return out


def make_edges(RefEnv):
    out = {}
    for name, sc in SCENARIOS.items():
        m, n, k = sc["m"], sc["n"], sc["k"]
        env = RefEnv(m, n, k, 1, device="cpu")
        env.reset()
        for (r, c) in sc["black"]:
            env.boards[0, 0, r, c] = 1
        for (r, c) in sc["white"]:
            env.boards[0, 1, r, c] = 1
        env.current_player[0] = sc["side"]
        env.move_counts[0] = sc["moves_made"]
        rows = []
        for a in sc["plies"]:
            obs, rew, done = env.step(torch.tensor([a]))
            rows.append(np.concatenate([
                env.boards.numpy().reshape(-1).astype(np.int8),
                obs["action_mask"].numpy().reshape(-1).astype(np.int8),
                np.array([rew[0].item(), float(done[0].item()), env.current_player[0].item(),
                          env.move_counts[0].item()]).astype(np.int8),
            ]))
        out[name] = np.stack(rows)  # per ply: boards(2*m*n) | mask(m*n) | reward, done, side, moves
    return out


def make_masked_logits():
    """Masked-categorical head of the reference nets (cnn.py:69-80 = resnet.py:84-95)."""
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference/src")
    for name in [k for k in sys.modules if k.split(".")[0] in ("env", "selfplay", "alg", "utils")]:
        del sys.modules[name]
    from utils.model_export import create_model_from_architecture

    out = {}
    g = torch.Generator().manual_seed(0)
    for arch in ("cnn_b_s", "resnet_b_s"):
        torch.manual_seed(0)
        net = create_model_from_architecture(arch, obs_shape=(2, 9, 9), action_dim=81)
        net.eval()
        # make the policy head non-trivial (the reference initialises it with gain 0.01)
        with torch.no_grad():
            for p in net.parameters():
                if p.dim() > 1:
                    p.mul_(3.0)
        b = 96
        stones = torch.rand(b, 81, generator=g)
        fill = torch.linspace(0.0, 1.0, b).unsqueeze(1)  # from empty to completely full boards
        black = (stones < fill * 0.5).float()
        white = ((stones >= fill * 0.5) & (stones < fill)).float()
        obs = torch.stack([black, white], dim=1).reshape(b, 2, 9, 9)
        mask = (black + white).reshape(b, 81) == 0
        mask[-1] = False  # an all-masked row -> uniform (cnn.py:76-77)
        with torch.no_grad():
            raw, _ = net(obs, None)
            masked, value = net(obs, mask)
        out[arch + "_raw_logits"] = raw.logits.numpy()
        out[arch + "_mask"] = mask.numpy()
        out[arch + "_masked_logits"] = masked.logits.numpy()
        out[arch + "_probs"] = masked.probs.numpy()
    for name in [k for k in sys.modules if k.split(".")[0] in ("env", "selfplay", "alg", "utils")]:
        del sys.modules[name]
    return out


def more_boards():
    RefEnv, RefWrap, _ = import_reference()
    for (m, n, k, nenv, steps, seeds) in MORE_ENV_CASES:
        for seed in seeds:
            path = os.path.join(OUT, f"boards_env_{m}x{n}x{k}_s{seed}.npz")
            np.savez_compressed(path, **make_env_log(RefEnv, m, n, k, nenv, steps, seed))
            print("wrote", os.path.basename(path), os.path.getsize(path))
    for (m, n, k, nenv, steps, opp, seed) in MORE_SELFPLAY_CASES:
        path = os.path.join(OUT, f"boards_selfplay_{m}x{n}x{k}_{opp}_s{seed}.npz")
        np.savez_compressed(path, **make_selfplay_trace(RefEnv, RefWrap, m, n, k, nenv, steps, opp, seed))
        print("wrote", os.path.basename(path), os.path.getsize(path))


def main():
    if "--more-boards" in sys.argv:  # only the round-4 additions: the older files are left as they are
        return more_boards()
    RefEnv, RefWrap, _ = import_reference()
    for (m, n, k, nenv, steps, seeds) in ENV_CASES:
        for seed in seeds:
            path = os.path.join(OUT, f"env_{m}x{n}x{k}_s{seed}.npz")
            np.savez_compressed(path, **make_env_log(RefEnv, m, n, k, nenv, steps, seed))
            print("wrote", os.path.basename(path), os.path.getsize(path))
    for (m, n, k, nenv, steps, opp, seed) in SELFPLAY_CASES:
        path = os.path.join(OUT, f"selfplay_{m}x{n}x{k}_{opp}_s{seed}.npz")
        np.savez_compressed(path, **make_selfplay_trace(RefEnv, RefWrap, m, n, k, nenv, steps, opp, seed))
        print("wrote", os.path.basename(path), os.path.getsize(path))
    path = os.path.join(OUT, "edges.npz")
    np.savez_compressed(path, **make_edges(RefEnv))
    print("wrote", os.path.basename(path), os.path.getsize(path))
    path = os.path.join(OUT, "masked_logits.npz")
    np.savez_compressed(path, **make_masked_logits())
    print("wrote", os.path.basename(path), os.path.getsize(path))


if __name__ == "__main__":
    main()

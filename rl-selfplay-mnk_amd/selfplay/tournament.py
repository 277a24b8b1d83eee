"""Head-to-head games on the raw HIP env, without the self-play wrapper -- the hot loop of the reference's
tournament runner (``/root/reference/src/model_comparison/match_runner.py:125-218``,
``MatchRunner._play_batch_games``).

Same bookkeeping (policy 1 as black or as white, every mover sees itself in channel 0, a game counts once --
win / loss for policy 1 by who made the winning ply, draw otherwise), different execution: the reference
rebuilds ``nonzero`` index lists, boolean-masked observation subsets and ``.any()`` / ``.item()`` checks every
ply (5+ host synchronisations per ply).  All games of a batch start together and alternate in lockstep, so
here ply p is one ``mnk_observe`` (mover's view) + one policy call on the full batch + one ``mnk_step``; games
that are over are simply no longer counted.  The loop runs the fixed m*n plies and synchronises once.
"""
import torch

from env.torch_vector_mnk_env import TorchVectorMnkEnv


def play_batch_games(p1_policy, p2_policy, mnk_config, n_games: int, p1_is_black: bool, device="cuda"):
    """Returns (wins, losses, draws) of policy 1 over ``n_games`` parallel games."""
    if n_games == 0:
        return 0, 0, 0
    m, n, k = mnk_config
    env = TorchVectorMnkEnv(m, n, k, num_envs=n_games, device=device)
    dev = env._dev
    env.reset()
    p1_side = 0 if p1_is_black else 1
    over = torch.zeros(n_games, dtype=torch.bool, device=dev)
    wins = torch.zeros((), dtype=torch.long, device=dev)
    losses = torch.zeros((), dtype=torch.long, device=dev)
    draws = torch.zeros((), dtype=torch.long, device=dev)
    obs = torch.empty((n_games, 2, m, n), dtype=torch.float32, device=dev)
    mask = torch.empty((n_games, m * n), dtype=torch.bool, device=dev)
    rewards = torch.empty(n_games, dtype=torch.float32, device=dev)
    dones = torch.empty(n_games, dtype=torch.bool, device=dev)
    side = torch.empty(n_games, dtype=torch.long, device=dev)
    for ply in range(m * n):
        mover = ply & 1  # every game still running is at the same ply
        side.fill_(mover)
        env.observe_into(obs, mask, flip_side=side)  # match_runner.py:163-193: the mover sees itself in channel 0
        policy = p1_policy if mover == p1_side else p2_policy
        with torch.no_grad():
            actions = policy.act({"observation": obs, "action_mask": mask}, deterministic=False)
        actions = actions.to(torch.long).contiguous()
        env.step_into(actions, rewards, dones)
        fresh = dones & ~over  # match_runner.py:200-213
        won = fresh & (rewards == 1.0)
        if mover == p1_side:
            wins += won.sum()
        else:
            losses += won.sum()
        draws += (fresh & (rewards == 0.0)).sum()
        over |= dones
    assert bool(over.all()), "a game outlived m*n plies"
    return int(wins.item()), int(losses.item()), int(draws.item())


def play_match(p1_policy, p2_policy, mnk_config, games_per_pair: int, device="cuda"):
    """Half the games with policy 1 as black, half as white (match_runner.py:92-123).
    Returns dict(wins, losses, draws, score) for policy 1."""
    first = games_per_pair // 2
    w1, l1, d1 = play_batch_games(p1_policy, p2_policy, mnk_config, first, True, device)
    w2, l2, d2 = play_batch_games(p1_policy, p2_policy, mnk_config, games_per_pair - first, False, device)
    wins, losses, draws = w1 + w2, l1 + l2, d1 + d2
    return {"wins": wins, "losses": losses, "draws": draws,
            "score": (wins + 0.5 * draws) / max(1, games_per_pair)}

"""Agent-vs-benchmark validation on the HIP env -- the caller the reference has in
``/root/reference/src/selfplay/validation.py:6-44`` (``validate_gpu``), same arguments, same result keys.

What the reference computes: ``n_episodes`` envs, the first half with the agent as black and the second half as
white (:14-15), one game per env -- the reward of the step that first terminates an env is that env's result
(:28-32) -- and the shares of +1 / -1 / 0 results (:34-44).  It polls ``active_mask.any()`` on the host after every
step.  A game ends within ceil(m*n/2) + 1 agent steps, so here the loop has that fixed length, the per-env results
accumulate on the device, and the host synchronises once, when it reads the three counts.
"""
import torch

from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper

_KEY = "validation/vs_benchmark/"


def first_episode_results(wrapper, agent_policy, obs, max_agent_steps: int):
    """Plays ``max_agent_steps`` steps from ``obs`` and returns, per env, the reward of the step that first
    terminated it (f32 [N]: +1 win, -1 loss, 0 draw) and a bool [N] that is still set where no game ended.  Envs
    restart after their first game (autoreset) and keep playing; those later games do not count.  No host
    synchronisation."""
    n = wrapper.num_envs
    result = torch.zeros(n, dtype=torch.float32, device=wrapper._dev)
    open_games = torch.ones(n, dtype=torch.bool, device=wrapper._dev)
    for _ in range(max_agent_steps):
        with torch.no_grad():
            actions = agent_policy.act(obs, deterministic=False)  # validation.py:24
        obs, rewards, terminated, _, _ = wrapper.step(actions)
        result = torch.where(open_games & terminated, rewards, result)
        open_games &= ~terminated
    return result, open_games


def validate_gpu(agent_policy, opponent_policy, mnk_config, n_episodes=1024, device="cuda"):
    m, n, k = mnk_config
    wrapper = TorchSelfPlayWrapper(TorchVectorMnkEnv(m, n, k, num_envs=n_episodes, device=device))
    wrapper.set_opponent(opponent_policy)
    sides = (torch.arange(n_episodes, device=device) >= n_episodes // 2).to(torch.long)  # black first, then white
    obs, _ = wrapper.reset(options={"agent_side": sides})
    result, unfinished = first_episode_results(wrapper, agent_policy, obs, (m * n + 1) // 2 + 1)
    counts = [(result == x).sum() for x in (1.0, -1.0, 0.0)] + [unfinished.sum()]
    wins, losses, draws, still_open = (int(v) for v in torch.stack(counts).tolist())  # the one synchronisation
    if still_open:
        raise RuntimeError("validate_gpu: a game outlived ceil(m*n/2)+1 agent steps")
    return {
        _KEY + "win_rate": wins / n_episodes,
        _KEY + "loss_rate": losses / n_episodes,
        _KEY + "draw_rate": draws / n_episodes,
        _KEY + "score_rate": (wins + 0.5 * draws) / n_episodes,
        _KEY + "games_played": n_episodes,
    }

"""Golden fixture G6: a reference network end to end (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_policy.py

Plays 64 reference envs (9x9x5) to mid-game through the reference wrapper, takes the agent's canonical
observation, and runs the reference's ``cnn_b_s`` (``alg/architectures/configs.py:49-56``, seed-0 init, eval
mode, fp32) on it.  Stored: the network's parameters and buffers (data), the position (packed planes, sides),
the canonical observation (packed) and mask, and what the reference computed: masked log-probabilities, values,
argmax actions.  No reference source is stored."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.packing import pack_boards, pack_cells  # noqa: E402

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
from env.torch_vector_mnk_env import TorchVectorMnkEnv  # noqa: E402
from selfplay.policy import RandomPolicy  # noqa: E402
from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper  # noqa: E402
from utils.model_export import create_model_from_architecture  # noqa: E402


def main():
    torch.manual_seed(0)
    m, n, k, nenv = 9, 9, 5, 64
    net = create_model_from_architecture("cnn_b_s", obs_shape=(2, m, n), action_dim=m * n).eval()
    env = TorchVectorMnkEnv(m, n, k, nenv, device="cpu")
    wrap = TorchSelfPlayWrapper(env)
    wrap.set_opponent(RandomPolicy(m * n))
    sides = torch.arange(nenv) % 2
    obs, _ = wrap.reset(options={"agent_side": sides})
    agent = RandomPolicy(m * n)
    for t in range(14):  # 28-29 stones on the board, no autoreset in flight
        obs, rew, term, _, _ = wrap.step(agent.act(obs))
    obs = wrap.get_agent_obs()
    with torch.no_grad():
        dist, value = net(obs["observation"], obs["action_mask"])
    out = {f"param/{k}": v.numpy() for k, v in net.state_dict().items()}
    out.update(
        planes=pack_boards(env.boards.numpy(), m, n), meta_side=env.current_player.numpy().astype(np.uint8),
        meta_moves=env.move_counts.numpy().astype(np.int32), agent_side=wrap.agent_side.numpy().astype(np.uint8),
        obs_planes=pack_boards(obs["observation"].numpy(), m, n), obs_mask=pack_cells(obs["action_mask"].numpy(), m, n),
        logp=dist.logits.numpy(), value=value.numpy(), argmax=torch.argmax(dist.logits, dim=1).numpy(),
        pending=wrap.pending_resets.numpy(),
    )
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "policy_cnn_b_s.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", int(wrap.pending_resets.sum()), "envs awaiting reset")


if __name__ == "__main__":
    main()

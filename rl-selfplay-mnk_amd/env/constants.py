"""Player / channel / reward constants of the MNK env.

Same names and values as the reference module (``src/env/constants.py:1-9``), because callers
import them by name; the HIP kernels hard-wire the same encoding (side bit 0 = black, channel 0 =
the viewer's own stones, rewards exactly +1.0 / -1.0 / 0.0).
"""
PLAYER_BLACK, PLAYER_WHITE = 0, 1          # bit 0 of the packed meta word
CHANNEL_ME, CHANNEL_ENEMY = 0, 1           # observation planes as the viewer sees them
REWARD_WIN, REWARD_LOSS, REWARD_DRAW = 1.0, -1.0, 0.0

"""Constants of the MNK env -- same names and values as the reference's
``src/env/constants.py:1-9`` so callers can import them unchanged."""
PLAYER_BLACK = 0
PLAYER_WHITE = 1

CHANNEL_ME = 0
CHANNEL_ENEMY = 1

REWARD_WIN = 1.0
REWARD_LOSS = -1.0
REWARD_DRAW = 0.0

"""Bounded pool of frozen opponents, uniform choice -- the host-side bookkeeping of
``/root/reference/src/selfplay/opponent_pool.py:5-19`` (``train.py:98-114,123`` drives it)."""
import random
from collections import deque


class OpponentPool:
    def __init__(self, max_size=5):
        self.max_size = max_size
        self.pool = deque(maxlen=max_size)  # the oldest opponent falls out when full

    def add_opponent(self, opponent):
        self.pool.append(opponent)

    def get_random_opponent(self):
        return random.choice(self.pool) if self.pool else None

    def size(self):
        return len(self.pool)

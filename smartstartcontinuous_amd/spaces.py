"""Minimal stand-ins for the gym 0.10.5 objects the reference touches on the hot path
(gym itself is not a dependency): ``Box`` (``observation_space`` / ``action_space``:
``.low .high .shape``, used at DDPG_Baselines_agent.py:152-157,238, NND_MB_agent.py:500-501,
policy_random.py:9-11) and ``EnvSpec`` (``env.spec.id``, rlTrain.py:52-54)."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            low = np.asarray(low, dtype=dtype)
            high = np.asarray(high, dtype=dtype)
            shape = low.shape
        else:
            low = np.full(shape, low, dtype=dtype)
            high = np.full(shape, high, dtype=dtype)
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

    def sample(self):
        return np.random.uniform(self.low, self.high, self.shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())

    def __repr__(self):
        return f"Box{self.shape}"


class EnvSpec:
    def __init__(self, id, max_episode_steps=None):
        self.id = id
        self.max_episode_steps = max_episode_steps

    def __repr__(self):
        return f"EnvSpec({self.id})"

"""test stand-in for gymnasium (see ../README.md)"""
import numpy as np
from . import spaces, envs   # noqa: F401
from .envs.registration import register, make, registry   # noqa: F401


class Env:
    metadata = {}
    observation_space = None
    action_space = None
    np_random = None

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self.np_random = np.random.default_rng(seed)

    def close(self):
        pass

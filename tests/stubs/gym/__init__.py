"""test stand-in for classic gym (see ../README.md)"""
from . import spaces, utils, envs   # noqa: F401
from .envs.registration import register, make, registry   # noqa: F401


class Env:
    metadata = {}
    observation_space = None
    action_space = None

    def close(self):
        pass

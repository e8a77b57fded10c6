"""Gymnasium adapters over projectd_env (reference pyprojectd/projectd_gymnasium: id "ProjectD-v0", max_episode_steps 80000).
gymnasium is not a dependency of the package: importing this module without it raises ImportError with that message.
  ProjectDGymnasium     gymnasium.Env, one car, the reference signatures
  ProjectDGymnasiumVec  gymnasium.vector.VectorEnv-shaped batch (N lanes, one kernel launch per step, auto-reset)"""
import numpy as np
try:
    import gymnasium as gym
    from gymnasium import spaces
except ImportError as e:   # pragma: no cover
    raise ImportError('projectd_gymnasium needs the gymnasium package (not installed in this image)') from e

import projectd_env as E
import projectd_adapters as A

MAX_EPISODE_STEPS = A.MAX_EPISODE_STEPS


def _boxes(cfg):
    return A.boxes(cfg, spaces.Box)


class ProjectDGymnasium(A.SingleCar, gym.Env):
    def __init__(self, base_dir=None, **settings):
        gym.Env.__init__(self)
        self._open(spaces.Box, base_dir, **settings)

    def step(self, action):
        return self.advance(action)

    def reset(self, *, seed=None, options=None):
        super().reset(seed=seed)
        return self.restart(), {}

    def render(self):
        self.impl.render()

    def seed(self, seed=None):   # compat with legacy gym (projectd_gymnasium.py:38-40)
        pass


class ProjectDGymnasiumVec:
    """N lanes; step(actions[N,2]) -> obs[N,24], reward[N], terminated[N], truncated[N], infos (gymnasium VectorEnv contract,
    autoreset: a finished lane's next observation is the first of its new episode)."""

    def __init__(self, num_envs, base_dir=None, device=0, **settings):
        self.impl = E.ProjectDVecEnv(num_envs, base_dir, device=device, auto_reset=True, **settings)
        self.num_envs = num_envs
        obs_box, act_box = _boxes(self.impl.cfg)
        self.single_observation_space, self.single_action_space = obs_box, act_box
        self.observation_space = spaces.Box(low=np.tile(obs_box.low, (num_envs, 1)), high=np.tile(obs_box.high, (num_envs, 1)), dtype=np.float32)
        self.action_space = spaces.Box(low=np.tile(act_box.low, (num_envs, 1)), high=np.tile(act_box.high, (num_envs, 1)), dtype=np.float32)

    def reset(self, *, seed=None, options=None):
        return self.impl.reset(), {}

    def step(self, actions):
        obs, reward, terminated, truncated, info = self.impl.step(actions)
        too_long = self.impl.step_id >= MAX_EPISODE_STEPS
        if too_long.any():
            truncated = truncated | too_long
            self.impl.reset(too_long)
        return obs, reward, terminated, truncated, info

    def close(self):
        self.impl.close()


def register():
    from gymnasium.envs.registration import register as _register
    _register(id=A.ENV_ID, entry_point='projectd_gymnasium:ProjectDGymnasium', max_episode_steps=MAX_EPISODE_STEPS)

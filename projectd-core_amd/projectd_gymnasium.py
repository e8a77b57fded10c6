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

MAX_EPISODE_STEPS = 80000


def _boxes(cfg):
    lo, hi = E.obs_bounds(cfg)
    return spaces.Box(low=lo, high=hi, dtype=np.float32), spaces.Box(low=np.array([-1, -1], np.float32), high=np.array([1, 1], np.float32), dtype=np.float32)


class ProjectDGymnasium(gym.Env):
    def __init__(self, base_dir=None, **settings):
        super().__init__()
        self.impl = E.ProjectDEnv(base_dir, **settings)
        self.observation_space, self.action_space = _boxes(self.impl.cfg)

    def step(self, action):
        obs, reward, terminated, truncated, info = self.impl.step(action)
        truncated = truncated or self.impl.step_id >= MAX_EPISODE_STEPS
        return obs, reward, terminated, truncated, info

    def reset(self, *, seed=None, options=None):
        super().reset(seed=seed)
        return self.impl.reset(), {}

    def render(self):
        self.impl.render()

    def close(self):
        self.impl.close()

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
    _register(id='ProjectD-v0', entry_point='projectd_gymnasium:ProjectDGymnasium', max_episode_steps=MAX_EPISODE_STEPS)

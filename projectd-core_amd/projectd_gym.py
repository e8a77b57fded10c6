"""Classic-gym adapter over projectd_env (reference pyprojectd/projectd_gym/projectd_gym.py:10-42 and __init__.py:1-8):
`ProjectDEnvGym(gym.Env)` with the 4-tuple step / bare reset of gym < 0.26, registered as "ProjectD-v0" with
max_episode_steps = 80000.  gym is not a dependency of the package: importing this module without it raises ImportError."""
import numpy as np
try:
    import gym
    from gym import spaces as gym_spaces
    from gym.utils import seeding as gym_seeding
except ImportError as e:   # pragma: no cover
    raise ImportError('projectd_gym needs the gym package (not installed in this image)') from e

import projectd_env as E

MAX_EPISODE_STEPS = 80000


class ProjectDEnvGym(gym.Env):
    def __init__(self, **kwargs):
        super().__init__()
        self.seed()
        self.impl = E.ProjectDEnv(**kwargs)
        obs_low, obs_high = self.impl._get_obs_space()
        a_low, a_high = self.impl._get_action_space()
        self.observation_space = gym_spaces.Box(low=obs_low, high=obs_high, dtype=np.float32)
        self.action_space = gym_spaces.Box(low=a_low, high=a_high, dtype=np.float32)

    def close(self):
        self.impl.close()

    def step(self, action):
        state, reward, terminate, truncate, info = self.impl.step(action)
        return state, reward, (terminate or truncate), {}

    def reset(self):
        return self.impl.reset()

    def render(self, mode='human'):
        self.impl.render()

    def seed(self, seed=None):
        self.np_random, seed = gym_seeding.np_random(seed)
        return [seed]


def register():
    from gym.envs.registration import register as _register
    _register(id='ProjectD-v0', entry_point='projectd_gym:ProjectDEnvGym', max_episode_steps=MAX_EPISODE_STEPS)

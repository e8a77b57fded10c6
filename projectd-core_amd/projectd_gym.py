"""Classic-gym face of the env (drop-in for pyprojectd/projectd_gym: class ProjectDEnvGym, id "ProjectD-v0"): gym < 0.26 conventions --
step returns (obs, reward, done, info), reset returns the observation, seed returns the list of seeds used.  Everything else comes from
projectd_adapters.SingleCar.  gym is not a dependency of the package: importing this module without it raises ImportError."""
try:
    import gym
    from gym import spaces as gym_spaces
    from gym.utils import seeding as gym_seeding
except ImportError as e:   # pragma: no cover
    raise ImportError('projectd_gym needs the gym package (not installed in this image)') from e

import projectd_adapters as A

MAX_EPISODE_STEPS = A.MAX_EPISODE_STEPS


class ProjectDEnvGym(A.SingleCar, gym.Env):
    metadata = {'render.modes': ['human']}

    def __init__(self, **settings):
        gym.Env.__init__(self)
        self.seed()
        self._open(gym_spaces.Box, **settings)

    def step(self, action):
        obs, reward, terminated, truncated, _ = self.advance(action)
        return obs, reward, terminated or truncated, {}

    def reset(self):
        return self.restart()

    def seed(self, seed=None):
        self.np_random, used = gym_seeding.np_random(seed)
        return [used]


def register():
    from gym.envs.registration import register as _register
    _register(id=A.ENV_ID, entry_point='projectd_gym:ProjectDEnvGym', max_episode_steps=MAX_EPISODE_STEPS)

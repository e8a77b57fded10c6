"""What the gym / gymnasium / SB3 adapters share: the single-car core (spaces from the env's bounds, the 80000-step episode limit the
reference registers its env with, pyprojectd/projectd_gym/__init__.py:1-8) and the space builders.  The adapter modules add only their
framework's calling convention on top."""
import numpy as np
import projectd_env as E

MAX_EPISODE_STEPS = 80000
ENV_ID = 'ProjectD-v0'


def boxes(cfg, box_cls):
    """(observation box, action box) of one car in the given framework's Box class"""
    lo, hi = E.obs_bounds(cfg)
    one = np.ones(2, np.float32)
    return box_cls(low=lo, high=hi, dtype=np.float32), box_cls(low=-one, high=one, dtype=np.float32)


class SingleCar:
    """one ProjectDEnv behind a framework adapter: `advance` = env.step with the registered episode limit folded into `truncated`"""

    def _open(self, box_cls, base_dir=None, **settings):
        self.impl = E.ProjectDEnv(base_dir, **settings)
        self.observation_space, self.action_space = boxes(self.impl.cfg, box_cls)

    def advance(self, action):
        obs, reward, terminated, truncated, info = self.impl.step(action)
        return obs, reward, bool(terminated), bool(truncated or self.impl.step_id >= MAX_EPISODE_STEPS), info

    def restart(self):
        return self.impl.reset()

    def render(self, mode='human'):
        self.impl.render()

    def close(self):
        self.impl.close()

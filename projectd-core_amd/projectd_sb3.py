"""Stable-Baselines3-shaped vector env over the batched stepper (reference training setup: rl_zoo3 / SB3 SAC on "ProjectD-v0",
pyprojectd/hyperparams/sac.yml: `normalize: {norm_obs: True, norm_reward: False}`).

  ProjectDSB3VecEnv   the stable_baselines3.common.vec_env.VecEnv interface (step_async / step_wait / reset / get_attr / set_attr /
                      env_method / env_is_wrapped / seed / close, num_envs, observation_space, action_space) over ProjectDVecEnv:
                      N lanes, one kernel launch per step.  Subclasses SB3's VecEnv when SB3 is importable (so VecNormalize,
                      VecMonitor ... wrap it), and is a plain class with the same methods otherwise.
  RunningObsNorm      SB3 VecNormalize's observation normalisation (RunningMeanStd update, clip 10, epsilon 1e-8) for use without
                      SB3; `wrap_normalize(env)` returns SB3's VecNormalize when available, else this.

Episode ends follow SB3's convention (same-step reset, the default here): on the step where a lane's episode ends, step_wait() returns for it
done = True, the terminal reward, the terminal observation in infos[i]["terminal_observation"] and, as observation, the NEW episode's first one --
the lane's reset tick (teleport + step([0, 0]), reference projectd_env.py:216-227) is run before the step returns, with every other lane held
(pdb_step_host_held: one more launch in which only the workgroups of the lanes that reset do anything).  Time limits (max_episode_steps 80000)
end an episode the same way, with infos[i]["TimeLimit.truncated"] = True.  Every stored transition is a real one; VecNormalize sees each
observation once.
same_step_reset=False keeps the kernel's free next-step form (gymnasium's AutoresetMode.NEXT_STEP): the step after an episode end is the lane's
reset tick -- it returns the new episode's first observation with reward 0 and done False and is flagged infos[i]["reset_tick"] = True for a
buffer that wants to drop it; on the step that ends the episode the observation returned is the terminal one."""
import numpy as np
import projectd_env as E

try:   # pragma: no cover  (SB3 is not in this image)
    from stable_baselines3.common.vec_env import VecEnv as _Base, VecNormalize as _VecNormalize
    from gymnasium import spaces as _spaces
except ImportError:
    _Base, _VecNormalize, _spaces = object, None, None

MAX_EPISODE_STEPS = 80000


class _Box:
    """the two attributes SB3 reads from a Box when gymnasium is absent"""
    def __init__(self, low, high, dtype=np.float32):
        self.low, self.high, self.dtype, self.shape = low, high, np.dtype(dtype), low.shape


def _box(low, high):
    return _spaces.Box(low=low, high=high, dtype=np.float32) if _spaces is not None else _Box(low, high)


class ProjectDSB3VecEnv(_Base):
    def __init__(self, num_envs, base_dir=None, device=0, max_episode_steps=MAX_EPISODE_STEPS, same_step_reset=True, **settings):
        self.impl = E.ProjectDVecEnv(num_envs, base_dir, device=device, auto_reset=True, same_step_reset=same_step_reset, **settings)
        self.same_step_reset = bool(same_step_reset)
        lo, hi = E.obs_bounds(self.impl.cfg)
        obs_space, act_space = _box(lo, hi), _box(np.array([-1, -1], np.float32), np.array([1, 1], np.float32))
        if _Base is not object:
            super().__init__(num_envs, obs_space, act_space)
        else:
            self.num_envs, self.observation_space, self.action_space = num_envs, obs_space, act_space
        self.max_episode_steps = max_episode_steps
        self.render_mode = None
        self._actions = None

    # ---- VecEnv interface ----
    def reset(self):
        return self.impl.reset()

    def step_async(self, actions):
        self._actions = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, 2)

    def step_wait(self):
        obs, reward, terminated, truncated, info = self.impl.step(self._actions)
        too_long = (self.impl.step_id >= self.max_episode_steps) & ~terminated
        terminal = dict(info.get('terminal_observation', {}))   # same-step mode: the lanes that terminated have taken their reset tick already
        if too_long.any():
            self.impl.reset(too_long)   # teleport now; the reset tick on the lanes' next step -- or, same-step, right away
            if self.same_step_reset:
                extra = {}
                obs = self.impl._reset_tick_now(too_long, obs, extra)
                terminal.update(extra['terminal_observation'])
        done = terminated | too_long
        reset_tick = self.impl.last_reset_tick if hasattr(self.impl, 'last_reset_tick') else np.zeros(self.num_envs, bool)
        infos = [{} for _ in range(self.num_envs)]
        for i in np.nonzero(done)[0]:
            infos[i]['terminal_observation'] = terminal[int(i)] if int(i) in terminal else obs[i].copy()
            infos[i]['TimeLimit.truncated'] = bool(too_long[i])
        for i in np.nonzero(reset_tick)[0]:
            infos[i]['reset_tick'] = True
        return obs, reward, done, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.impl.close()

    def get_attr(self, attr_name, indices=None):
        return [getattr(self.impl.cfg, attr_name) if hasattr(self.impl.cfg, attr_name) else getattr(self.impl, attr_name) for _ in self._indices(indices)]

    def set_attr(self, attr_name, value, indices=None):
        raise AttributeError('the lanes of a batch share one configuration: construct the env with %s=...' % attr_name)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        raise AttributeError('no per-lane env objects behind a batch')

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False for _ in self._indices(indices)]

    def seed(self, seed=None):
        import PyProjectD as pd
        if seed is not None:
            pd.setBatchSeeds(self.impl.batch, (np.arange(self.num_envs, dtype=np.uint32) + np.uint32(seed)))   # one C-runtime rand() per lane
        return [None if seed is None else seed + i for i in range(self.num_envs)]

    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        return [indices] if isinstance(indices, int) else indices


class RunningObsNorm:
    """VecNormalize(norm_obs=True, norm_reward=False) without SB3: same running mean / variance update (parallel-variance merge of
    each batch), same clipping; `training=False` freezes the statistics (evaluation)."""
    def __init__(self, venv, clip_obs=10.0, epsilon=1e-8):
        self.venv, self.clip_obs, self.epsilon, self.training = venv, clip_obs, epsilon, True
        n = venv.observation_space.shape[0]
        self.mean, self.var, self.count = np.zeros(n, np.float64), np.ones(n, np.float64), 1e-4
        self.num_envs, self.observation_space, self.action_space = venv.num_envs, venv.observation_space, venv.action_space

    def _update(self, x):
        bm, bv, bc = x.mean(axis=0), x.var(axis=0), x.shape[0]
        delta, tot = bm - self.mean, self.count + bc
        m2 = self.var * self.count + bv * bc + np.square(delta) * self.count * bc / tot
        self.mean, self.var, self.count = self.mean + delta * bc / tot, m2 / tot, tot

    def normalize_obs(self, obs):
        return np.clip((obs - self.mean) / np.sqrt(self.var + self.epsilon), -self.clip_obs, self.clip_obs).astype(np.float32)

    def reset(self):
        obs = self.venv.reset()
        if self.training:
            self._update(obs)
        return self.normalize_obs(obs)

    def step(self, actions):
        obs, reward, done, infos = self.venv.step(actions)
        if self.training:
            self._update(obs)
        for i in np.nonzero(done)[0]:
            infos[i]['terminal_observation'] = self.normalize_obs(infos[i]['terminal_observation'])
        return self.normalize_obs(obs), reward, done, infos

    def close(self):
        self.venv.close()


def wrap_normalize(venv, **kw):
    """sac.yml's `normalize`: SB3's VecNormalize when SB3 is installed, RunningObsNorm otherwise"""
    if _VecNormalize is not None and isinstance(venv, _Base):   # pragma: no cover
        return _VecNormalize(venv, norm_obs=True, norm_reward=False, **kw)
    return RunningObsNorm(venv, **kw)

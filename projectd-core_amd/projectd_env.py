"""Environment surface of the batched stepper, mirroring reference pyprojectd/projectd_env.py (ProjectDEnv).

Two classes over the PyProjectD-compatible module built from csrc/pybind/pyprojectd.cpp:

  ProjectDVecEnv  N cars = N independent (simulator, car) pairs of the reference, one kernel launch per tick.
                  step(actions[N,2]) -> obs[N,24], reward[N], terminated[N], truncated[N], info.  The per-lane reward /
                  termination bookkeeping is the reference's (projectd_env.py:173-206): stepReward minus 50 on collision /
                  off-track / stuck, terminate when the episode's cumulative reward falls below -200.
  ProjectDEnv     the reference's single-env signature (step(action[2]) -> obs[24], reward, terminated, truncated, {}),
                  driven through the classic per-simulator calls -- the unmodified reference env works the same way.

Observation layout (24 float32, projectd_env.py:239-273): localVelocity xyz, localAngularVelocity xyz, tyreNdSlip[4],
bodyVsTrack, velocityVsTrack, lookAhead[5], probes[0..6].  Action: a0 = steer, a1 -> gas = linscale(a1,-1,1,0.1,1).
Nothing here computes physics; there is no CPU fallback (stepping without a GPU fails in the module)."""
import math, os, sys
import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, PKG)
import PyProjectD as pd  # noqa: E402  (in-tree pybind11 module; raises ImportError if it was not built)

SIM_DT = 1.0 / 333.0
OBS_DIM = 24
FLAG_COLLISION, FLAG_OFFTRACK, FLAG_STUCK = 1, 2, 4    # pdb_step_out.flags (include/pdb_types.h)

DEFAULT_SCORING = {
    'SmoothSteerSpeed': 10.0, 'MinBonusSpeed': 5.0, 'MaxBonusSpeed': 200.0, 'StallRpm': 300.0, 'DirectionThreshold': 0.75,
    'OutOfTrackThreshold': 0.51, 'ApproachDistance': 3.5, 'CriticalDistance': 2.0, 'TravelBonus': 0.1, 'TravelSplineBonus': 0.01,
    'DriftBonus': 0.0, 'SpeedBonus': 0.0, 'ThrottleBonus': 0.0, 'EngineRpmBonus': 0.0, 'DirectionBonus': 0.0, 'DirectionPenalty': 0.0,
    'ObstApproachPenalty': 0.0, 'CollisionPenalty': 0.0, 'OffTrackPenalty': 0.0, 'GearGrindPenalty': 0.0, 'StallPenalty': 0.0,
}
DEFAULT_TUNES = {'ks_toyota_ae86_drift': {'FRONT_BIAS': 55.0, 'DIFF_POWER': 30.0, 'DIFF_COAST': 30.0, 'FINAL_RATIO': 5.0,
                                          'PRESSURE_LF': 28.0, 'PRESSURE_RF': 28.0, 'PRESSURE_LR': 28.0, 'PRESSURE_RR': 28.0}}


class EnvConfig:
    """the class attributes of the reference env, as one settings object shared by both env classes"""
    track_name = 'driftplayground'
    car_model = 'ks_toyota_ae86_drift'
    smooth_controls = True
    auto_clutch = True
    auto_shift = True
    auto_blip = True
    range_velocity = 100
    range_angularVelocity = 100
    range_tyreNdSlip = 10
    range_lookAhead = math.pi
    range_probe = 50
    terminate_on_hit = True
    terminate_off_track = True
    terminate_when_stuck = True
    terminate_hit_penalty = 50.0
    terminate_off_track_penalty = 50.0
    terminate_stuck_penalty = 50.0
    terminate_low_reward = -200.0
    stuck_timeout = 5.0
    teleport_mode = 0
    teleport_on_reset = True
    teleport_on_hit = False
    teleport_off_track = False
    min_gas = 0.1
    max_gas = 1.0

    def __init__(self, **kw):
        self.scoring_vars = dict(DEFAULT_SCORING)
        self.car_tunes = {k: dict(v) for k, v in DEFAULT_TUNES.items()}
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError('unknown env setting %r' % k)
            setattr(self, k, v)


def obs_bounds(cfg):
    hi = np.array([cfg.range_velocity] * 3 + [cfg.range_angularVelocity] * 3 + [cfg.range_tyreNdSlip] * 4 + [1.0, 1.0] +
                  [cfg.range_lookAhead] * 5 + [cfg.range_probe] * 7, dtype=np.float32)
    lo = -hi.copy()
    lo[6:10] = 0.0      # tyreNdSlip >= 0
    lo[17:24] = 0.0     # probe distances >= 0
    return lo, hi


def _configure_simulator(cfg, base_dir):
    """projectd_env.py:118-136: createSimulator, loadTrack, addCar, teleport, assists, tunes, scoring vars"""
    sim = pd.createSimulator(base_dir)
    if sim < 0:
        raise RuntimeError('createSimulator(%r) failed' % base_dir)
    pd.loadTrack(sim, cfg.track_name)
    car = pd.addCar(sim, cfg.car_model)
    if car < 0:
        pd.destroySimulator(sim)
        raise RuntimeError('addCar(%r) failed (track %r)' % (cfg.car_model, cfg.track_name))
    pd.teleportCarByMode(sim, car, cfg.teleport_mode)
    pd.setCarAutoTeleport(sim, car, cfg.teleport_on_hit, cfg.teleport_off_track, cfg.teleport_mode)
    pd.setCarAssists(sim, car, cfg.auto_clutch, cfg.auto_shift, cfg.auto_blip)
    for name, value in cfg.car_tunes.get(cfg.car_model, {}).items():
        pd.setCarTune(sim, car, name, value)
    for name, value in cfg.scoring_vars.items():
        pd.setScoringVar(sim, car, name, value)
    ctl = pd.CarControls()
    pd.setCarControls(sim, car, cfg.smooth_controls, ctl)
    return sim, car


class ProjectDVecEnv:
    def __init__(self, num_envs, base_dir=None, device=0, auto_reset=True, same_step_reset=False, **settings):
        """same_step_reset (with auto_reset): a lane whose episode ends in a step() takes its reset tick -- teleport + step([0, 0]),
        projectd_env.py:216-227 -- before that step() returns, every other lane held (pdb_step_host_held): the step returns done = True, the terminal
        reward, and the NEW episode's first observation; the terminal observation is in info['terminal_observation'] (a dict lane -> obs).  This is
        stable-baselines3's VecEnv convention.  Default False: the reset tick is the lane's next step (gymnasium's AutoresetMode.NEXT_STEP), at no cost."""
        base_dir = base_dir or default_base_dir()
        self.cfg = EnvConfig(**settings)
        self.num_envs = int(num_envs)
        self.auto_reset = auto_reset
        self.same_step_reset = bool(same_step_reset and auto_reset)
        self.sim, self.car = _configure_simulator(self.cfg, base_dir)
        self.batch = pd.createBatch(self.sim, self.num_envs, device)
        if self.batch < 0:
            pd.destroySimulator(self.sim)
            raise RuntimeError('createBatch failed (no GPU? there is no CPU fallback)')
        if self.cfg.stuck_timeout != 5.0:
            pd.setBatchStuckTimeout(self.batch, float(self.cfg.stuck_timeout))
        self._set_kernel_env(self.auto_reset)
        self.total_reward = np.zeros(self.num_envs, dtype=np.float64)
        self.last_reset_tick = np.zeros(self.num_envs, dtype=bool)
        self.step_id = np.zeros(self.num_envs, dtype=np.int64)
        self.pending_reset = np.zeros(self.num_envs, dtype=bool)
        self.observation_bounds = obs_bounds(self.cfg)
        self.action_bounds = (np.array([-1.0, -1.0], np.float32), np.array([1.0, 1.0], np.float32))

    def _set_kernel_env(self, on):
        """auto_reset: the reward / termination / reset-tick rules below run inside the step kernel (pdb_set_env), step() only unpacks"""
        c = self.cfg
        pd.setBatchEnv(self.batch, bool(on), bool(c.terminate_on_hit), bool(c.terminate_off_track), bool(c.terminate_when_stuck),
                       float(c.terminate_hit_penalty), float(c.terminate_off_track_penalty), float(c.terminate_stuck_penalty), float(c.terminate_low_reward),
                       bool(c.teleport_on_reset), int(c.teleport_mode))
        self.kernel_env = bool(on)

    def close(self):
        if self.batch >= 0:
            pd.destroyBatch(self.batch); self.batch = -1
        if self.sim >= 0:
            pd.destroySimulator(self.sim); self.sim = -1

    def _raw_step(self, actions):
        out = pd.stepBatch(self.batch, np.ascontiguousarray(actions, dtype=np.float32).reshape(self.num_envs, 2), SIM_DT)
        if out.shape != (self.num_envs, 26):
            raise RuntimeError('stepBatch failed')
        return out[:, :OBS_DIM], out[:, 24].astype(np.float64), out[:, 25].view(np.int32)

    def step(self, actions):
        """One tick of every lane.  A lane whose episode ended on the previous call was teleported to the start then; on
        this call its action is replaced by the reference reset's zero action (projectd_env.py:216-227: teleport, one
        step([0,0]), clear the episode sums), its reward is 0 and the observation returned is the new episode's first."""
        cfg = self.cfg
        a = np.array(actions, dtype=np.float32).reshape(self.num_envs, 2)
        if self.kernel_env:   # everything below happened inside the kernel (flags bit 3 = terminated, bit 4 = this was the reset tick)
            obs, reward, flags = self._raw_step(a)
            terminated = (flags & 8) != 0
            fresh = (flags & 16) != 0
            self.last_reset_tick = fresh
            self.total_reward += reward; self.total_reward[fresh] = 0.0
            self.step_id += 1; self.step_id[fresh] = 0
            info = {'episode_reward': self.total_reward.copy()} if terminated.any() else {}
            if self.same_step_reset and terminated.any():
                obs = self._reset_tick_now(terminated, obs, info)
            return obs, reward.astype(np.float32), terminated, np.zeros(self.num_envs, dtype=bool), info
        fresh = self.pending_reset.copy()
        a[fresh] = 0.0
        obs, reward, flags = self._raw_step(a)
        terminated = np.zeros(self.num_envs, dtype=bool)
        if cfg.terminate_on_hit:
            hit = (flags & FLAG_COLLISION) != 0
            reward = reward - cfg.terminate_hit_penalty * hit; terminated |= hit
        if cfg.terminate_off_track:
            off = (flags & FLAG_OFFTRACK) != 0
            reward = reward - cfg.terminate_off_track_penalty * off; terminated |= off
        if cfg.terminate_when_stuck:
            stuck = (flags & FLAG_STUCK) != 0
            reward = reward - cfg.terminate_stuck_penalty * stuck; terminated |= stuck
        self.total_reward += reward
        terminated |= self.total_reward < cfg.terminate_low_reward
        self.step_id += 1
        # the reset tick itself: the reference discards its reward / termination and zeroes the sums afterwards
        reward[fresh] = 0.0; terminated[fresh] = False
        self.total_reward[fresh] = 0.0; self.step_id[fresh] = 0
        self.pending_reset[:] = False
        truncated = np.zeros(self.num_envs, dtype=bool)
        info = {}
        if self.auto_reset and terminated.any():
            info['episode_reward'] = self.total_reward.copy()
            if cfg.teleport_on_reset:
                pd.resetBatch(self.batch, terminated.astype(np.uint8), int(cfg.teleport_mode))
            self.pending_reset |= terminated
        return obs, reward.astype(np.float32), terminated, truncated, info

    def _reset_tick_now(self, done, obs, info):
        """same-step reset (kernel env mode): the done lanes' reset tick right away, everybody else held; their rows of `obs` become the new episodes'
        first observations, the terminal ones go to info['terminal_observation']"""
        info['terminal_observation'] = {int(i): obs[i].copy() for i in np.nonzero(done)[0]}
        out = pd.stepBatchHeld(self.batch, np.zeros((self.num_envs, 2), np.float32), (~done).astype(np.uint8), SIM_DT)
        if out.shape != (self.num_envs, 26):
            raise RuntimeError('stepBatchHeld failed')
        fresh = (out[:, 25].view(np.int32) & 16) != 0
        if not np.array_equal(fresh & done, done):
            raise RuntimeError('a lane that ended its episode did not take its reset tick')
        obs = obs.copy(); obs[done] = out[done, :OBS_DIM]
        self.total_reward[done] = 0.0; self.step_id[done] = 0
        self.last_reset_tick = np.zeros(self.num_envs, dtype=bool)   # nobody's NEXT step is a reset tick
        return obs

    def reset(self, mask=None):
        """Reset every lane (mask None) -- teleport + one zero-action tick, returns the first observations -- or schedule
        the masked lanes: they are teleported now and take their zero-action reset tick inside the next step()."""
        if mask is None:
            was = self.kernel_env
            if was:
                self._set_kernel_env(False)
            if self.cfg.teleport_on_reset:
                pd.resetBatch(self.batch, None, int(self.cfg.teleport_mode))
            obs, _, _ = self._raw_step(np.zeros((self.num_envs, 2), np.float32))
            self.total_reward[:] = 0.0; self.step_id[:] = 0; self.pending_reset[:] = False
            if was:
                pd.clearBatchEpisodes(self.batch)
                self._set_kernel_env(True)
            return obs
        m = np.ascontiguousarray(mask).astype(bool)
        if self.kernel_env:   # teleport now; the kernel treats the lanes' next tick as their reset tick
            if self.cfg.teleport_on_reset:
                pd.resetBatch(self.batch, m.astype(np.uint8), int(self.cfg.teleport_mode))
            pd.markBatchResetTick(self.batch, m.astype(np.uint8))
            return None
        if self.cfg.teleport_on_reset:
            pd.resetBatch(self.batch, m.astype(np.uint8), int(self.cfg.teleport_mode))
        self.pending_reset |= m
        return None


def obs_from_state(s):
    """CarState -> the 24-slot observation (projectd_env.py:239-273)"""
    return np.array([s.localVelocity.x, s.localVelocity.y, s.localVelocity.z,
                     s.localAngularVelocity.x, s.localAngularVelocity.y, s.localAngularVelocity.z,
                     s.tyreNdSlip[0], s.tyreNdSlip[1], s.tyreNdSlip[2], s.tyreNdSlip[3],
                     s.bodyVsTrack, s.velocityVsTrack,
                     s.lookAhead[0], s.lookAhead[1], s.lookAhead[2], s.lookAhead[3], s.lookAhead[4],
                     s.probes[0], s.probes[1], s.probes[2], s.probes[3], s.probes[4], s.probes[5], s.probes[6]], dtype=np.float32)


def linscale(x, x0, x1, r0, r1):
    x = min(max(x, x0), x1)
    return ((r1 - r0) * (x - x0)) / (x1 - x0) + r0


def default_base_dir():
    """the reference env finds its content next to the module (projectd_env.py:9); here: $PROJECTD_BASE"""
    d = os.environ.get('PROJECTD_BASE')
    if not d:
        raise RuntimeError('no content directory: pass base_dir=... or set PROJECTD_BASE (the directory holding cfg/ and content/)')
    return d


class ProjectDEnv:
    """single env through the classic calls (setCarControls / stepSimulator / getCarState), reference signature"""

    def __init__(self, base_dir=None, **settings):
        base_dir = base_dir or default_base_dir()
        self.cfg = EnvConfig(**settings)
        self.sim, self.car = _configure_simulator(self.cfg, base_dir)
        self.dstate = pd.CarState()
        self.dcontrols = pd.CarControls()
        self.step_id = 0
        self.total_reward = 0.0

    def close(self):
        if self.sim >= 0:
            pd.destroySimulator(self.sim); self.sim = -1

    def _get_obs_space(self):
        return obs_bounds(self.cfg)

    def _get_action_space(self):
        return np.array([-1.0, -1.0], np.float32), np.array([1.0, 1.0], np.float32)

    def step(self, action):
        cfg = self.cfg
        self.dcontrols.steer = float(action[0])
        self.dcontrols.gas = linscale(float(action[1]), -1.0, 1.0, cfg.min_gas, cfg.max_gas)
        if not cfg.auto_clutch:
            self.dcontrols.clutch = 1.0
        if not cfg.auto_shift:
            self.dcontrols.requestedGearIndex = 2
        pd.setCarControls(self.sim, self.car, cfg.smooth_controls, self.dcontrols)
        pd.stepSimulator(self.sim, SIM_DT)
        pd.getCarState(self.sim, self.car, self.dstate)
        self.step_id += 1
        s = self.dstate
        reward = s.stepReward
        terminate = False
        if cfg.terminate_on_hit and s.collisionFlag != 0:
            reward -= cfg.terminate_hit_penalty; terminate = True
        if cfg.terminate_off_track and s.outOfTrackFlag != 0:
            reward -= cfg.terminate_off_track_penalty; terminate = True
        if cfg.terminate_when_stuck and s.lastTrackPointTimestamp + cfg.stuck_timeout < s.timestamp:
            reward -= cfg.terminate_stuck_penalty; terminate = True
        self.total_reward += reward
        if self.total_reward < cfg.terminate_low_reward:
            terminate = True
        return obs_from_state(s), reward, terminate, False, {}

    def reset(self):
        if self.cfg.teleport_on_reset:
            pd.teleportCarByMode(self.sim, self.car, self.cfg.teleport_mode)
        state, _, _, _, _ = self.step(np.array([0, 0, 0]))
        self.step_id = 0
        self.total_reward = 0
        return state

    def render(self):
        pass

"""Device-resident vector env for learners that live on the GPU: actions, observations, rewards and episode flags are torch
tensors on the batch's device; the step kernel reads the action tensor and writes the observation block in place (zero copy
through pdb_actions_device / pdb_out_device), and the reference env's reward / termination bookkeeping
(pyprojectd/projectd_env.py:173-227) runs as a handful of elementwise torch ops on the same stream.  Episode ends never reach
the host: the `terminated` mask is written into the batch's reset mask on the device and the next tick teleports those cars
(Car::teleportByMode + Car::reset) before it steps them -- step() contains no host synchronisation.  PyTorch here is plumbing:
device tensors and streams, no physics."""
import ctypes as C
import numpy as np
import torch

import pdbatch
import pdb_ctypes as pc
import projectd_env as E


class _DevView:
    def __init__(self, ptr, shape, typestr='<f4'):
        self.__cuda_array_interface__ = {'shape': shape, 'typestr': typestr, 'data': (ptr, False), 'version': 2}


class ProjectDTorchVecEnv:
    def __init__(self, num_envs, params, track_blob, device=0, **settings):
        self.cfg = cfg = E.EnvConfig(**settings)
        self.num_envs = n = int(num_envs)
        self.dev = torch.device('cuda:%d' % device)
        # setCarAutoTeleport (projectd_env.py:124): the teleport inside the tick that raises the flag
        lib = pc.load_product()
        if lib.pdb_set_auto_teleport(C.byref(params), int(cfg.teleport_on_hit), int(cfg.teleport_off_track), int(cfg.teleport_mode)) != 0:
            raise ValueError(lib.pdb_last_error().decode())
        self.batch = pdbatch.Batch(n, params, track_blob, device=device, action_mode=1)
        self.batch.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        if cfg.stuck_timeout != 5.0:
            self.batch.set_stuck_timeout(cfg.stuck_timeout)
        self.reset_mask = torch.as_tensor(_DevView(self.batch.reset_mask_ptr(), (n,), '|u1'), device=self.dev)
        self.act = torch.as_tensor(_DevView(self.batch.actions_device_ptr(), (n, 2)), device=self.dev)
        self.out = torch.as_tensor(_DevView(self.batch.out_device_ptr(), (n, 26)), device=self.dev)
        self.flags = torch.as_tensor(_DevView(self.batch.out_device_ptr(), (n, 26), '<i4'), device=self.dev)[:, 25]
        self.total_reward = torch.zeros(n, dtype=torch.float64, device=self.dev)
        self.pending_reset = torch.zeros(n, dtype=torch.bool, device=self.dev)
        self.step_id = torch.zeros(n, dtype=torch.int64, device=self.dev)

    def close(self):
        self.batch.close()

    def reset(self):
        if self.cfg.teleport_on_reset:
            self.batch.reset(None, self.cfg.teleport_mode)
        self.act.zero_()
        self.batch.step_async()
        self.total_reward.zero_(); self.step_id.zero_(); self.pending_reset.zero_()
        return self.out[:, :E.OBS_DIM]

    def step(self, actions):
        """actions: float32 tensor [N, 2] on the device.  Returns views / tensors on the device: obs [N, 24] (a view of the
        batch's output block: consume it before the next step), reward [N], terminated [N], truncated [N]."""
        cfg = self.cfg
        fresh = self.pending_reset
        self.act.copy_(torch.where(fresh[:, None], torch.zeros_like(actions), actions))
        self.batch.step_async()
        obs = self.out[:, :E.OBS_DIM]
        reward = self.out[:, 24].to(torch.float64)
        fl = self.flags
        terminated = torch.zeros(self.num_envs, dtype=torch.bool, device=self.dev)
        if cfg.terminate_on_hit:
            hit = (fl & E.FLAG_COLLISION) != 0
            reward = reward - cfg.terminate_hit_penalty * hit; terminated = terminated | hit
        if cfg.terminate_off_track:
            off = (fl & E.FLAG_OFFTRACK) != 0
            reward = reward - cfg.terminate_off_track_penalty * off; terminated = terminated | off
        if cfg.terminate_when_stuck:
            stuck = (fl & E.FLAG_STUCK) != 0
            reward = reward - cfg.terminate_stuck_penalty * stuck; terminated = terminated | stuck
        self.total_reward += reward
        terminated = terminated | (self.total_reward < cfg.terminate_low_reward)
        self.step_id += 1
        reward = torch.where(fresh, torch.zeros_like(reward), reward)
        terminated = terminated & ~fresh
        self.total_reward = torch.where(fresh, torch.zeros_like(self.total_reward), self.total_reward)
        self.step_id = torch.where(fresh, torch.zeros_like(self.step_id), self.step_id)
        self.pending_reset = terminated.clone()
        if cfg.teleport_on_reset:   # the next tick teleports these lanes first (value = 1 + Car::teleportByMode's mode), then steps them
            self.reset_mask.copy_(terminated.to(torch.uint8) * (1 + int(cfg.teleport_mode)))
        return obs, reward.to(torch.float32), terminated, torch.zeros_like(terminated)

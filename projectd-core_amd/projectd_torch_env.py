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


FLAG_TERMINATED, FLAG_RESET_TICK = 8, 16   # pdb_step_out.flags in env mode (include/pdb_types.h)


class ProjectDTorchVecEnv:
    def __init__(self, num_envs, params, track_blob, device=0, **settings):
        self.cfg = cfg = E.EnvConfig(**settings)
        self.num_envs = n = int(num_envs)
        self.dev = torch.device('cuda:%d' % device)
        # setCarAutoTeleport (projectd_env.py:124): the teleport inside the tick that raises the flag
        lib = pc.load_product()
        params = pc.CarParams.from_buffer_copy(bytes(params))   # the caller's block stays as it was
        if lib.pdb_set_auto_teleport(C.byref(params), int(cfg.teleport_on_hit), int(cfg.teleport_off_track), int(cfg.teleport_mode)) != 0:
            raise ValueError(lib.pdb_last_error().decode())
        self.batch = pdbatch.Batch(n, params, track_blob, device=device, action_mode=1)
        self.batch.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        self.batch.set_env(cfg)   # rewards with penalties, terminations, the reset tick: all inside the step kernel
        self.act = torch.as_tensor(_DevView(self.batch.actions_device_ptr(), (n, 2)), device=self.dev)
        self.out = torch.as_tensor(_DevView(self.batch.out_device_ptr(), (n, 26)), device=self.dev)
        self.flags = torch.as_tensor(_DevView(self.batch.out_device_ptr(), (n, 26), '<i4'), device=self.dev)[:, 25]

    def close(self):
        self.batch.close()

    def reset(self):
        """every lane: teleport + the zero-action tick (projectd_env.py:216-227); returns the first observations"""
        self.batch.set_env(self.cfg, enabled=False)
        if self.cfg.teleport_on_reset:
            self.batch.reset(None, self.cfg.teleport_mode)
        self.act.zero_()
        self.batch.step_async()
        self.batch.clear_episodes()      # on the device: no record crosses PCIe
        self.batch.set_env(self.cfg)
        return self.out[:, :E.OBS_DIM]

    def step(self, actions):
        """actions: float32 tensor [N, 2] on the device.  One kernel launch (plus the contact pass's): the env's reward with its
        penalties, the terminations, and -- on the tick after a termination -- the teleport and the zero action of the env's
        reset() all happen inside it.  Returns views on the batch's output block (consume them before the next step): obs [N, 24],
        reward [N], terminated [N] (bool), truncated [N].  No host synchronisation."""
        self.act.copy_(actions)
        self.batch.step_async()
        terminated = (self.flags & FLAG_TERMINATED) != 0
        return self.out[:, :E.OBS_DIM], self.out[:, 24], terminated, torch.zeros_like(terminated)

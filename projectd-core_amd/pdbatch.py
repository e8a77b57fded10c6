"""Python surface of the pdbatch C ABI (include/pdbatch.h): ctypes bindings + a batched environment mirroring
pyprojectd/projectd_env.py (ProjectDEnv.step/reset) for N cars at once.  PyTorch is used only for device
buffers / streams by callers that want zero-copy actions and observations; nothing here computes physics."""
import ctypes as C, os, sys, tempfile
import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
sys.path.insert(0, PKG)
import pdb_ctypes as pc  # ctypes views of include/pdb_types.h  # noqa: E402

SIM_DT = 1.0 / 333.0   # projectd_env.py:19


class Batch:
    """N (simulator, car) pairs resident on one GPU."""

    def __init__(self, n_cars, params, track_blob, device=0, action_mode=1):
        self.lib = pc.load_product()
        self.n = n_cars
        self.action_mode = action_mode
        self.stride = 8 if action_mode == 2 else 2   # floats per car (include/pdbatch.h pdb_action_mode)
        self.params = params
        self.track = track_blob
        self.h = self.lib.pdb_create(device, n_cars, C.byref(params), track_blob, len(track_blob), action_mode)
        if not self.h:
            raise RuntimeError('pdb_create failed: %s' % self.lib.pdb_last_error().decode())

    def close(self):
        if self.h:
            self.lib.pdb_destroy(self.h)
            self.h = None

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.pdb_last_error().decode())

    def step_host(self, actions, want_out=True):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n, self.stride)
        out = (pc.StepOut * self.n)()
        self._chk(self.lib.pdb_step_host(self.h, a.ctypes.data_as(C.c_void_p), C.c_float(SIM_DT), C.byref(out)))
        o = np.frombuffer(out, dtype=np.dtype(pc.StepOut))
        return o

    def set_actions(self, actions):
        import torch  # device plumbing only
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n, self.stride)
        self._chk(self.lib.pdb_step_host(self.h, a.ctypes.data_as(C.c_void_p), C.c_float(SIM_DT), None))

    def step_async(self):
        self._chk(self.lib.pdb_step_async(self.h, C.c_float(SIM_DT)))

    def sync(self):
        self._chk(self.lib.pdb_sync(self.h))

    def set_partitions(self, parts):
        """cut the batch into `parts` free-running car ranges, one HIP stream each (pdb_step_ring)"""
        self._chk(self.lib.pdb_set_partitions(self.h, parts))

    def set_partition_params(self, part, params):
        """a car block of its own for one partition (None: the batch's): same model, different tunes / weights"""
        self._chk(self.lib.pdb_set_partition_params(self.h, part, C.byref(params) if params is not None else None))

    def step_ring(self, n_ticks, ring_ptr=None, ring_slots=1, first_slot=0, join=True, fork=True):
        """enqueue n_ticks ticks of every car, partition by partition; tick i writes outputs to ring slot (first_slot + i) % ring_slots.
        join=False: the batch's stream is not held back; order consumers with wait_partitions().  fork=False: the partitions do not wait for what is queued
        on the batch's stream (the caller orders its dependencies on the partitions' streams itself)"""
        self._chk(self.lib.pdb_step_ring(self.h, C.c_float(SIM_DT), n_ticks, C.c_void_p(ring_ptr) if ring_ptr else None, ring_slots, first_slot, (1 if join else 0) | (0 if fork else 2)))

    def wait_partitions(self, stream_ptr=None):
        """make a stream (default: the batch's) wait for every partition's last enqueued kernel"""
        self._chk(self.lib.pdb_wait_partitions(self.h, C.c_void_p(stream_ptr) if stream_ptr else None))

    def step_partition(self, part, out_ptr=None):
        """one tick of one partition on its own stream (nothing forked or joined)"""
        self._chk(self.lib.pdb_step_partition(self.h, C.c_float(SIM_DT), part, C.c_void_p(out_ptr) if out_ptr else None))

    def set_world_size(self, cars_per_world):
        """multi-car simulators: worlds of `cars_per_world` consecutive lanes (coupled through the slipstream: Car::updateAirPressure)"""
        self._chk(self.lib.pdb_set_world_size(self.h, int(cars_per_world)))

    def get_slipstreams(self):
        """-> (SlipState * n, SlipState * n): the two buffers of the cars' wakes (a car's next tick reads the one of its simFrame's parity)"""
        buf = (pc.SlipState * (2 * self.n))()
        self._chk(self.lib.pdb_get_slipstreams(self.h, 0, self.n, buf))
        return [(pc.SlipState * self.n).from_buffer(buf, 0), (pc.SlipState * self.n).from_buffer(buf, C.sizeof(pc.SlipState) * self.n)]

    def set_slipstreams(self, both):
        buf = (pc.SlipState * (2 * self.n))()
        C.memmove(buf, both[0], C.sizeof(pc.SlipState) * self.n); C.memmove(C.byref(buf, C.sizeof(pc.SlipState) * self.n), both[1], C.sizeof(pc.SlipState) * self.n)
        self._chk(self.lib.pdb_set_slipstreams(self.h, 0, self.n, buf))

    def host_mirrors(self):
        """numpy views of the library's page-locked host mirrors: actions float32[n, stride], outputs (structured pdb_step_out)[n]"""
        pa = self.lib.pdb_host_actions(self.h); po = self.lib.pdb_host_out(self.h)
        if not pa or not po:
            raise RuntimeError(self.lib.pdb_last_error().decode())
        a = np.ctypeslib.as_array((C.c_float * (self.n * self.stride)).from_address(pa)).reshape(self.n, self.stride)
        o = np.frombuffer((pc.StepOut * self.n).from_address(po), dtype=np.dtype(pc.StepOut))
        return a, o

    def step_host_partition(self, part):
        """enqueue on the partition's stream: its action rows up (from the mirror), one tick, its output rows down (into the mirror)"""
        self._chk(self.lib.pdb_step_host_partition(self.h, C.c_float(SIM_DT), part))

    def wait_host_partition(self, part):
        self._chk(self.lib.pdb_wait_host_partition(self.h, part))

    def comm_unique_ids(self, parts):
        """rank 0: one 128-byte RCCL id per partition (bytes, to be handed to every rank)"""
        buf = (C.c_uint8 * (128 * parts))()
        for p in range(parts):
            self._chk(self.lib.pdb_comm_unique_id(C.byref(buf, 128 * p)))
        return bytes(buf)

    def comm_init(self, world, rank, ids):
        """collective over the ranks, after set_partitions: the partitions' RCCL communicators inside the library"""
        buf = (C.c_uint8 * len(ids)).from_buffer_copy(ids)
        self._chk(self.lib.pdb_comm_init(self.h, world, rank, buf, len(ids) // 128))

    def step_exchange_partition(self, part, scatter_src_ptr, gathered_ptr):
        """on the partition's stream: the learner's action rows in, one tick, the output rows of every rank out (device pointers)"""
        self._chk(self.lib.pdb_step_exchange_partition(self.h, C.c_float(SIM_DT), part, C.c_void_p(scatter_src_ptr), C.c_void_p(gathered_ptr)))

    def partition_stream(self, part):
        return self.lib.pdb_partition_stream(self.h, part)

    def partition_range(self, part):
        f = C.c_int(); c = C.c_int()
        self._chk(self.lib.pdb_partition_range(self.h, part, C.byref(f), C.byref(c)))
        return f.value, c.value

    def partition_mark(self):
        self._chk(self.lib.pdb_partition_mark(self.h))

    def partition_elapsed_ms(self, part):
        ms = C.c_float(); cars = C.c_int()
        self._chk(self.lib.pdb_partition_elapsed_ms(self.h, part, C.byref(ms), C.byref(cars)))
        return ms.value, cars.value

    def set_stream(self, stream_ptr):
        self._chk(self.lib.pdb_set_stream(self.h, C.c_void_p(stream_ptr)))

    def event_record(self, which):
        self._chk(self.lib.pdb_event_record(self.h, which))

    def event_elapsed_ms(self):
        ms = C.c_float()
        self._chk(self.lib.pdb_event_elapsed_ms(self.h, C.byref(ms)))
        return ms.value

    def upload_actions(self, actions):
        """host -> device copy of the action array without stepping (hipMemcpy through a torch-free ctypes call)"""
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n, self.stride)
        hip = C.CDLL('libamdhip64.so')
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        rc = hip.hipMemcpy(self.lib.pdb_actions_device(self.h), a.ctypes.data_as(C.c_void_p), a.nbytes, 1)
        if rc != 0:
            raise RuntimeError('hipMemcpy failed: %d' % rc)

    def set_out_device_ptr(self, ptr):
        self._chk(self.lib.pdb_set_out_device(self.h, C.c_void_p(ptr)))

    def out_device_ptr(self):
        return self.lib.pdb_out_device(self.h)

    def actions_device_ptr(self):
        return self.lib.pdb_actions_device(self.h)

    def step(self, ticks=1):
        self._chk(self.lib.pdb_step_n(self.h, C.c_float(SIM_DT), ticks))

    def get_state(self, first=0, count=None):
        count = self.n - first if count is None else count
        st = (pc.DynState * count)()
        self._chk(self.lib.pdb_get_state(self.h, first, count, C.byref(st)))
        return st

    def set_state(self, states, first=0):
        self._chk(self.lib.pdb_set_state(self.h, first, len(states), C.byref(states)))

    def get_contacts(self, first=0, count=None):
        """the cars' live contact joints: [count][MAX_CONTACTS] pdb_contact, the first DynState.numContacts of a row alive"""
        count = self.n - first if count is None else count
        ct = ((pc.Contact * pc.MAX_CONTACTS) * count)()
        self._chk(self.lib.pdb_get_contacts(self.h, first, count, C.byref(ct)))
        return ct

    def set_contacts(self, contacts, first=0):
        self._chk(self.lib.pdb_set_contacts(self.h, first, len(contacts), C.byref(contacts)))

    def get_car_state(self, first=0, count=None):
        count = self.n - first if count is None else count
        cs = (pc.CarState * count)()
        self._chk(self.lib.pdb_get_car_state(self.h, first, count, C.byref(cs)))
        return cs

    def reset(self, mask=None, mode=0):
        """teleportCarByMode(mode) for the masked cars (host mask; the teleport runs on the device): 0 Start, 1 Nearest, 2 Random"""
        if mask is None:
            self._chk(self.lib.pdb_reset_mode(self.h, None, mode))
        else:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            self._chk(self.lib.pdb_reset_mode(self.h, m.ctypes.data_as(C.c_void_p), mode))

    def reset_device(self, mask_ptr, mode=0):
        """the same with a uint8 mask that already lives on the device: asynchronous on the batch's stream, no host round trip"""
        self._chk(self.lib.pdb_reset_device(self.h, C.c_void_p(mask_ptr), mode))

    def clear_episodes(self, mask_ptr=None):
        """env mode: zero the episode sums of the masked cars (device uint8 mask; None: all) on the device, asynchronously"""
        self._chk(self.lib.pdb_clear_episodes(self.h, C.c_void_p(mask_ptr) if mask_ptr else None))

    def reset_mask_ptr(self):
        """device array of n bytes: a car whose byte is 1 + mode is teleported at the top of its next tick, which clears the byte"""
        return self.lib.pdb_reset_mask_device(self.h)

    def set_contact_grid(self, workgroups):
        """0 = the contact pass's grid follows the load (default); > 0 = fixed (for launches recorded into a caller's graph)"""
        self._chk(self.lib.pdb_set_contact_grid(self.h, int(workgroups)))

    def set_law(self, weights, bias0=None, table=None, table_device_ptr=None, period=0):
        """the device law (pdb_set_law): the next tick's two action components evaluated by the tick's own launches, a = bias + obs[:24] @ weights (float32, the products
        summed as a balanced binary tree over 32 slots); bias = bias0 [2], or row pdb_dyn_state.lawTick of `table` [period, n, 2] (a numpy array: copied to the
        device) / of the device memory at table_device_ptr (period rows; the caller keeps it alive).  weights=None removes the law"""
        import numpy as np
        if weights is None:
            self._chk(self.lib.pdb_set_law(self.h, None, None, None, 0, 0)); return
        w = np.ascontiguousarray(weights, dtype=np.float32)
        assert w.shape == (24, 2)
        b0 = None if bias0 is None else np.ascontiguousarray(bias0, dtype=np.float32)
        assert b0 is None or b0.shape == (2,)
        b0p = None if b0 is None else b0.ctypes.data_as(C.c_void_p)
        if table_device_ptr is not None:
            self._chk(self.lib.pdb_set_law(self.h, w.ctypes.data_as(C.c_void_p), b0p, C.c_void_p(int(table_device_ptr)), int(period), 1))
        elif table is not None:
            t = np.ascontiguousarray(table, dtype=np.float32)
            assert t.ndim == 3 and t.shape[1:] == (self.n, 2)
            self._chk(self.lib.pdb_set_law(self.h, w.ctypes.data_as(C.c_void_p), b0p, t.ctypes.data_as(C.c_void_p), t.shape[0], 0))
        else:
            self._chk(self.lib.pdb_set_law(self.h, w.ctypes.data_as(C.c_void_p), b0p, None, 0, 0))

    def set_lane_tunes(self, blocks, first=0):
        """per-lane setup and reward weights (PyProjectD.cpp:328-365 is per simulator = per env): lane first + i takes the eight env tunes
        (FRONT_BIAS, DIFF_POWER, DIFF_COAST, FINAL_RATIO, PRESSURE_*) and the scoring variables of blocks[i], a pdb_car_params that went
        through pdb_set_car_tune / pdb_set_scoring_var (None: that lane back to the batch's own block)"""
        rows = (pc.LaneTune * len(blocks))()
        for i, P in enumerate(blocks):
            if P is not None:
                self._chk(self.lib.pdb_lane_tune_from_params(C.byref(P), C.byref(rows[i])))
        self._chk(self.lib.pdb_set_lane_tunes(self.h, first, len(blocks), rows))

    def set_lane_setups(self, blocks, first=0):
        """the rest of the setup lane by lane (pdb_set_lane_setups): lane first + i takes brake power, differential preload, gear ratios, anti-roll bars, rev limiter,
        turbo settings and the per-wheel dampers / springs / bump stops / rod lengths / packers / toe / camber of blocks[i], a pdb_car_params that went through
        pdb_set_car_tune; blocks=None with a count puts lanes back to their own block's values"""
        if isinstance(blocks, int):
            self._chk(self.lib.pdb_set_lane_setups(self.h, first, blocks, None)); return
        rows = (pc.LaneSetup * len(blocks))()
        for i, P in enumerate(blocks):
            self._chk(self.lib.pdb_lane_setup_from_params(C.byref(P if P is not None else self.params), C.byref(rows[i])))
        self._chk(self.lib.pdb_set_lane_setups(self.h, first, len(blocks), rows))

    def set_env(self, cfg=None, **kw):
        """env mode (pdb_set_env): the reward / termination / reset rules of projectd_env.py:173-227 inside the tick.  cfg: an object with
        the reference env's attribute names (projectd_env.EnvConfig), or keyword overrides; set_env(enabled=False) switches it off."""
        g = lambda k, d: kw.get(k, getattr(cfg, k, d) if cfg is not None else d)
        c = pc.EnvConfig()
        c.enabled = 1 if kw.get('enabled', True) else 0
        c.terminate_on_hit = int(g('terminate_on_hit', True)); c.terminate_off_track = int(g('terminate_off_track', True)); c.terminate_when_stuck = int(g('terminate_when_stuck', True))
        c.hit_penalty = float(g('terminate_hit_penalty', 50.0)); c.off_track_penalty = float(g('terminate_off_track_penalty', 50.0)); c.stuck_penalty = float(g('terminate_stuck_penalty', 50.0))
        c.low_reward = float(g('terminate_low_reward', -200.0))
        c.teleport_on_reset = int(g('teleport_on_reset', True)); c.teleport_mode = int(g('teleport_mode', 0))
        self._chk(self.lib.pdb_set_env(self.h, C.byref(c)))
        st = g('stuck_timeout', 5.0)
        if st != 5.0:
            self.set_stuck_timeout(st)

    def set_stuck_timeout(self, seconds):
        self._chk(self.lib.pdb_set_stuck_timeout(self.h, C.c_double(seconds)))

    def set_seed(self, seeds):
        s = np.ascontiguousarray(seeds, dtype=np.uint32).reshape(self.n)
        self._chk(self.lib.pdb_set_seed(self.h, s.ctypes.data_as(C.c_void_p)))

    def sample_kernel(self, every):
        """HIP events around every `every`-th first-pass launch of each launch site (0 = off)"""
        self._chk(self.lib.pdb_sample_kernel(self.h, int(every)))

    def sampled_kernel_us(self):
        """(average first-pass duration in us, number of sampled launches, average cars per sampled launch); waits for them, clears the samples"""
        us = C.c_double(); n = C.c_int(); cars = C.c_double()
        self._chk(self.lib.pdb_sampled_kernel_us(self.h, C.byref(us), C.byref(n), C.byref(cars)))
        return us.value, n.value, cars.value

    def kernel_time_us(self):
        us = C.c_double(); n = C.c_int()
        self._chk(self.lib.pdb_kernel_time_us(self.h, C.byref(us), C.byref(n)))
        return us.value, n.value


def packed_params(name='ks_toyota_ae86_drift.env'):
    P = pc.CarParams()
    data = open(os.path.join(PKG, 'data', name + '.pdcar'), 'rb').read()
    assert len(data) == C.sizeof(P)
    C.memmove(C.byref(P), data, len(data))
    return P


def synthetic_track(kind='flat', **gen_args):
    """blob of one of the synthetic tracks; gen_args go to its generator (e.g. step=0.9 for a dense mountain-road spline)"""
    import synthetic_tracks
    lib = pc.load_product()
    d = tempfile.mkdtemp(prefix='pdb_base_')
    synthetic_tracks.make_base(d, tracks=() if gen_args else (kind,))
    if gen_args:
        synthetic_tracks.GENERATORS[kind](os.path.join(d, 'content', 'tracks', kind), **gen_args)
    return pc.build_track(lib, d, kind)


REFERENCE_TRACKS = ('driftplayground', 'ebisu_touge', 'yamanashi_short', 'euphoria_hillside_park', 'ek_akina', 'ks_nordschleife',
                    'ks_nordschleife_walls')


def reference_track(name='driftplayground'):
    """blob of one of the reference's own tracks (projectd_env.py:23 defaults to driftplayground), from its packed form
    (data/tracks/<name>.pdtrack.z, written in the build container by tools/pack_tracks.py: the four meshes through Sim/Track.cpp's
    build, ek_akina / ks_nordschleife as a ribbon around their shipped spline; ks_nordschleife_walls =
    that ribbon with guard rails along both edges)"""
    assert name in REFERENCE_TRACKS, name
    return pc.load_track_pack(name)

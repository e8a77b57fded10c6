"""Episode resets on the device (ScoringSystem.cpp:194-225 for the in-tick form): Car::teleportByMode +
Car::reset run in HIP from the same source as the host library's pdb_teleport_by_mode (csrc/host/reset_core.hpp), which the
reference-TU goldens pin (resets*, teleports, autotele_*).  Here: the device paths against that host function, bit for bit."""
import ctypes as C
import numpy as np
import pytest
import pdb_ctypes as pc

pytestmark = pytest.mark.gpu


def _run(b, acts, ticks):
    for _ in range(ticks):
        b.step_host(acts)


def _setup(n, track='touge', seed=5, ticks=500, **gen):
    import pdbatch, parity_util
    P = pdbatch.packed_params()
    trk = pdbatch.synthetic_track(track, **gen)
    lib = pc.load_product()
    S0 = pc.DynState()
    assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
    b = pdbatch.Batch(n, P, trk, device=0, action_mode=1)
    acts = parity_util.make_actions(n, seed)
    b.set_seed(np.arange(n, dtype=np.uint32) * 7919 + 1)
    _run(b, acts, ticks)
    return lib, P, trk, b, acts


@pytest.mark.parametrize('mode', [0, 1, 2])
def test_device_reset_equals_the_host_teleport(built, mode):
    """pdb_reset_mode (host mask, teleport on the device) == pdb_teleport_by_mode applied to the downloaded records: every byte
    of the record, for Start / Nearest (each car's own trackLocation) / Random (each car's own rand() state)"""
    n = 48
    lib, P, trk, b, acts = _setup(n)
    try:
        before = b.get_state()
        mask = (np.arange(n) % 3 != 1).astype(np.uint8)
        want = [bytes(before[i]) for i in range(n)]
        for i in range(n):
            if mask[i]:
                s = pc.DynState.from_buffer_copy(want[i])
                assert lib.pdb_teleport_by_mode(C.byref(P), trk, mode, C.byref(s)) == 0
                want[i] = bytes(s)
        b.reset(mask, mode)
        after = b.get_state()
        for i in range(n):
            assert bytes(after[i]) == want[i], 'car %d (masked %d)' % (i, mask[i])
        if mode == 2:   # the draws differ from car to car (own generator state each)
            locs = {round(float(after[i].body[0].pos[0]), 3) for i in range(n) if mask[i]}
            assert len(locs) > 10
    finally:
        b.close()


def test_reset_mask_is_consumed_at_the_top_of_the_next_tick(built):
    """deferred form: bytes 1 + mode in pdb_reset_mask_device -> the next tick teleports those cars, steps them and clears the
    bytes; equals an immediate pdb_reset_mode followed by the same tick.  No extra launch, no host round trip."""
    import torch
    n = 33
    lib, P, trk, b, acts = _setup(n, ticks=400)
    lib2, P2, trk2, b2, _ = _setup(n, ticks=400)
    try:
        class V:
            def __init__(self, ptr, n): self.__cuda_array_interface__ = {'shape': (n,), 'typestr': '|u1', 'data': (ptr, False), 'version': 2}
        rm = torch.as_tensor(V(b.reset_mask_ptr(), n), device='cuda:0')
        req = np.zeros(n, dtype=np.uint8)
        req[::4] = 1; req[1::8] = 2; req[2::16] = 3     # Start, Nearest, Random
        rm.copy_(torch.from_numpy(req).cuda()); torch.cuda.synchronize()
        a = acts.copy(); a[req != 0] = 0.0                 # the env's reset tick steps with the zero action
        for mode in (0, 1, 2):
            b2.reset((req == mode + 1).astype(np.uint8), mode)
        b.step_host(a); b2.step_host(a)
        torch.cuda.synchronize()
        assert int(rm.sum().item()) == 0                   # consumed
        s1, s2 = b.get_state(), b2.get_state()
        for i in range(n):
            assert bytes(s1[i]) == bytes(s2[i]), i
        _run(b, acts, 50); _run(b2, acts, 50)
        s1, s2 = b.get_state(), b2.get_state()
        assert all(bytes(s1[i]) == bytes(s2[i]) for i in range(n))
    finally:
        b.close(); b2.close()


@pytest.mark.parametrize('track,auto,ticks', [('walled', 1 | 2 | (1 << 2), 2600), ('touge', 2 | (2 << 2), 2400), ('walled', 1 | (0 << 2), 3000)])
def test_in_tick_auto_teleport_matches_the_oracle(built, track, auto, ticks):
    """setCarAutoTeleport: the kernel teleports inside the tick that raises the flag (ScoringSystem.cpp:194-225); the oracle does it
    through the product's host function; every state scalar, every checked tick, bit for bit"""
    import parity_util
    seen = {'tele': 0}
    last = {}

    def on_tick(t, i, sg, sc):
        p = (sg.body[0].pos[0], sg.body[0].pos[2])
        if i in last and (p[0] - last[i][0]) ** 2 + (p[1] - last[i][1]) ** 2 > 25.0:
            seen['tele'] += 1
        last[i] = p

    def params(P):
        P.autoTeleport = auto
    worst = parity_util.run_parity(n_cars=24, ticks=ticks, seed=11, track=track, check_every=3, on_tick=on_tick, params_fn=params,
                                   actions_fn=(lambda t, a: a * np.array([3.0, 1.0], np.float32)) if track == 'touge' else None)
    assert worst == 0.0, worst
    assert seen['tele'] >= 3, seen


def test_torch_env_runs_episodes_without_the_host(built):
    """ProjectDTorchVecEnv: terminations -> reset mask on the device -> teleport at the top of the next tick; equals the same
    bookkeeping done on the host with pdb_reset_mode between the ticks, lane for lane, over many episode ends (half the cars idle
    and get `stuck` after stuck_timeout = 0.4 s, the others run into the walls of the strip)"""
    import torch, pdbatch, projectd_torch_env, projectd_env
    n, T = 32, 900
    trk = pdbatch.synthetic_track('walled')
    kw = dict(stuck_timeout=0.4, terminate_low_reward=-1.0e9)
    te = projectd_torch_env.ProjectDTorchVecEnv(n, pdbatch.packed_params(), trk, device=0, **kw)
    rng = np.random.RandomState(3)
    base = rng.uniform(-1, 1, (n, 2)).astype(np.float32); base[:, 0] *= 0.2
    base[::2, 1] = -1.0      # gas 0.1: creeps, no new track point within 0.4 s
    base[1::2, 1] = 1.0; base[1::2, 0] = np.where(np.arange(n // 2) % 2 == 0, 0.06, -0.06)   # full throttle into a side wall
    te.reset()
    ends = 0
    hist = []
    act = torch.from_numpy(base).cuda()
    for t in range(T):
        o, r, term, trunc = te.step(act)
        hist.append((o.clone().cpu().numpy(), r.cpu().numpy(), term.cpu().numpy()))
        ends += int(term.sum().item())
    te.close()
    assert ends >= 40, ends
    b = pdbatch.Batch(n, pdbatch.packed_params(), trk, device=0, action_mode=1)
    b.set_stuck_timeout(0.4)
    cfg = projectd_env.EnvConfig(**kw)
    b.reset(None, 0)
    b.step_host(np.zeros((n, 2), np.float32))
    total = np.zeros(n); pending = np.zeros(n, bool)
    for t in range(T):
        a = base.copy(); a[pending] = 0.0
        out = b.step_host(a)
        obs = np.array(out['obs']); rew = np.array(out['reward'], dtype=np.float64); fl = np.array(out['flags'])
        term = np.zeros(n, bool)
        for bit, pen in ((1, cfg.terminate_hit_penalty), (2, cfg.terminate_off_track_penalty), (4, cfg.terminate_stuck_penalty)):
            m = (fl & bit) != 0; rew = rew - pen * m; term |= m
        total += rew; term |= total < cfg.terminate_low_reward
        rew[pending] = 0.0; term[pending] = False; total[pending] = 0.0
        ho, hr, ht = hist[t]
        assert np.array_equal(ht, term), t
        assert np.array_equal(ho, obs), t
        assert np.array_equal(hr, rew.astype(np.float32)), t
        pending = term.copy()
        if term.any():
            b.reset(term.astype(np.uint8), 0)
    b.close()


def test_env_mode_recreates_a_car_whose_state_is_no_longer_finite(built):
    """not in the reference env: a NaN in a car's record (a solver blow-up; here injected) raises flags bit 5, ends the episode, and the next tick re-creates the
    car from a fresh record and teleports it -- the lane is alive again two ticks later, every other lane is byte-identical to a run without the injection, and the
    launch does not crawl (on an open track the locator's fallback trace walked 250 000 steps for such a car: 9 ms per tick)"""
    import time, pdbatch, projectd_env, parity_util
    n = 96
    P = pdbatch.packed_params(); trk = pdbatch.reference_track('ek_akina')
    acts = parity_util.make_actions(n, 5); acts[:, 1] = np.abs(acts[:, 1])
    runs = []
    for inject in (False, True):
        b = pdbatch.Batch(n, P, trk, device=0, action_mode=1)
        b.set_seed(np.arange(n, dtype=np.uint32) * 7919 + 1)
        b.reset(mode=2)
        b.set_env(projectd_env.EnvConfig(teleport_mode=2))
        for _ in range(120):
            b.step_host(acts)
        if inject:
            st = b.get_state()
            st[17].body[0].avel[1] = float('nan'); st[17].body[2].lvel[0] = float('inf')
            b.set_state(st)
        flags, dts = [], []
        for _ in range(40):
            t0 = time.perf_counter(); o = b.step_host(acts); dts.append(time.perf_counter() - t0)
            flags.append(np.array(o['flags']))
        runs.append((b.get_state(), np.array(flags), np.median(dts)))
        b.close()
    (s0, f0, d0), (s1, f1, d1) = runs
    assert (f1[0, 17] & 32) != 0 and (f1[0, 17] & 8) != 0            # faulted and terminated on the tick it shows
    assert (f1[1, 17] & 16) != 0 and (f1[2:, 17] & 32).max() == 0    # reset tick, then finite for good
    raw0 = np.frombuffer(bytes(s0), dtype=np.uint8).reshape(n, -1); raw1 = np.frombuffer(bytes(s1), dtype=np.uint8).reshape(n, -1)
    others = np.arange(n) != 17
    assert (raw0[others] == raw1[others]).all()
    assert np.isfinite(np.frombuffer(bytes(s1[17]), dtype=np.float32)[24:200]).all()
    assert d1 < 3.0 * d0 + 1e-3, (d0, d1)

"""The 49 scripted scenarios of oracle/scenarios.h (the reference-TU goldens' scripts) as a set-up + driver usable from any
test: car block, track blob, initial state the way tests/test_oracle_golden.py prepares them, and a tick-by-tick driver that
steps the CPU oracle and -- optionally -- a GPU batch through the same script (actions, mid-run resets / teleports, the boost)."""
import ctypes as C, os, sys
import numpy as np
import pdb_ctypes as pc
from conftest import car_params

HERE = os.path.dirname(os.path.abspath(__file__))
REF_CONTENT = os.environ.get('PDB_REF_CONTENT', '/root/reference/content')   # (the override lets a test hide it, as on the GPU box)


class Skip(Exception):
    pass


RIBBON_TRACKS = ('ek_akina', 'ks_nordschleife')   # shipped with their spline only: the road is generated around it


def track_blob(hostlib, track, base_dir):
    """the product's track blob for a scenario's track; Skip when it needs reference content that is absent"""
    import synthetic_tracks
    if track in ('flat', 'touge', 'walled'):
        gen = {'flat': synthetic_tracks.gen_flat, 'touge': synthetic_tracks.gen_touge, 'walled': synthetic_tracks.gen_walled}[track]
        gen(os.path.join(base_dir, 'content', 'tracks', track))
        return pc.build_track(hostlib, base_dir, track)
    if not os.path.isdir(os.path.join(REF_CONTENT, 'tracks', track)):
        try:    # a machine without the reference's content: the blob packed in the build container (tools/pack_tracks.py)
            return pc.load_track_pack(track)
        except RuntimeError as e:
            raise Skip(str(e))
    if track in RIBBON_TRACKS:
        synthetic_tracks.ribbon_track_from(os.path.join(REF_CONTENT, 'tracks', track), os.path.join(base_dir, 'content', 'tracks', track))
        return pc.build_track(hostlib, base_dir, track)
    return pc.build_track(hostlib, '/root/reference', track)


TELEPORT_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float)


def teleport_state(hostlib, P, blob, state_ref, kind, a, b, c):
    """one mid-run teleport of a scenario's script on a state record, through the PRODUCT's host functions: kind 0 teleportCarToSpline(a),
    1 teleportCarToPits(int(a)), 2 teleportCarToLocation(a, b, c)"""
    if kind == 0:
        assert hostlib.pdb_teleport_to_spline(C.byref(P), blob, C.c_float(a), state_ref) == 0
    elif kind == 1:
        assert hostlib.pdb_teleport_to_pit(C.byref(P), blob, int(a), state_ref) == 0
    else:
        assert hostlib.pdb_teleport_to_location(C.byref(P), blob, C.c_float(a), C.c_float(b), C.c_float(c), state_ref) == 0


def teleport_callback(hostlib, P, blob):
    """the callback cpuref_run_scenario_cb wants (keep a reference to it while the run lasts)"""
    return TELEPORT_CB(lambda sp, kind, a, b, c: teleport_state(hostlib, P, blob, C.c_void_p(sp), kind, a, b, c))


def setup(orc, hostlib, sid, base_dir):
    """-> dict(name, track, model, P, blob, S0, fields).  Raises Skip when the scenario needs reference content that is absent."""
    import synthetic_tracks
    name = orc.cpuref_scenario_name(sid).decode()
    track = orc.cpuref_scenario_track(sid).decode()
    model = orc.cpuref_scenario_car(sid).decode()
    blob = track_blob(hostlib, track, base_dir)
    P = car_params(model)
    nm = C.c_char_p(); val = C.c_float()
    if orc.cpuref_scenario_tune(sid, 0, C.byref(nm), C.byref(val)):
        if not os.path.isdir(os.path.join(REF_CONTENT, 'cars')):
            # the cars' setup.ini is absent: the block as tools/pack_cars.py left it after the env's tunes and this scenario's list
            # (tests/test_loader.py holds the packed block against this very loop in the build container)
            P = car_params(name, kind='tuned')
        else:
            P = pc.env_params(hostlib, '/root/reference', model)
            i = 0
            while orc.cpuref_scenario_tune(sid, i, C.byref(nm), C.byref(val)):
                hostlib.pdb_set_car_tune(C.byref(P), b'/root/reference', model.encode(), nm.value, val.value, 0)
                i += 1
    i = 0
    while orc.cpuref_scenario_scoring(sid, i, C.byref(nm), C.byref(val)):
        assert hostlib.pdb_set_scoring_var(C.byref(P), nm.value, val.value) == 0
        i += 1
    P.collider.enabled = orc.cpuref_scenario_collide(sid)
    at = orc.cpuref_scenario_auto_teleport(sid)
    assert hostlib.pdb_set_auto_teleport(C.byref(P), at & 1, (at >> 1) & 1, (at >> 2) & 3) == 0
    ticks = C.c_int(); full = C.c_int(); assists = (C.c_int * 4)()
    assert orc.cpuref_scenario_info(sid, C.byref(ticks), C.byref(full), assists) == 0
    hostlib.pdb_set_assists(C.byref(P), assists[0], assists[1], assists[2], assists[3])
    f = (C.c_int * 8)(); orc.cpuref_scenario_fields(sid, f)
    S0 = pc.DynState()
    assert hostlib.pdb_initial_state(C.byref(P), blob, C.byref(S0)) == 0
    d2 = C.c_float()
    two = orc.cpuref_scenario_two_car(sid, C.byref(d2))
    S1 = None
    if two:   # a second car of the same model in the same simulator, put down ahead of the first (teleportCarToSpline)
        S1 = pc.DynState()
        assert hostlib.pdb_initial_state(C.byref(P), blob, C.byref(S1)) == 0
        assert hostlib.pdb_teleport_to_spline(C.byref(P), blob, d2, C.byref(S1)) == 0
    return dict(sid=sid, name=name, track=track, model=model, P=P, blob=blob, S0=S0, ticks=ticks.value, full=full.value,
                reset_every=f[0], tele_dist=f[1], boost_at=f[2], feedback=f[3], auto_tele=f[5], two_car=two, S1=S1)


def run_two_car_script(orc, sc, out0, out1):
    """a twoCar scenario through the oracle exactly like the reference-TU harness ran it: two probe files"""
    P, blob = sc['P'], sc['blob']
    h0 = orc.cpuref_create(C.byref(P), blob, len(blob), C.byref(sc['S0'])); h1 = orc.cpuref_create(C.byref(P), blob, len(blob), C.byref(sc['S1']))
    try:
        assert orc.cpuref_run_scenario2(h0, h1, sc['sid'], out0.encode(), out1.encode()) == 0
    finally:
        orc.cpuref_destroy(h0); orc.cpuref_destroy(h1)


def drive_two_cars(orc, sc, batch=None, max_ticks=None, on_tick=None):
    """Step two oracle cars of one simulator -- and a GPU batch of two lanes forming one world (pdb_set_world_size(2)), if given -- through a twoCar scenario:
    each car's tick reads the slipstream the OTHER's last tick left.  on_tick(t, (h0, h1), batch) after every tick."""
    P, blob, sid = sc['P'], sc['blob'], sc['sid']
    hs = [orc.cpuref_create(C.byref(P), blob, len(blob), C.byref(sc['S0'])), orc.cpuref_create(C.byref(P), blob, len(blob), C.byref(sc['S1']))]
    orc.cpuref_set_guid(hs[1], 1)   # the second car of the simulator (Car.h physicsGUID)
    slips = (pc.SlipState * 2)()

    def step_both(a):   # a: float32[2][2]
        for c in range(2):
            orc.cpuref_get_slip(hs[c], C.byref(slips[c]))
        for c in range(2):
            orc.cpuref_set_other_slips(hs[c], C.byref(slips[1 - c]), 1)
        for c in range(2):
            orc.cpuref_step_env(hs[c], float(a[c, 0]), float(a[c, 1]))
        if batch is not None:
            batch.step_host(a)
    a = np.zeros((2, 2), np.float32)
    step_both(a)
    if on_tick is not None:
        on_tick(-1, hs, batch)
    n = sc['ticks'] if max_ticks is None else min(sc['ticks'], max_ticks)
    o = pc.StepOut(); a2 = (C.c_float * 2)()
    try:
        for t in range(n):
            if sc['feedback']:
                for c, fn in ((0, orc.cpuref_scenario_feedback), (1, orc.cpuref_scenario_feedback2)):
                    orc.cpuref_get_out(hs[c], C.byref(o)); obs = (C.c_float * 24)(*o.obs); fn(sid, t, obs, a2); a[c] = (a2[0], a2[1])
            else:
                orc.cpuref_scenario_action(sid, t, a2); a[0] = (a2[0], a2[1])
                orc.cpuref_scenario_action2(sid, t, a2); a[1] = (a2[0], a2[1])
            step_both(a)
            if on_tick is not None:
                on_tick(t, hs, batch)
    finally:
        for h in hs:
            orc.cpuref_destroy(h)


def drive(orc, hostlib, sc, batch=None, max_ticks=None, on_tick=None):
    """Step the oracle (handle created here) -- and the 1-car GPU batch, if given -- through scenario sc.  The GPU batch must have
    been created with action mode FULL for sc['full'] scenarios and ENV otherwise.  on_tick(t, handle, batch) after every tick."""
    P, blob, sid = sc['P'], sc['blob'], sc['sid']
    h = orc.cpuref_create(C.byref(P), blob, len(blob), C.byref(sc['S0']))

    def _tele_mode(state_ptr, mode):
        assert hostlib.pdb_teleport_by_mode(C.byref(P), blob, mode, C.c_void_p(state_ptr)) == 0
    hook = C.CFUNCTYPE(None, C.c_void_p, C.c_int)(_tele_mode)
    orc.cpuref_set_auto_teleport_hook(h, C.cast(hook, C.c_void_p))
    zero2 = np.zeros((1, 2), np.float32); zero8 = np.zeros((1, 8), np.float32); zero8[0, 5] = -1.0

    def step_both(a2=None, a8=None):
        if a8 is not None:
            orc.cpuref_step_controls(h, a8.ctypes.data_as(C.c_void_p))
            if batch is not None:
                batch.step_host(a8.reshape(1, 8))
        else:
            orc.cpuref_step_env(h, float(a2[0]), float(a2[1]))
            if batch is not None:
                if sc['full']:   # a FULL-mode batch fed an env action: the env mapping on the host (steer, gas = envGas(a1))
                    a = zero8.copy(); a[0, 0] = a2[0]; a[0, 4] = orc.cpuref_env_gas(C.c_float(float(a2[1])))
                    batch.step_host(a)
                else:
                    batch.step_host(np.asarray(a2, np.float32).reshape(1, 2))

    def teleport_both(k):
        abc = (C.c_float * 3)()
        kind = orc.cpuref_scenario_teleport(sid, k, abc)
        s = pc.DynState(); orc.cpuref_get_state(h, C.byref(s))
        off = [float(x) for x in abc]

        def args(st):   # a location teleport is relative to where the chassis is (float32 sums, like the harness)
            return [float(np.float32(st.body[0].pos[i]) + np.float32(off[i])) for i in range(3)] if kind == 2 else off
        teleport_state(hostlib, P, blob, C.byref(s), kind, *args(s))
        orc.cpuref_set_state(h, C.byref(s))
        if batch is not None:
            g = batch.get_state()
            teleport_state(hostlib, P, blob, C.byref(g[0]), kind, *args(g[0]))
            batch.set_state(g)
    step_both(a2=(0.0, 0.0))                      # env.reset(): teleport (already in S0) + step([0, 0])
    if on_tick is not None:
        on_tick(-1, h, batch)                     # the goldens' first record (tick -1) is taken here
    n = sc['ticks'] if max_ticks is None else min(sc['ticks'], max_ticks)
    o = pc.StepOut()
    a2 = (C.c_float * 2)(); a8 = np.zeros(8, np.float32)
    try:
        for t in range(n):
            if sc['reset_every'] and t > 0 and t % sc['reset_every'] == 0:
                teleport_both(t // sc['reset_every'] - 1)
                step_both(a2=(0.0, 0.0))
            if sc['boost_at'] and t == sc['boost_at']:
                s = pc.DynState(); orc.cpuref_get_state(h, C.byref(s))
                for b in range(P.numBodies):
                    s.body[b].lvel[2] = 50.0
                orc.cpuref_set_state(h, C.byref(s))
                if batch is not None:
                    g = batch.get_state()
                    for b in range(P.numBodies):
                        g[0].body[b].lvel[2] = 50.0
                    batch.set_state(g)
            if sc['feedback']:
                orc.cpuref_get_out(h, C.byref(o))
                obs = (C.c_float * 24)(*o.obs)
                orc.cpuref_scenario_feedback(sid, t, obs, a2)
                step_both(a2=(a2[0], a2[1]))
            elif sc['full']:
                orc.cpuref_scenario_controls(sid, t, a8.ctypes.data_as(C.c_void_p))
                step_both(a8=a8)
            else:
                orc.cpuref_scenario_action(sid, t, a2)
                step_both(a2=(a2[0], a2[1]))
            if on_tick is not None:
                on_tick(t, h, batch)
    finally:
        orc.cpuref_destroy(h)

"""GPU-vs-oracle parity helpers shared by tests/test_gpu_parity.py, __graft_entry__.smoke() and bench.py."""
import ctypes as C, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
import pdb_ctypes as pc
import oracle_ctypes

_DT = np.dtype(pc.DynState)


def _leaves():
    """(name, byte offset, numpy scalar dtype) of every leaf of pdb_dyn_state, padding fields skipped, in declaration order"""
    out = []

    def walk(prefix, dt, off):
        if dt.names:
            for n in dt.names:
                if n.startswith('_pad') or n == 'lawTick':   # (lawTick: the device law's row counter, pdb_set_law -- not reference state; tests/test_device_law.py checks it)
                    continue
                walk(prefix + '.' + n if prefix else n, dt.fields[n][0], off + dt.fields[n][1])
        elif dt.subdtype:
            base, shape = dt.subdtype
            cnt = int(np.prod(shape))
            for i in range(cnt):
                walk('%s[%d]' % (prefix, i), base, off + i * base.itemsize)
        else:
            out.append((prefix, off, dt))
    walk('', _DT, 0)
    return out


_LAYOUT = None


def _layout():
    """per scalar type: the leaves' offsets (in units of the type) and their places in the float / integer vectors -- worked out once"""
    global _LAYOUT
    if _LAYOUT is None:
        fn, inn, groups = [], [], {}
        for name, off, dt in _leaves():
            isf = dt.kind == 'f'
            pos = len(fn) if isf else len(inn)
            (fn if isf else inn).append(name)
            assert off % dt.itemsize == 0
            g = groups.setdefault((dt.str, isf), ([], []))
            g[0].append(off // dt.itemsize); g[1].append(pos)
        _LAYOUT = (fn, inn, [(np.dtype(k[0]), k[1], np.array(v[0]), np.array(v[1])) for k, v in groups.items()])
    return _LAYOUT


def state_vectors(st):
    """Flatten one DynState into (float vector, int vector, names) for comparison."""
    fn, inn, groups = _layout()
    raw = bytes(st)
    fl = np.zeros(len(fn)); it = np.zeros(len(inn), dtype=np.int64)
    for dt, isf, idx, pos in groups:
        v = np.frombuffer(raw, dtype=dt)[idx]
        if isf:
            fl[pos] = v
        else:
            it[pos] = v
    return fl, it, fn, inn


# per-field scale floor for the relative error: max(|ref|, 1e-3 * scale)
def field_scale(name):
    if '.pos' in name or 'contactPoint' in name or 'ContactPoint' in name or 'pointCachePos' in name:
        return 1.0          # metres
    if '.R[' in name or '.q[' in name or 'contactNormal' in name:
        return 1.0
    if 'lvel' in name or 'lastVelocity' in name or name == 'speed':
        return 10.0         # m/s
    if 'avel' in name:
        return 1.0          # rad/s
    if name.endswith('.load') or 'Fx' in name or 'Fy' in name:
        return 1000.0       # N
    if 'Mz' in name or 'localMX' in name:
        return 100.0
    if 'angularVelocity' in name or 'oldAngularVelocity' in name or 'Vel' in name or 'rootVelocity' in name:
        return 10.0         # rad/s
    if '.T[' in name or 'coreTemp' in name or 'practicalTemp' in name or 'waterT' in name:
        return 20.0
    return 1.0


_GROUPS = {}


def _groups(fn):
    """Index groups: components of one 3-vector / quaternion / rotation matrix share a denominator (the group's
    max-norm) -- a component that happens to be ~0 has no meaningful relative error of its own; scalars stand alone."""
    key = tuple(fn)
    if key not in _GROUPS:
        g = {}
        for i, n in enumerate(fn):
            base = n[:n.rindex('[')] if n.endswith(']') and any(t in n for t in ('.pos[', '.q[', '.R[', '.lvel[', '.avel[', 'contactPoint[', 'ContactPoint[', 'contactNormal[', 'lastVelocity[', 'pointCachePos[')) else n
            g.setdefault(base, []).append(i)
        gid = np.zeros(len(fn), dtype=np.int64)
        for k, (b, idx) in enumerate(g.items()):
            gid[idx] = k
        _GROUPS[key] = (gid, len(g))
    return _GROUPS[key]


def compare_states(sg, sc):
    """Worst relative float deviation |x_gpu - x_cpu| / max(||x_cpu||_group, 1e-3 * scale), integer mismatches."""
    fg, ig, fn, inn = state_vectors(sg)
    fc, ic, _, _ = state_vectors(sc)
    scale = np.array([field_scale(n) for n in fn])
    gid, ng = _groups(fn)
    gmax = np.zeros(ng)
    np.maximum.at(gmax, gid, np.abs(fc))
    den = np.maximum(gmax[gid], 1e-3 * scale)
    rel = np.abs(fg - fc) / den
    rel[np.isnan(fg) != np.isnan(fc)] = np.inf
    rel[np.isnan(fg) & np.isnan(fc)] = 0
    w = int(np.argmax(rel))
    bad_int = [(inn[i], int(ig[i]), int(ic[i])) for i in np.where(ig != ic)[0]]
    return float(rel[w]), fn[w], float(fg[w]), float(fc[w]), bad_int


def make_actions(n, seed, lo=-0.3, hi=0.3):
    """config-2 style: per-car constant action, steer ~ U(-0.3,0.3), a1 ~ U(-1,1), PCG-free numpy."""
    import sharding
    return sharding.global_actions(n, seed, lo, hi)


def run_parity(n_cars=8, ticks=200, seed=1234, resync=False, verbose=False, check_every=1, actions_fn=None, model='ks_toyota_ae86_drift', track='flat',
               on_tick=None, params_fn=None, spread=None, threads=None, lane_params_fn=None, lane_setups=False):
    """Step `n_cars` cars for `ticks` ticks on the GPU (through the C ABI) and in the CPU oracle, from the same
    initial state.  resync=True re-injects the oracle state into the GPU before every tick (single-tick parity).
    spread=(lo, hi): car i starts from teleportCarToSpline(lo + (hi - lo) * i / n_cars) (the product's host function, for both
    sides) instead of the start pose.  threads: step the oracle's cars on that many host threads (the calls release the GIL).
    lane_params_fn(i, Pi): edit car i's own copy of the block (the eight env tunes, the scoring variables); the oracle steps car i with it, the
    GPU batch is created with the common block and gets the copies through set_lane_tunes (pdb_set_lane_tunes).
    Returns the worst relative deviation over all cars, ticks and float fields; raises on integer mismatches."""
    import pdbatch
    P = pdbatch.packed_params(model + '.env')
    if params_fn is not None:
        params_fn(P)
    trk = track if isinstance(track, (bytes, bytearray)) else pdbatch.synthetic_track(track)
    lib = pc.load_product(); orc = oracle_ctypes.load_oracle(portable_math=True)
    S0 = pc.DynState()
    assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
    b = pdbatch.Batch(n_cars, P, trk, device=0, action_mode=1)
    Pl = [P] * n_cars
    if lane_params_fn is not None:
        Pl = []
        for i in range(n_cars):
            Pi = pc.CarParams.from_buffer_copy(bytes(P)); lane_params_fn(i, Pi); Pl.append(Pi)
        b.set_lane_tunes(Pl)
        if lane_setups:   # the rest of SetupManager's tunes, lane by lane (pdb_set_lane_setups)
            b.set_lane_setups(Pl)
    if spread is None:
        hs = [orc.cpuref_create(C.byref(Pl[i]), trk, len(trk), C.byref(S0)) for i in range(n_cars)]
    else:
        init = (pc.DynState * n_cars)()
        for i in range(n_cars):
            C.memmove(C.byref(init[i]), C.byref(S0), C.sizeof(S0))
            assert lib.pdb_teleport_to_spline(C.byref(P), trk, C.c_float(spread[0] + (spread[1] - spread[0]) * i / n_cars), C.byref(init[i])) == 0
        b.set_state(init)
        hs = [orc.cpuref_create(C.byref(Pl[i]), trk, len(trk), C.byref(init[i])) for i in range(n_cars)]
    pool = None
    if threads and threads > 1 and not P.autoTeleport:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(threads)
    hook = None
    if P.autoTeleport:   # setCarAutoTeleport: the oracle's in-tick Car::teleportByMode is the product's HOST function
        def _tele(state_ptr, mode):
            assert lib.pdb_teleport_by_mode(C.byref(P), trk, mode, C.c_void_p(state_ptr)) == 0
        hook = C.CFUNCTYPE(None, C.c_void_p, C.c_int)(_tele)
        for h in hs:
            orc.cpuref_set_auto_teleport_hook(h, C.cast(hook, C.c_void_p))
    acts = make_actions(n_cars, seed)
    worst = 0.0; worst_info = None
    try:
        for t in range(ticks):
            a = acts if actions_fn is None else actions_fn(t, acts)
            if resync and t > 0:
                arr = (pc.DynState * n_cars)()
                cts = ((pc.Contact * pc.MAX_CONTACTS) * n_cars)()
                for i in range(n_cars):
                    orc.cpuref_get_state(hs[i], C.byref(arr[i]))
                    orc.cpuref_get_contacts(hs[i], C.byref(cts[i]))
                b.set_state(arr)
                b.set_contacts(cts)
            b.step_host(a)
            if pool is None:
                for i in range(n_cars):
                    orc.cpuref_step_env(hs[i], float(a[i, 0]), float(a[i, 1]))
            else:
                list(pool.map(lambda i: orc.cpuref_step_env(hs[i], float(a[i, 0]), float(a[i, 1])), range(n_cars)))
            if (t % check_every) == 0 or t == ticks - 1:
                sg = b.get_state()
                cg = b.get_contacts()
                for i in range(n_cars):
                    sc = pc.DynState()
                    orc.cpuref_get_state(hs[i], C.byref(sc))
                    rel, name, vg, vc, bad_int = compare_states(sg[i], sc)
                    if sc.numContacts > 0 and not bad_int:   # the live contact joints, bit for bit
                        cc = (pc.Contact * pc.MAX_CONTACTS)()
                        orc.cpuref_get_contacts(hs[i], C.byref(cc))
                        if bytes(cc)[:32 * sc.numContacts] != bytes(cg[i])[:32 * sc.numContacts]:
                            raise AssertionError('contact joints differ: car %d tick %d' % (i, t))
                    if bad_int:
                        raise AssertionError('integer state mismatch car %d tick %d: %s' % (i, t, bad_int[:5]))
                    if rel > worst:
                        worst = rel; worst_info = (t, i, name, vg, vc)
                    if on_tick is not None:
                        on_tick(t, i, sg[i], sc)
    finally:
        b.close()
        if pool is not None:
            pool.shutdown()
        for h in hs:
            orc.cpuref_destroy(h)
    if verbose:
        print('parity: n=%d ticks=%d resync=%s worst rel=%.3e at %s' % (n_cars, ticks, resync, worst, worst_info))
    return worst


def run_replicated(n_cars, distinct, ticks, track, model='ks_toyota_ae86_drift', seed=7, spread=(0.0, 1.0), check_every=50, partitions=None, threads=8,
                   law=None, resets=None, verbose=False, lane_params_fn=None, host_pipeline=False):
    """Full-size parity by replication (BASELINE's car counts, the oracle at `distinct` cars): `distinct` different (start point on the lap, input) pairs tiled over
    a batch of n_cars.  Size-independent properties, checked every `check_every` ticks and at the end:
      (1) every replica of a representative holds the byte-identical record AND the byte-identical live contact joints wherever it sits in the batch (any
          workgroup, any partition, any place in the contact pass's queue, any snapshot slot);
      (2) the `distinct` representatives equal the CPU oracle: every float and integer of the record, the live contact joints byte for byte.
    law=None: per-car constant actions (make_actions), the ticks between two checks enqueued back to back (partitions: free-running ranges through pdb_step_ring).
    law(obs[m,24] float32, t, ids[m]) -> actions[m,2] float32: closed loop, evaluated on the host from the observation rows in float32 -- for the GPU batch from the rows
    pdb_step_host returns, for the oracle from cpuref_get_out -- so equal observations give equal actions.
    resets=(bits, mode): the env's episode rule on the host (projectd_env.py:173-227 without the reward sums): a car whose output flags meet `bits` is teleported by
    Car::teleportByMode(mode) before its next tick, which it takes with the zero action; GPU: pdb_reset_mode, oracle: the product's host function on the oracle's record.
    host_pipeline (with a law and partitions): the GPU batch is fed the way BASELINE configs[4] words it -- "actions fed from host", every tick -- through the library's
    pipelined form: the law's rows go into the page-locked action mirror, every partition's upload + tick + download is enqueued on its own stream
    (pdb_step_host_partition), and the rows are read out of the output mirror as each partition's download lands (pdb_wait_host_partition).
    lane_params_fn(k, Pk): representative k's own copy of the car block (tunes, scoring variables): the oracle steps it with that block, the GPU batch carries it as the
    rows of its replicas' lanes (pdb_set_lane_tunes + pdb_set_lane_setups).
    Returns dict(worst, max_in_contact (cars with live joints at a check, over the whole batch), contact_checks, resets)."""
    import pdbatch
    from concurrent.futures import ThreadPoolExecutor
    P = pdbatch.packed_params(model + '.env')
    trk = track if isinstance(track, (bytes, bytearray)) else (pdbatch.reference_track(track) if track in pdbatch.REFERENCE_TRACKS else pdbatch.synthetic_track(track))
    lib = pc.load_product(); orc = oracle_ctypes.load_oracle(portable_math=True)
    S0 = pc.DynState()
    assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
    rep = np.arange(n_cars) % distinct                      # car i replicates representative i % distinct: the replicas of one representative are spread over every partition
    init = (pc.DynState * distinct)()
    for k in range(distinct):
        C.memmove(C.byref(init[k]), C.byref(S0), C.sizeof(S0))
        assert lib.pdb_teleport_to_spline(C.byref(P), trk, C.c_float(spread[0] + (spread[1] - spread[0]) * k / distinct), C.byref(init[k])) == 0
        init[k].randState = 1 + 7919 * k                   # (the Random teleport mode draws from the car's own generator)
    raw0 = np.frombuffer(bytes(init), dtype=np.uint8).reshape(distinct, C.sizeof(pc.DynState))
    allinit = (pc.DynState * n_cars).from_buffer_copy(raw0[rep].tobytes())
    b = pdbatch.Batch(n_cars, P, trk, device=0, action_mode=1)
    b.set_state(allinit)
    if partitions:
        b.set_partitions(partitions)
    Pk = [P] * distinct
    if lane_params_fn is not None:
        Pk = []
        for k in range(distinct):
            Q = pc.CarParams.from_buffer_copy(bytes(P)); lane_params_fn(k, Q); Pk.append(Q)
        lanes = [Pk[int(r)] for r in rep]
        b.set_lane_tunes(lanes); b.set_lane_setups(lanes)
    hs = [orc.cpuref_create(C.byref(Pk[k]), trk, len(trk), C.byref(init[k])) for k in range(distinct)]
    pool = ThreadPoolExecutor(threads)
    base = make_actions(distinct, seed)
    base[:, 1] = np.abs(base[:, 1])                         # enough throttle that every car gets going
    res = dict(worst=0.0, max_in_contact=0, contact_checks=0, resets=0)
    nb = C.sizeof(pc.DynState)
    off_nc = pc.DynState.numContacts.offset

    def check(t):
        st = b.get_state(); ct = b.get_contacts()
        raw = np.frombuffer(st, dtype=np.uint8).reshape(n_cars, nb)
        craw = np.frombuffer(ct, dtype=np.uint8).reshape(n_cars, pc.MAX_CONTACTS * 32)
        nc = raw[:, off_nc:off_nc + 4].copy().view(np.int32)[:, 0]
        res['max_in_contact'] = max(res['max_in_contact'], int((nc > 0).sum()))
        for k in range(distinct):
            reps = raw[k::distinct]
            if not (reps == reps[0]).all():
                bad = int(np.argmax((reps != reps[0]).any(axis=1)))
                raise AssertionError('tick %d: replicas of representative %d differ (first bad lane %d, byte %d)' % (t, k, bad * distinct + k, int(np.argmax(reps[bad] != reps[0]))))
            live = int(nc[k]) * 32
            if live > 0:
                creps = craw[k::distinct, :live]
                if not (creps == creps[0]).all():
                    raise AssertionError('tick %d: live contact joints of the replicas of representative %d differ' % (t, k))
            sc = pc.DynState(); orc.cpuref_get_state(hs[k], C.byref(sc))
            rel, name, vg, vc, bad_int = compare_states(st[k], sc)
            if bad_int:
                raise AssertionError('tick %d: integer state mismatch, representative %d: %s' % (t, k, bad_int[:5]))
            if sc.numContacts > 0:
                cc = (pc.Contact * pc.MAX_CONTACTS)(); orc.cpuref_get_contacts(hs[k], C.byref(cc))
                if bytes(cc)[:32 * sc.numContacts] != craw[k, :32 * sc.numContacts].tobytes():
                    raise AssertionError('tick %d: contact joints differ from the oracle, representative %d' % (t, k))
                res['contact_checks'] += 1
            if rel > res['worst']:
                res['worst'] = rel
                if verbose:
                    print('tick %d representative %d: %s gpu %r oracle %r rel %.3e' % (t, k, name, vg, vc, rel))
    try:
        if law is None:
            acts = base[rep]
            b.step_host(acts)                               # uploads the actions; tick 0
            list(pool.map(lambda k: orc.cpuref_step_env(hs[k], float(base[k, 0]), float(base[k, 1])), range(distinct)))
            t = 1
            while t < ticks:
                m = min(check_every, ticks - t)
                b.step_ring(m, join=True)                     # plain launches, the contact pass's grid following the load (one part: on the batch's own stream)

                def many(k):
                    for _ in range(m):
                        orc.cpuref_step_env(hs[k], float(base[k, 0]), float(base[k, 1]))
                list(pool.map(many, range(distinct)))
                t += m
                check(t)
        else:
            ids = np.arange(distinct)
            obs_o = np.zeros((distinct, 24), np.float32); obs_g = np.zeros((n_cars, 24), np.float32)
            pend_o = np.zeros(distinct, bool); pend_g = np.zeros(n_cars, bool)
            oo = pc.StepOut()
            for t in range(ticks):
                # the law on the representatives' rows, the same call (same shapes) for both sides; the replicas take their representative's action -- their
                # observation rows are held equal to it every tick (below), their records at every check
                ag = law(obs_g[:distinct], t, ids).astype(np.float32)[rep]; ao = law(obs_o, t, ids).astype(np.float32)
                if resets is not None:
                    bits, mode = resets
                    if pend_g.any():
                        b.reset(pend_g.astype(np.uint8), mode); ag[pend_g] = 0.0
                    for k in np.where(pend_o)[0]:
                        s = pc.DynState(); orc.cpuref_get_state(hs[k], C.byref(s))
                        assert lib.pdb_teleport_by_mode(C.byref(P), trk, mode, C.byref(s)) == 0
                        orc.cpuref_set_state(hs[k], C.byref(s)); ao[k] = 0.0
                        res['resets'] += 1
                if host_pipeline:
                    assert partitions and resets is None
                    ha, ho = b.host_mirrors()
                    ha[:, :2] = ag
                    for p in range(partitions):
                        b.step_host_partition(p)
                    list(pool.map(lambda k: orc.cpuref_step_env(hs[k], float(ao[k, 0]), float(ao[k, 1])), range(distinct)))   # (the oracle steps while the partitions' ticks are in flight)
                    for p in range(partitions):
                        b.wait_host_partition(p)
                    out = {'obs': np.array(ho['obs']), 'flags': np.array(ho['flags'])}
                else:
                    out = b.step_host(ag)
                    list(pool.map(lambda k: orc.cpuref_step_env(hs[k], float(ao[k, 0]), float(ao[k, 1])), range(distinct)))
                obs_g = np.array(out['obs'], dtype=np.float32); fg = np.array(out['flags'])
                if n_cars % distinct == 0:
                    assert (obs_g.reshape(-1, distinct, 24).view(np.uint32) == obs_g[:distinct].view(np.uint32)).all() and (fg.reshape(-1, distinct) == fg[:distinct]).all(), 'tick %d: a replica\'s output row differs' % t
                fo = np.zeros(distinct, np.int32)
                for k in range(distinct):
                    orc.cpuref_get_out(hs[k], C.byref(oo)); obs_o[k] = np.frombuffer(oo, dtype=np.float32, count=24); fo[k] = oo.flags
                if resets is not None:
                    was_g, was_o = pend_g, pend_o
                    pend_g = ((fg & resets[0]) != 0) & ~was_g; pend_o = ((fo & resets[0]) != 0) & ~was_o   # (the reset tick's own termination is discarded)
                if (t + 1) % check_every == 0 or t == ticks - 1:
                    check(t + 1)
    finally:
        b.close(); pool.shutdown()
        for h in hs:
            orc.cpuref_destroy(h)
    return res

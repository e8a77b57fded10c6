#!/usr/bin/env python3
"""ORACLE / TEST INFRASTRUCTURE (build container only).  Assembles the base directory the
reference-TU harness runs in: oracle/_ref/base/{cfg/sim.ini, content/cars/<model>/data, content/tracks/*}.
Car data and sim.ini are copied from /root/reference into oracle/_ref (git-ignored build output);
nothing from the reference enters the repository history."""
import os, shutil, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, '..', 'projectd-core_amd'))
import synthetic_tracks as gen_track
REF = '/root/reference'
def make_multilink_car(base, src='ks_toyota_supra_mkiv_drift', dst='pdb_ml_supra'):
    """No shipped car uses the reference's multilink suspension (Car/SuspensionML.cpp), so there is nothing to run it on.  This
    derives one: the Supra's data directory with suspensions.ini rewritten to TYPE=ML front and rear, the five links
    (JOINTn_CAR / JOINTn_TYRE) taken from the double wishbone's pick-up points (top rear/front, bottom rear/front, steering
    link).  The reference TUs load and run it like any other car, which pins the oracle's restatement of that class."""
    import re
    s = os.path.join(REF, 'content', 'cars', src, 'data'); d = os.path.join(base, 'content', 'cars', dst, 'data')
    if os.path.isdir(d):
        shutil.rmtree(d)
    shutil.copytree(s, d)
    os.system('chmod -R u+w "%s"' % d)
    p = os.path.join(d, 'suspensions.ini')
    raw = open(p, newline='').read()
    eol = '\r\n' if '\r\n' in raw else '\n'          # keep the file's own line endings (the reference's reader does not strip CR)
    text = raw.replace('\r\n', '\n')
    out = []
    for sec in re.split(r'(?m)^(?=\[)', text):
        m = re.match(r'\[(FRONT|REAR)\]', sec)
        if m:
            kv = dict(l.split('=', 1) for l in sec.splitlines()[1:] if '=' in l)
            links = [('WBCAR_TOP_REAR', 'WBTYRE_TOP'), ('WBCAR_TOP_FRONT', 'WBTYRE_TOP'), ('WBCAR_BOTTOM_REAR', 'WBTYRE_BOTTOM'),
                     ('WBCAR_BOTTOM_FRONT', 'WBTYRE_BOTTOM'), ('WBCAR_STEER', 'WBTYRE_STEER')]
            sec = sec.replace('TYPE=DWB', 'TYPE=ML').rstrip('\n') + '\n'
            for i, (c, t) in enumerate(links):
                sec += 'JOINT%d_CAR=%s\nJOINT%d_TYRE=%s\n' % (i, kv[c].strip(), i, kv[t].strip())
            sec += '\n'
        out.append(sec)
    open(p, 'w', newline='').write(''.join(out).replace('\n', eol))


def make_heave_car(base, src='ks_mazda_rx7_tuned', dst='pdb_heave_rx7'):
    """No shipped car carries [HEAVE_FRONT] / [HEAVE_REAR] (Car/HeaveSpring.cpp stays idle with k = 0).  This derives one: the
    tuned RX-7 (double wishbones all round) with a third spring/damper across each axle, stiff enough that its bump stops and
    packer come into play on the mountain road."""
    s = os.path.join(REF, 'content', 'cars', src, 'data'); d = os.path.join(base, 'content', 'cars', dst, 'data')
    if os.path.isdir(d):
        shutil.rmtree(d)
    shutil.copytree(s, d)
    os.system('chmod -R u+w "%s"' % d)
    p = os.path.join(d, 'suspensions.ini')
    raw = open(p, newline='').read()
    eol = '\r\n' if '\r\n' in raw else '\n'
    def section(name, k, rod, up, dn, packer, bump, rebound):
        return ['', '[%s]' % name, 'SPRING_RATE=%d' % k, 'PROGRESSIVE_SPRING_RATE=15000', 'ROD_LENGTH=%.3f' % rod, 'BUMPSTOP_UP=%.3f' % up,
                'BUMPSTOP_DN=%.3f' % dn, 'PACKER_RANGE=%.3f' % packer, 'BUMP_STOP_RATE=0', 'DAMP_BUMP=%d' % bump, 'DAMP_REBOUND=%d' % rebound,
                'DAMP_FAST_BUMP=%d' % (bump // 2), 'DAMP_FAST_REBOUND=%d' % (rebound // 2), 'DAMP_FAST_BUMPTHRESHOLD=0.08', 'DAMP_FAST_REBOUNDTHRESHOLD=0']
    extra = section('HEAVE_FRONT', 30000, 0.02, 0.012, 0.02, 0.03, 1800, 3500) + section('HEAVE_REAR', 22000, 0.015, 0.015, 0.025, 0.035, 1500, 3000) + ['']
    open(p, 'w', newline='').write(raw.rstrip('\r\n') + eol + eol.join(extra))


def make_fwd_car(base, src='ks_toyota_ae86_drift', dst='pdb_fwd_ae86'):
    """No shipped car is front-wheel drive; Drivetrain::step2WD serves both.  The AE86 with [TRACTION] TYPE=FWD pins that branch."""
    s = os.path.join(REF, 'content', 'cars', src, 'data'); d = os.path.join(base, 'content', 'cars', dst, 'data')
    if os.path.isdir(d):
        shutil.rmtree(d)
    shutil.copytree(s, d)
    os.system('chmod -R u+w "%s"' % d)
    p = os.path.join(d, 'drivetrain.ini')
    raw = open(p, newline='').read()
    assert 'TYPE=RWD' in raw
    open(p, 'w', newline='').write(raw.replace('TYPE=RWD', 'TYPE=FWD'))


def make_slip_car(base, src='ks_toyota_ae86_drift', dst='pdb_slip_ae86'):
    """No shipped car carries [SLIPSTREAM] in its aero.ini (AeroMap.cpp:25-29 reads it where it is there; SlipStream.h's defaults apply otherwise: a wake of a quarter
    of a second's travel).  The AE86 with EFFECT_GAIN_MULT = 1.5 and SPEED_FACTOR_MULT = 4 (a wake of one second's travel) pins both, and lets two cars a few car
    lengths apart draft each other in the two-car fixtures."""
    s = os.path.join(REF, 'content', 'cars', src, 'data'); d = os.path.join(base, 'content', 'cars', dst, 'data')
    if os.path.isdir(d):
        shutil.rmtree(d)
    shutil.copytree(s, d)
    os.system('chmod -R u+w "%s"' % d)
    p = os.path.join(d, 'aero.ini')
    raw = open(p, newline='').read()
    eol = '\r\n' if '\r\n' in raw else '\n'
    assert 'SLIPSTREAM' not in raw
    open(p, 'w', newline='').write(raw.rstrip('\r\n') + eol + eol.join(['', '[SLIPSTREAM]', 'EFFECT_GAIN_MULT=1.5', 'SPEED_FACTOR_MULT=4.0', '']))


def make_cold_car(base, src='ks_mazda_rx7_tuned', dst='pdb_cold_rx7'):
    """Four branches no shipped car takes: [OVERLAP] (a torque ripple away from the ideal rpm, Engine.cpp:96-101,300-307), [THROTTLE_RESPONSE] (a second throttle curve blended in by rpm, Engine.cpp:150-154,344-366),
    [COAST_SETTINGS] (a throttle offset rising with rpm, Engine.cpp:61-67,198-207) and [EBB] (brake bias following the front axle's
    share of the load, BrakeSystem.cpp:28-32,92-113).  The tuned RX-7 with the three sections added pins them."""
    s = os.path.join(REF, 'content', 'cars', src, 'data'); d = os.path.join(base, 'content', 'cars', dst, 'data')
    if os.path.isdir(d):
        shutil.rmtree(d)
    shutil.copytree(s, d)
    os.system('chmod -R u+w "%s"' % d)
    for fn, extra in (('engine.ini', ['', '[THROTTLE_RESPONSE]', 'RPM_REFERENCE=6000', 'LUT=(|0=0|20=35|50=72|80=93|100=100|)', '', '[COAST_SETTINGS]', 'LUT=(|0=0.0|1=0.08|2=0.15|)',
                                      'DEFAULT=1', 'ACTIVATION_RPM=1500', '', '[OVERLAP]', 'FREQUENCY=1.3', 'GAIN=0.02', 'IDEAL_RPM=4200', '']),
                      ('brakes.ini', ['', '[EBB]', 'FRONT_SHARE_MULTIPLIER=1.25', ''])):
        p = os.path.join(d, fn)
        raw = open(p, newline='').read()
        eol = '\r\n' if '\r\n' in raw else '\n'
        open(p, 'w', newline='').write(raw.rstrip('\r\n') + eol + eol.join(extra))


def make_curves_car(base, src='ks_toyota_ae86_drift', dst='pdb_curves_ae86'):
    """The LUT forms of the tyre's load sensitivity and camber factor, which no shipped car carries: DY_CURVE / DX_CURVE (Tyre.cpp:161-165 ->
    SCTM::getStaticDY / getStaticDX, TyreModel.cpp:121-146) and DCAMBER_LUT (Tyre.cpp:207-211 -> TyreModel.cpp:49-57), read through the
    natural cubic spline of Curve::getCubicSplineValue (Core/Curve.cpp:117-126) -- the front compound with DCAMBER_LUT_SMOOTH=1 (spline),
    the rear one without (piecewise linear).  The AE86 with the keys added to its first compounds pins them."""
    s = os.path.join(REF, 'content', 'cars', src, 'data'); d = os.path.join(base, 'content', 'cars', dst, 'data')
    if os.path.isdir(d):
        shutil.rmtree(d)
    shutil.copytree(s, d)
    os.system('chmod -R u+w "%s"' % d)
    p = os.path.join(d, 'tyres.ini')
    raw = open(p, newline='').read()
    eol = '\r\n' if '\r\n' in raw else '\n'
    add = {'[FRONT]': ['DY_CURVE=(|0=1.46|800=1.38|1600=1.30|2400=1.235|3300=1.18|4500=1.12|6000=1.07|8000=1.03|)',
                       'DX_CURVE=(|0=1.52|1000=1.41|2000=1.33|3000=1.27|4500=1.20|6500=1.14|9000=1.09|)',
                       'DCAMBER_LUT=(|-8=1.05|-5=1.062|-3=1.05|-1.5=1.03|0=1.0|1.5=0.965|3=0.93|6=0.85|)', 'DCAMBER_LUT_SMOOTH=1'],
           '[REAR]': ['DY_CURVE=(|0=1.44|900=1.36|1800=1.285|2700=1.225|3600=1.175|5000=1.115|7000=1.06|)',
                      'DX_CURVE=(|0=1.50|1200=1.39|2400=1.31|3600=1.245|5200=1.18|7500=1.12|)',
                      'DCAMBER_LUT=(|-6=1.055|-3=1.045|0=1.0|3=0.935|6=0.86|)', 'DCAMBER_LUT_SMOOTH=0']}
    out = []
    for line in raw.split(eol):
        out.append(line)
        if line.strip() in add:
            out.extend(add[line.strip()])
    open(p, 'w', newline='').write(eol.join(out))


def make_ground_effect_car(base, src='dthwsh_mazda_rx7_fc3s_sr20', dst='pdb_gh_fc3s'):
    """Wings whose lift and drag coefficients follow their height above the plane of the tyres' contact points (LUT_GH_CL / LUT_GH_CD, Wing.cpp:38-44,
    134-139,181-186; Car::getPointGroundHeight, Car.cpp:1380-1405): every shipped car leaves the two keys empty.  The FC3S with LUTs on its front wing
    (both), its body (lift only) and its rear wing (drag only) pins them -- over the mountain road, where pitch and roll move the heights about."""
    s = os.path.join(REF, 'content', 'cars', src, 'data'); d = os.path.join(base, 'content', 'cars', dst, 'data')
    if os.path.isdir(d):
        shutil.rmtree(d)
    shutil.copytree(s, d)
    os.system('chmod -R u+w "%s"' % d)
    open(os.path.join(d, 'pdb_gh_cl.lut'), 'w').write('-0.05|1.75\n0.0|1.62\n0.04|1.44\n0.08|1.3\n0.15|1.17\n0.25|1.08\n0.4|1.02\n0.8|1.0\n')
    open(os.path.join(d, 'pdb_gh_cd.lut'), 'w').write('-0.05|1.3\n0.0|1.24\n0.06|1.15\n0.15|1.07\n0.3|1.02\n0.8|1.0\n')
    p = os.path.join(d, 'aero.ini')
    raw = open(p, newline='').read()
    eol = '\r\n' if '\r\n' in raw else '\n'
    want = {'[WING_1]': {'LUT_GH_CL': 'pdb_gh_cl.lut', 'LUT_GH_CD': 'pdb_gh_cd.lut'}, '[WING_0]': {'LUT_GH_CL': 'pdb_gh_cl.lut'}, '[WING_2]': {'LUT_GH_CD': 'pdb_gh_cd.lut'}}
    out = []; sec = None
    for line in raw.split(eol):
        t = line.strip()
        if t.startswith('['):
            sec = t
        k = t.split('=')[0]
        if sec in want and k in want[sec]:
            line = '%s=%s' % (k, want[sec][k])
        out.append(line)
    open(p, 'w', newline='').write(eol.join(out))


def make_aerodata_car(base, src='ks_toyota_ae86_drift', dst='pdb_aerodata_ae86'):
    """An aero.ini without wings: AeroMap's own drag (with its yaw / pitch sensitivities and the angular drag) and lift from [DATA]
    (AeroMap.cpp:49-58,85-139) -- every shipped car has [WING_n] sections instead.  The AE86 with its aero.ini replaced pins the path."""
    s = os.path.join(REF, 'content', 'cars', src, 'data'); d = os.path.join(base, 'content', 'cars', dst, 'data')
    if os.path.isdir(d):
        shutil.rmtree(d)
    shutil.copytree(s, d)
    os.system('chmod -R u+w "%s"' % d)
    open(os.path.join(d, 'aero.ini'), 'w', newline='').write('\r\n'.join(['[HEADER]', 'VERSION=2', '', '[DATA]', 'REFERENCE_AREA=1.9', 'FRONT_SHARE=0.42', 'CD=0.36', 'CL=0.14', 'CDX=0.35', 'CDY=0.8', '']))


def make_wingctrl_car(base, src='dthwsh_mazda_rx7_fc3s_sr20', dst='pdb_wingctrl_fc3s'):
    """Wing dynamic controllers (aero.ini [DYNAMIC_CONTROLLER_n], AeroMap.cpp:66-82, Car/WingDynamicController.cpp, Wing.cpp:105-124): no shipped car has them.
    The FC3S with four: the rear wing's angle rising with speed and, on top, with the brake pedal (an air brake); the front wing's scaled down with
    lateral g; the body's nudged by the throttle -- ADD and MULT, filters, limits that bind."""
    s = os.path.join(REF, 'content', 'cars', src, 'data'); d = os.path.join(base, 'content', 'cars', dst, 'data')
    if os.path.isdir(d):
        shutil.rmtree(d)
    shutil.copytree(s, d)
    os.system('chmod -R u+w "%s"' % d)
    open(os.path.join(d, 'pdb_wc_speed.lut'), 'w').write('0|0\n40|1.5\n90|5\n150|9\n220|12\n')
    open(os.path.join(d, 'pdb_wc_brake.lut'), 'w').write('0|0\n0.2|4\n1|22\n')
    open(os.path.join(d, 'pdb_wc_latg.lut'), 'w').write('-2|0.55\n-0.5|0.9\n0|1\n0.5|0.9\n2|0.55\n')
    open(os.path.join(d, 'pdb_wc_gas.lut'), 'w').write('0|-1\n0.5|0\n1|1.5\n')
    p = os.path.join(d, 'aero.ini')
    raw = open(p, newline='').read()
    eol = '\r\n' if '\r\n' in raw else '\n'
    extra = []
    for i, (wing, inp, comb, lutf, filt, up, dn) in enumerate(((2, 'SPEED_KMH', 'ADD', 'pdb_wc_speed.lut', 0.97, 16, 0), (2, 'BRAKE', 'ADD', 'pdb_wc_brake.lut', 0.9, 30, 0),
                                                                 (1, 'LATG', 'MULT', 'pdb_wc_latg.lut', 0.8, 10, -10), (0, 'GAS', 'ADD', 'pdb_wc_gas.lut', 0.5, 3, -0.5))):
        extra += ['', '[DYNAMIC_CONTROLLER_%d]' % i, 'WING=%d' % wing, 'COMBINATOR=%s' % comb, 'INPUT=%s' % inp, 'LUT=%s' % lutf, 'FILTER=%g' % filt, 'UP_LIMIT=%g' % up, 'DOWN_LIMIT=%g' % dn]
    open(p, 'w', newline='').write(raw.rstrip('\r\n') + eol + eol.join(extra) + eol)


def make_wingctrl2_car(base, src='dthwsh_mazda_rx7_fc3s_sr20', dst='pdb_wingctrl2_fc3s'):
    """The wing controllers' other four inputs (WingDynamicController.cpp:77-107): the rear suspensions' travel (metres; such a wing steps after the
    suspensions of the tick), the longitudinal g and the steering input."""
    s = os.path.join(REF, 'content', 'cars', src, 'data'); d = os.path.join(base, 'content', 'cars', dst, 'data')
    if os.path.isdir(d):
        shutil.rmtree(d)
    shutil.copytree(s, d)
    os.system('chmod -R u+w "%s"' % d)
    open(os.path.join(d, 'pdb_wc_travl.lut'), 'w').write('-0.05|-2\n0|0\n0.06|5\n')
    open(os.path.join(d, 'pdb_wc_travr.lut'), 'w').write('-0.05|0.8\n0|1\n0.06|1.3\n')
    open(os.path.join(d, 'pdb_wc_long.lut'), 'w').write('-1.5|6\n0|0\n1|-2\n')
    open(os.path.join(d, 'pdb_wc_steer.lut'), 'w').write('-1|1.5\n0|0\n1|1.5\n')
    p = os.path.join(d, 'aero.ini')
    raw = open(p, newline='').read()
    eol = '\r\n' if '\r\n' in raw else '\n'
    extra = []
    for i, (wing, inp, comb, lutf, filt, up, dn) in enumerate(((2, 'SUS_TRAVEL_LR', 'ADD', 'pdb_wc_travl.lut', 0.9, 14, -3), (2, 'SUS_TRAVEL_RR', 'MULT', 'pdb_wc_travr.lut', 0.7, 16, -4),
                                                                 (1, 'LONG', 'ADD', 'pdb_wc_long.lut', 0.85, 8, -4), (0, 'STEER', 'ADD', 'pdb_wc_steer.lut', 0.6, 4, -1))):
        extra += ['', '[DYNAMIC_CONTROLLER_%d]' % i, 'WING=%d' % wing, 'COMBINATOR=%s' % comb, 'INPUT=%s' % inp, 'LUT=%s' % lutf, 'FILTER=%g' % filt, 'UP_LIMIT=%g' % up, 'DOWN_LIMIT=%g' % dn]
    open(p, 'w', newline='').write(raw.rstrip('\r\n') + eol + eol.join(extra) + eol)


def make_dynctrl_car(base, src='ks_toyota_supra_mkiv_drift', dst='pdb_dynctrl_supra'):
    """DynamicController files (Car/DynamicController.cpp) -- chains of filtered LUT stages over car signals: ctrl_wastegate0.ini (the first turbo's
    wastegate from rpm, scaled by gear; Engine.cpp:124-143,368-376), ctrl_turbo1.ini (the second turbo's maxBoost: a constant scaled by the throttle),
    ctrl_single_lock.ini (the differential's preload from speed and brake, the power ramp then 0; Drivetrain.cpp:144-151,603-607).  No shipped
    car has any of them; the Supra with all three pins the evaluator and its three consumers."""
    s = os.path.join(REF, 'content', 'cars', src, 'data'); d = os.path.join(base, 'content', 'cars', dst, 'data')
    if os.path.isdir(d):
        shutil.rmtree(d)
    shutil.copytree(s, d)
    os.system('chmod -R u+w "%s"' % d)
    def ctrl(name, stages):
        out = []
        for i, st in enumerate(stages):
            out += ['[CONTROLLER_%d]' % i] + ['%s=%s' % kv for kv in st] + ['']
        open(os.path.join(d, name), 'w', newline='').write('\n'.join(out))
    ctrl('ctrl_wastegate0.ini', [[('INPUT', 'RPMS'), ('COMBINATOR', 'ADD'), ('LUT', '(|0=0.6|2500=0.7|4500=1.0|6500=1.15|8000=0.9|)'), ('FILTER', '0.95'), ('UP_LIMIT', '1.1'), ('DOWN_LIMIT', '0.5')],
                                 [('INPUT', 'GEAR'), ('COMBINATOR', 'MULT'), ('LUT', '(|0=0.8|1=0.85|2=0.95|3=1.0|5=1.05|)'), ('FILTER', '0.5'), ('UP_LIMIT', '0'), ('DOWN_LIMIT', '0')]])
    ctrl('ctrl_turbo1.ini', [[('INPUT', 'CONST'), ('COMBINATOR', 'ADD'), ('CONST_VALUE', '1.25'), ('FILTER', '0.99'), ('UP_LIMIT', '0'), ('DOWN_LIMIT', '0')],
                             [('INPUT', 'GAS'), ('COMBINATOR', 'MULT'), ('LUT', '(|0=0.7|0.5=0.9|1=1.0|)'), ('FILTER', '0.9'), ('UP_LIMIT', '1.3'), ('DOWN_LIMIT', '0.6')]])
    lock = None
    ctrl('ctrl_single_lock.ini', [[('INPUT', 'CONST'), ('COMBINATOR', 'ADD'), ('CONST_VALUE', '25'), ('FILTER', '0'), ('UP_LIMIT', '0'), ('DOWN_LIMIT', '0')],
                                  [('INPUT', 'SPEED_KMH'), ('COMBINATOR', 'MULT'), ('LUT', '(|0=0.4|30=1.0|90=2.2|160=3.0|)'), ('FILTER', '0.8'), ('UP_LIMIT', '0'), ('DOWN_LIMIT', '0')],
                                  [('INPUT', 'BRAKE'), ('COMBINATOR', 'ADD'), ('LUT', '(|0=0|1=40|)'), ('FILTER', '0.9'), ('UP_LIMIT', '90'), ('DOWN_LIMIT', '8')]])


def make_brakectrl_car(base, src='ks_mazda_rx7_tuned', dst='pdb_brakectrl_rx7'):
    """The brake system's two DynamicController files: ctrl_ebb.ini (the front bias of the tick: a constant shifted by the longitudinal g, limits;
    BrakeSystem.cpp:64-69,90-93) and steer_brake_controller.ini (extra brake torque on the inner rear wheel from the steering input, faded in
    with speed; BrakeSystem.cpp:33-38,136-143) -- and the anti-roll bars' ctrl_arb_front.ini / ctrl_arb_rear.ini.  The tuned RX-7 with all four, through the brake script."""
    s = os.path.join(REF, 'content', 'cars', src, 'data'); d = os.path.join(base, 'content', 'cars', dst, 'data')
    if os.path.isdir(d):
        shutil.rmtree(d)
    shutil.copytree(s, d)
    os.system('chmod -R u+w "%s"' % d)
    open(os.path.join(d, 'ctrl_ebb.ini'), 'w', newline='').write('\n'.join([
        '[CONTROLLER_0]', 'INPUT=CONST', 'COMBINATOR=ADD', 'CONST_VALUE=0.62', 'FILTER=0', 'UP_LIMIT=0', 'DOWN_LIMIT=0', '',
        '[CONTROLLER_1]', 'INPUT=LONG', 'COMBINATOR=ADD', 'LUT=(|-1.5=0.14|0=0|1=-0.06|)', 'FILTER=0.85', 'UP_LIMIT=0.8', 'DOWN_LIMIT=0.5', '']))
    open(os.path.join(d, 'steer_brake_controller.ini'), 'w', newline='').write('\n'.join([
        '[CONTROLLER_0]', 'INPUT=STEER', 'COMBINATOR=ADD', 'LUT=(|-1=-260|-0.1=0|0.1=0|1=260|)', 'FILTER=0.7', 'UP_LIMIT=0', 'DOWN_LIMIT=0', '',
        '[CONTROLLER_1]', 'INPUT=SPEED_KMH', 'COMBINATOR=MULT', 'LUT=(|0=0|25=1|300=1|)', 'FILTER=0.5', 'UP_LIMIT=200', 'DOWN_LIMIT=-200', '']))
    # the anti-roll bars' rates from controller files (Car.cpp:158-167, AntirollBar.cpp:19-22; the bars step after the drivetrain: the rear one reads this tick's rpm and gear)
    open(os.path.join(d, 'ctrl_arb_front.ini'), 'w', newline='').write('\n'.join([
        '[CONTROLLER_0]', 'INPUT=SPEED_KMH', 'COMBINATOR=ADD', 'LUT=(|0=18000|60=30000|160=52000|)', 'FILTER=0.9', 'UP_LIMIT=0', 'DOWN_LIMIT=0', '',
        '[CONTROLLER_1]', 'INPUT=LATG', 'COMBINATOR=MULT', 'LUT=(|-1.5=1.3|0=1|1.5=1.3|)', 'FILTER=0.6', 'UP_LIMIT=60000', 'DOWN_LIMIT=10000', '']))
    open(os.path.join(d, 'ctrl_arb_rear.ini'), 'w', newline='').write('\n'.join([
        '[CONTROLLER_0]', 'INPUT=RPMS', 'COMBINATOR=ADD', 'LUT=(|0=9000|3000=14000|7000=26000|)', 'FILTER=0.8', 'UP_LIMIT=0', 'DOWN_LIMIT=0', '',
        '[CONTROLLER_1]', 'INPUT=GEAR', 'COMBINATOR=MULT', 'LUT=(|0=1.2|1=1.1|3=1.0|5=0.9|)', 'FILTER=0.3', 'UP_LIMIT=0', 'DOWN_LIMIT=0', '']))


def make_braketemp_car(base, src='ks_mazda_rx7_tuned', dst='pdb_braketemp_rx7'):
    """Brake disc temperatures (brakes.ini [TEMPS_FRONT] + [TEMPS_REAR], BrakeSystem.cpp:40-52,151-169: the reference names the F40, which it does not
    ship): each disc heats with the work done on it, cools with the air stream and scales its torque through PERF_CURVE; Car::reset puts the discs at
    the ambient temperature.  The tuned RX-7 with both sections, through the brake script with a reset in mid-run."""
    s = os.path.join(REF, 'content', 'cars', src, 'data'); d = os.path.join(base, 'content', 'cars', dst, 'data')
    if os.path.isdir(d):
        shutil.rmtree(d)
    shutil.copytree(s, d)
    os.system('chmod -R u+w "%s"' % d)
    p = os.path.join(d, 'brakes.ini')
    raw = open(p, newline='').read()
    eol = '\r\n' if '\r\n' in raw else '\n'
    extra = ['', '[TEMPS_FRONT]', 'COOL_TRANSFER=0.004', 'COOL_SPEED_FACTOR=0.006', 'TORQUE_K=0.9', 'PERF_CURVE=(|0=0.7|60=0.85|200=1.0|500=1.0|800=0.55|)', '',
             '[TEMPS_REAR]', 'COOL_TRANSFER=0.003', 'COOL_SPEED_FACTOR=0.004', 'TORQUE_K=0.7', 'PERF_CURVE=(|0=0.8|100=1.0|450=1.0|700=0.6|)', '']
    open(p, 'w', newline='').write(raw.rstrip('\r\n') + eol + eol.join(extra))


def make_ctrl_inputs_cars(base, src='ks_toyota_ae86_drift'):
    """The controller inputs that read the tyres' status (DynamicController.cpp:191-253), on two derived AE86s (eight stages per car): A -- the front
    anti-roll bar's rate from the driven axle's slip ratios and the axles' mean slip angles (stepped after the tyres and the drivetrain of the tick), the
    EBB's front bias from the front load spread and the steering angles (stepped before them: last tick's status); B -- the rear bar's rate from the
    largest slip angles, the oversteer factor and the rear / front wheel-speed ratio."""
    def car(dst, files):
        s = os.path.join(REF, 'content', 'cars', src, 'data'); d = os.path.join(base, 'content', 'cars', dst, 'data')
        if os.path.isdir(d):
            shutil.rmtree(d)
        shutil.copytree(s, d)
        os.system('chmod -R u+w "%s"' % d)
        for name, stages in files.items():
            out = []
            for i, (inp, comb, lutv, filt, up, dn) in enumerate(stages):
                out += ['[CONTROLLER_%d]' % i, 'INPUT=%s' % inp, 'COMBINATOR=%s' % comb, 'LUT=%s' % lutv, 'FILTER=%g' % filt, 'UP_LIMIT=%g' % up, 'DOWN_LIMIT=%g' % dn, '']
            open(os.path.join(d, name), 'w', newline='').write('\n'.join(out))
    car('pdb_ctrlin_a_ae86', {
        'ctrl_arb_front.ini': [('SLIPRATIO_MAX', 'ADD', '(|-1=30000|0=20000|0.3=26000|2=36000|)', 0.7, 0, 0), ('SLIPRATIO_AVG', 'ADD', '(|-1=-4000|0=0|1=5000|)', 0.5, 0, 0),
                               ('SLIPANGLE_FRONT_AVG', 'ADD', '(|-20=3000|0=0|20=3000|)', 0.8, 0, 0), ('SLIPANGLE_REAR_AVG', 'ADD', '(|-30=-5000|0=0|30=-5000|)', 0.6, 45000, 12000)],
        'ctrl_ebb.ini': [('LOAD_SPREAD_LF', 'ADD', '(|0=0.3|0.5=0.6|1=0.3|)', 0.2, 0, 0), ('LOAD_SPREAD_RF', 'ADD', '(|0=0.1|0.5=0.0|1=0.1|)', 0.4, 0, 0),
                         ('STEER_DEG', 'ADD', '(|-500=0.05|0=0|500=0.05|)', 0.9, 0, 0), ('WHEEL_STEER_DEG', 'ADD', '(|-1=-0.04|0=0|1=0.04|)', 0.3, 0.78, 0.5)]})
    car('pdb_ctrlin_b_ae86', {
        'ctrl_arb_rear.ini': [('SLIPANGLE_FRONT_MAX', 'ADD', '(|0=9000|10=12000|40=15000|)', 0.85, 0, 0), ('SLIPANGLE_REAR_MAX', 'ADD', '(|0=0|15=4000|60=9000|)', 0.6, 0, 0),
                              ('OVERSTEER_FACTOR', 'ADD', '(|-30=2500|0=0|30=-3000|)', 0.75, 0, 0), ('REAR_SPEED_RATIO', 'MULT', '(|0=0.8|1=1.0|1.5=1.25|3=1.4|)', 0.5, 30000, 4000)],
        # the suspension-travel inputs (millimetres), read by the brake system before this tick's suspension step
        'ctrl_ebb.ini': [('AVG_TRAVEL_REAR', 'ADD', '(|-40=0.7|0=0.62|60=0.55|)', 0.6, 0, 0), ('SUS_TRAVEL_LR', 'ADD', '(|-40=0.03|0=0|60=-0.03|)', 0.3, 0, 0),
                         ('SUS_TRAVEL_RR', 'ADD', '(|-40=-0.02|0=0|60=0.04|)', 0.8, 0.8, 0.45)]})


def make_dynctrl_ae86(base, src='ks_toyota_ae86_drift', dst='pdb_dynctrl_ae86'):
    """A 33-row car with a controller file: the launcher must route it through the row-guarded kernels (the exact-size ones are compiled without the
    controllers' call sites) -- tests/test_gpu_parity.py steps it against the oracle.  The differential's preload from speed and throttle."""
    s = os.path.join(REF, 'content', 'cars', src, 'data'); d = os.path.join(base, 'content', 'cars', dst, 'data')
    if os.path.isdir(d):
        shutil.rmtree(d)
    shutil.copytree(s, d)
    os.system('chmod -R u+w "%s"' % d)
    open(os.path.join(d, 'ctrl_single_lock.ini'), 'w', newline='').write('\n'.join([
        '[CONTROLLER_0]', 'INPUT=SPEED_KMH', 'COMBINATOR=ADD', 'LUT=(|0=10|40=35|120=80|)', 'FILTER=0.9', 'UP_LIMIT=0', 'DOWN_LIMIT=0', '',
        '[CONTROLLER_1]', 'INPUT=GAS', 'COMBINATOR=MULT', 'LUT=(|0=0.5|1=1.4|)', 'FILTER=0.7', 'UP_LIMIT=100', 'DOWN_LIMIT=5', '']))


def make_twobox_car(base, src='ks_toyota_ae86_drift', dst='pdb_twobox_ae86'):
    """Every shipped car has exactly one box collider; CarColliderManager::init (CarColliderManager.cpp:17-33) takes every COLLIDER_n section there is, each
    a box geom of its own on the chassis.  The AE86 with a second, lower box under its nose (a splitter) pins the loop."""
    s = os.path.join(REF, 'content', 'cars', src, 'data'); d = os.path.join(base, 'content', 'cars', dst, 'data')
    if os.path.isdir(d):
        shutil.rmtree(d)
    shutil.copytree(s, d)
    os.system('chmod -R u+w "%s"' % d)
    p = os.path.join(d, 'colliders.ini')
    raw = open(p, newline='').read()
    eol = '\r\n' if '\r\n' in raw else '\n'
    open(p, 'w', newline='').write(raw.rstrip('\r\n') + eol + eol.join(['', '[COLLIDER_1]', 'CENTRE=0 ,-0.33 ,1.55', 'GROUND_ENABLE=1', 'SIZE=1.30,0.06 ,0.50', '']))


def main():
    base = os.path.join(here, '_ref', 'base')
    os.makedirs(os.path.join(base, 'cfg'), exist_ok=True)
    shutil.copy(os.path.join(REF, 'cfg', 'sim.ini'), os.path.join(base, 'cfg', 'sim.ini'))
    for model in os.listdir(os.path.join(REF, 'content', 'cars')):
        dst = os.path.join(base, 'content', 'cars', model, 'data')
        if os.path.isdir(dst):
            shutil.rmtree(dst)
        shutil.copytree(os.path.join(REF, 'content', 'cars', model, 'data'), dst)
        os.system('chmod -R u+w "%s"' % dst)
    make_multilink_car(base)
    make_heave_car(base)
    make_fwd_car(base)
    make_cold_car(base)
    make_curves_car(base)
    make_ground_effect_car(base)
    make_aerodata_car(base)
    make_wingctrl_car(base)
    make_wingctrl2_car(base)
    make_dynctrl_car(base)
    make_dynctrl_ae86(base)
    make_brakectrl_car(base)
    make_ctrl_inputs_cars(base)
    make_braketemp_car(base)
    make_twobox_car(base)
    make_slip_car(base)
    gen_track.gen_flat(os.path.join(base, 'content', 'tracks', 'flat'))
    gen_track.gen_touge(os.path.join(base, 'content', 'tracks', 'touge'))
    gen_track.gen_walled(os.path.join(base, 'content', 'tracks', 'walled'))
    for trk in ('driftplayground', 'ebisu_touge', 'yamanashi_short', 'euphoria_hillside_park'):   # the shipped tracks that have both surfaces.bin and spline.bin
        dst = os.path.join(base, 'content', 'tracks', trk)
        if os.path.isdir(dst):
            shutil.rmtree(dst)
        shutil.copytree(os.path.join(REF, 'content', 'tracks', trk), dst)
        os.system('chmod -R u+w "%s"' % dst)
    for trk in ('ek_akina', 'ks_nordschleife'):   # shipped with their spline only: the road is a ribbon around it (synthetic_tracks.gen_ribbon)
        dst = os.path.join(base, 'content', 'tracks', trk)
        if os.path.isdir(dst):
            shutil.rmtree(dst)
        gen_track.ribbon_track_from(os.path.join(REF, 'content', 'tracks', trk), dst)
    print(base)
if __name__ == '__main__':
    main()

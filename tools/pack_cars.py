#!/usr/bin/env python3
"""Build container only: pack the reference's shipped cars that the loader supports into the product's own block format
(projectd-core_amd/data/<model>.env.pdcar = configured like pyprojectd/projectd_env.py, .default.pdcar = as loaded).
The blocks are what travels to machines without the reference's content/ directory."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
import pdb_ctypes as pc
REF = '/root/reference'
lib = pc.load_product(host_only=True)
out = os.path.join(ROOT, 'projectd-core_amd', 'data')
os.makedirs(out, exist_ok=True)
DERIVED = os.path.join(ROOT, 'oracle', '_ref', 'base')   # cars derived for the fixtures (oracle/make_base.py): the multilink Supra, the RX-7 with heave springs
models = [(REF, m) for m in sorted(os.listdir(os.path.join(REF, 'content', 'cars')))] + [(DERIVED, 'pdb_ml_supra'), (DERIVED, 'pdb_heave_rx7'), (DERIVED, 'pdb_fwd_ae86'), (DERIVED, 'pdb_cold_rx7'), (DERIVED, 'pdb_curves_ae86'), (DERIVED, 'pdb_gh_fc3s'), (DERIVED, 'pdb_aerodata_ae86'), (DERIVED, 'pdb_wingctrl_fc3s'), (DERIVED, 'pdb_dynctrl_supra'), (DERIVED, 'pdb_dynctrl_ae86'), (DERIVED, 'pdb_brakectrl_rx7'), (DERIVED, 'pdb_ctrlin_a_ae86'), (DERIVED, 'pdb_ctrlin_b_ae86'), (DERIVED, 'pdb_braketemp_rx7'), (DERIVED, 'pdb_wingctrl2_fc3s'), (DERIVED, 'pdb_twobox_ae86'), (DERIVED, 'pdb_slip_ae86')]
for REFB, m in models:
    P = pc.CarParams()
    if lib.pdb_build_car_model(REFB.encode(), m.encode(), C.byref(P)) != 0:
        print('%-36s not supported: %s' % (m, lib.pdb_last_error().decode())); continue
    open(os.path.join(out, m + '.default.pdcar'), 'wb').write(bytes(P))
    E = pc.env_params(lib, REFB, m)
    open(os.path.join(out, m + '.env.pdcar'), 'wb').write(bytes(E))
    print('%-36s bodies %d joints %d rows %d turbos %d gears %d' % (m, P.numBodies, P.numJoints, P.numRows, P.numTurbos, P.numGears))

# the scenarios that end their set-up with a list of setCarTune calls (oracle/scenarios.h: tuneSet): the block as it stands after the
# env's own tunes and that list, <scenario>.tuned.pdcar -- what tests/scenario_util.setup starts from where the cars' setup.ini is absent
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_ctypes
orc = oracle_ctypes.load_oracle()
for sid in range(orc.cpuref_num_scenarios()):
    nm = C.c_char_p(); val = C.c_float()
    if orc.cpuref_scenario_tune(sid, 0, C.byref(nm), C.byref(val)):
        name = orc.cpuref_scenario_name(sid).decode(); model = orc.cpuref_scenario_car(sid).decode()
        P = pc.env_params(lib, REF, model)
        i = 0
        while orc.cpuref_scenario_tune(sid, i, C.byref(nm), C.byref(val)):
            lib.pdb_set_car_tune(C.byref(P), REF.encode(), model.encode(), nm.value, val.value, 0)
            i += 1
        open(os.path.join(out, name + '.tuned.pdcar'), 'wb').write(bytes(P))
        print('%-36s %s + %d tunes' % (name + '.tuned', model, i))

"""ORACLE / TEST INFRASTRUCTURE: ctypes loader for oracle/liboracle*.so (the CPU restatement).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product package never does."""
import ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path):
    if not os.path.exists(path):
        raise RuntimeError('%s not built (run: python -c "import __graft_entry__ as g; g.build()")' % path)
    return C.CDLL(path)


def load_oracle(portable_math=False):
    """portable_math=False: glibc build, pinned against tests/golden (reference-TU trajectories);
    portable_math=True: same restatement with the product's reproducible elementary functions (bit-comparable with the GPU)."""
    lib = _load(os.path.join(ROOT, 'oracle', 'liboracle_pm.so' if portable_math else 'liboracle.so'))
    lib.cpuref_create.restype = C.c_void_p
    lib.cpuref_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.cpuref_destroy.argtypes = [C.c_void_p]
    lib.cpuref_set_state.argtypes = [C.c_void_p, C.c_void_p]
    lib.cpuref_get_state.argtypes = [C.c_void_p, C.c_void_p]
    lib.cpuref_step.argtypes = [C.c_void_p, C.c_float, C.c_float]
    lib.cpuref_step_env.argtypes = [C.c_void_p, C.c_float, C.c_float]
    lib.cpuref_step_controls.argtypes = [C.c_void_p, C.c_void_p]
    lib.cpuref_scenario_controls.argtypes = [C.c_int, C.c_int, C.c_void_p]
    lib.cpuref_scenario_info.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.cpuref_solver_freeflight.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.cpuref_last_system.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.cpuref_set_auto_teleport_hook.argtypes = [C.c_void_p, C.c_void_p]
    lib.cpuref_joint_unit.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]
    lib.cpuref_get_contacts.argtypes = [C.c_void_p, C.c_void_p]
    lib.cpuref_set_contacts.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.cpuref_last_contact_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.cpuref_contact_unit.argtypes = [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    lib.cpuref_get_out.argtypes = [C.c_void_p, C.c_void_p]
    lib.cpuref_get_car_state.argtypes = [C.c_void_p, C.c_void_p]
    lib.cpuref_env_gas.restype = C.c_float; lib.cpuref_env_gas.argtypes = [C.c_float]
    lib.cpuref_scenario_name.restype = C.c_char_p
    lib.cpuref_scenario_track.restype = C.c_char_p
    lib.cpuref_scenario_car.restype = C.c_char_p
    lib.cpuref_scenario_fields.argtypes = [C.c_int, C.c_void_p]
    lib.cpuref_scenario_teledist.restype = C.c_float; lib.cpuref_scenario_teledist.argtypes = [C.c_int]
    lib.cpuref_scenario_teleport.argtypes = [C.c_int, C.c_int, C.c_void_p]
    lib.cpuref_run_scenario2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_char_p]
    lib.cpuref_scenario_two_car.argtypes = [C.c_int, C.c_void_p]
    lib.cpuref_scenario_action2.argtypes = [C.c_int, C.c_int, C.c_void_p]; lib.cpuref_scenario_feedback2.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    for f in ('cpuref_get_slip', 'cpuref_set_slip'):
        getattr(lib, f).argtypes = [C.c_void_p, C.c_void_p]
    lib.cpuref_set_other_slips.argtypes = [C.c_void_p, C.c_void_p, C.c_int]; lib.cpuref_set_guid.argtypes = [C.c_void_p, C.c_int]
    lib.cpuref_scenario_action.argtypes = [C.c_int, C.c_int, C.c_void_p]
    lib.cpuref_scenario_feedback.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.cpuref_run_scenario.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
    lib.cpuref_run_scenario_cb.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p]
    lib.cpuref_bench.restype = C.c_double
    lib.cpuref_bench.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    return lib


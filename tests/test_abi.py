"""The drop-in boundary: every entry point include/pdbatch.h declares is exported by the built library, the POD
layouts agree with the python mirrors, and the product fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C, os, re
import pytest
import pdb_ctypes as pc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'pdbatch.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'//[^\n]*', '', src)
    return sorted(set(re.findall(r'\b(pdb_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for s in ('pdb_create', 'pdb_destroy', 'pdb_step', 'pdb_step_host', 'pdb_reset', 'pdb_get_car_state', 'pdb_build_car_model',
              'pdb_build_track', 'pdb_set_car_tune', 'pdb_set_scoring_var', 'pdb_set_assists', 'pdb_teleport_to_spline', 'pdb_last_error'):
        assert s in syms
    assert len(syms) >= 30


def test_library_exports_every_declared_symbol(built):
    lib = C.CDLL(os.path.join(ROOT, 'projectd-core_amd', 'libpdbatch.so'))
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_host_library_exports_the_loader_half(built):
    lib = C.CDLL(os.path.join(ROOT, 'projectd-core_amd', 'libpdbhost.so'))
    for s in ('pdb_last_error', 'pdb_version', 'pdb_build_car_model', 'pdb_set_car_tune', 'pdb_set_scoring_var', 'pdb_get_scoring_var',
              'pdb_set_assists', 'pdb_build_track', 'pdb_free', 'pdb_initial_state', 'pdb_teleport_to_spline'):
        assert hasattr(lib, s), s


def test_pod_sizes_match_the_header(built):
    """sizes are part of the ABI (pdb_types.h static_asserts the same numbers on the C side)"""
    src = open(os.path.join(ROOT, 'include', 'pdb_types.h')).read()
    sizes = dict(re.findall(r'static_assert\(sizeof\((\w+)\)\s*==\s*(\d+)', src))
    assert int(sizes['pdb_car_state']) == C.sizeof(pc.CarState) == 664      # reference CarState (Car/CarState.h), pack 4
    assert int(sizes['pdb_dyn_state']) == C.sizeof(pc.DynState)
    assert int(sizes['pdb_car_params']) == C.sizeof(pc.CarParams)
    assert int(sizes['pdb_step_out']) == C.sizeof(pc.StepOut) == 104
    assert C.sizeof(pc.DynState) % 16 == 0


def test_no_cpu_fallback(built):
    """on a box without a GPU the device half must refuse, with a message -- never compute on the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import pdbatch
    lib = pc.load_product()
    P = pdbatch.packed_params()
    trk = pdbatch.synthetic_track('flat')
    h = lib.pdb_create(0, 4, C.byref(P), trk, len(trk), 1)
    assert not h
    assert b'no usable HIP device' in lib.pdb_last_error()
    with pytest.raises(RuntimeError):
        pdbatch.Batch(4, P, trk)


def test_product_python_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'projectd-core_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.cpp', '.hpp', '.hip', '.inc', '.h')) or f == 'Makefile':
                t = open(os.path.join(dp, f), errors='ignore').read()
                assert 'oracle' not in t.lower() or f == 'pmath.hpp', os.path.join(dp, f)

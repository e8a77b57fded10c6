import os, sys, ctypes as C, tempfile
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))

# PyTorch bundles its own HIP runtime: when both live in one process, torch.cuda has to come up before libpdbatch.so pulls in
# the system's libamdhip64 (the other order leaves torch without a device).  Harmless on a box without a GPU.
try:
    import torch as _torch
    if _torch.cuda.is_available():
        _torch.cuda.init()
except Exception:   # pragma: no cover
    pass

import pdb_ctypes as pc  # noqa: E402
import oracle_ctypes  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def built():
    import __graft_entry__ as g, fcntl, tempfile
    with open(os.path.join(tempfile.gettempdir(), 'pdbatch_build.lock'), 'w') as lock:   # pytest-xdist workers build one at a time
        fcntl.flock(lock, fcntl.LOCK_EX)
        g.build()
    return True


@pytest.fixture(scope='session')
def hostlib(built):
    return pc.load_product(host_only=True)


@pytest.fixture(scope='session')
def oracle(built):
    return oracle_ctypes.load_oracle()


@pytest.fixture(scope='session')
def base_dir(built):
    import synthetic_tracks
    d = tempfile.mkdtemp(prefix='pdb_base_')
    synthetic_tracks.make_base(d, tracks=('flat', 'touge'))
    return d


@pytest.fixture(scope='session')
def env_params():
    """AE86 constant block configured like pyprojectd/projectd_env.py (tunes, assists, scoring vars): packed by the
    product loader in the build container (tests/test_loader.py checks it is reproducible there)."""
    P = pc.CarParams()
    data = open(os.path.join(ROOT, 'projectd-core_amd', 'data', 'ks_toyota_ae86_drift.env.pdcar'), 'rb').read()
    assert len(data) == C.sizeof(P)
    C.memmove(C.byref(P), data, len(data))
    return P


def car_params(model, kind='env'):
    """packed env-configured block of one of the supported cars (projectd-core_amd/data, tools/pack_cars.py);
    kind='tuned': model is a scenario's name, the block carries that scenario's setCarTune list as well"""
    P = pc.CarParams()
    data = open(os.path.join(ROOT, 'projectd-core_amd', 'data', model + '.' + kind + '.pdcar'), 'rb').read()
    assert len(data) == C.sizeof(P)
    C.memmove(C.byref(P), data, len(data))
    return P


@pytest.fixture(scope='session')
def flat_track(hostlib, base_dir):
    return pc.build_track(hostlib, base_dir, 'flat')


@pytest.fixture(scope='session')
def touge_track(hostlib, base_dir):
    return pc.build_track(hostlib, base_dir, 'touge')


@pytest.fixture(scope='session')
def state0(hostlib, env_params, flat_track):
    S = pc.DynState()
    assert hostlib.pdb_initial_state(C.byref(env_params), flat_track, C.byref(S)) == 0, hostlib.pdb_last_error()
    return S


def load_golden(name):
    z = np.load(os.path.join(HERE, 'golden', name + '.npz'))
    names = [str(x) for x in z['names']]
    n = len(z['ticks'])
    data = np.zeros((n, len(names)), dtype=np.float64)
    data[:, z['f32_cols']] = z['f32'].astype(np.float64)
    data[:, z['f64_cols']] = z['f64']
    return dict(names=names, idx={k: i for i, k in enumerate(names)}, ticks=z['ticks'], actions=z['actions'], data=data)

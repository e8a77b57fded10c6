"""The step's reproducible elementary functions (csrc/device/pmath.hpp): the product's host build, the oracle's
independent implementation of the same specification (oracle/cpu_ref/pm_ref.h) and a correctly-rounded numpy float64
evaluation must agree.  These functions are why GPU and CPU trajectories can be compared bit for bit; glibc (what the
reference's build calls) may differ from them by 1 ulp on a small fraction of inputs, which is the entire difference
between the two oracle builds (tests/test_oracle_math_modes.py)."""
import ctypes as C
import numpy as np
import pytest

FN = {'sin': 0, 'cos': 1, 'tan': 2, 'atan': 3, 'atan2': 4, 'asin': 5, 'acos': 6, 'pow': 7}


def samples(fn, n, rng):
    if fn in ('sin', 'cos', 'tan'):
        x = np.concatenate([rng.uniform(-np.pi, np.pi, n // 2), rng.uniform(-200.0, 200.0, n // 4), rng.normal(0, 1e-3, n // 4)])
        return x.astype(np.float32), None
    if fn == 'atan':
        return np.concatenate([rng.uniform(-4, 4, n // 2), rng.normal(0, 100, n // 2)]).astype(np.float32), None
    if fn == 'atan2':
        return rng.normal(0, 10, n).astype(np.float32), rng.normal(0, 10, n).astype(np.float32)
    if fn in ('asin', 'acos'):
        return rng.uniform(-1, 1, n).astype(np.float32), None
    x = np.exp(rng.uniform(np.log(1e-3), np.log(1e4), n)).astype(np.float32)
    return x, rng.uniform(0.05, 3.0, n).astype(np.float32)


def exact(fn, x, y):
    xd = x.astype(np.float64)
    f = {'sin': np.sin, 'cos': np.cos, 'tan': np.tan, 'atan': np.arctan, 'asin': np.arcsin, 'acos': np.arccos}
    if fn == 'atan2':
        return np.arctan2(xd, y.astype(np.float64))
    if fn == 'pow':
        return np.power(xd, y.astype(np.float64))
    return f[fn](xd)


def ulps(a, b):
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia); ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def call(lib, name, fn, x, y, *extra):
    out = np.empty_like(x)
    f = getattr(lib, name)
    yp = y.ctypes.data_as(C.c_void_p) if y is not None else None
    assert f(FN[fn], *extra, x.ctypes.data_as(C.c_void_p), yp, out.ctypes.data_as(C.c_void_p), len(x)) == 0
    return out


@pytest.mark.parametrize('fn', sorted(FN))
def test_product_and_oracle_implementations_agree_bitwise(hostlib, oracle, fn):
    rng = np.random.RandomState(7 + FN[fn])
    x, y = samples(fn, 200000, rng)
    p = call(hostlib, 'pdb_math_eval', fn, x, y)
    o = call(oracle, 'cpuref_math_eval', fn, x, y, 0)
    assert np.array_equal(p.view(np.int32), o.view(np.int32))


@pytest.mark.parametrize('fn', sorted(FN))
def test_correctly_rounded_and_close_to_glibc(hostlib, oracle, fn):
    rng = np.random.RandomState(99 + FN[fn])
    x, y = samples(fn, 200000, rng)
    p = call(hostlib, 'pdb_math_eval', fn, x, y)
    ref = exact(fn, x, y).astype(np.float32)     # float64 libm result rounded once: correctly rounded except near-ties
    fin = np.isfinite(ref)
    d = ulps(p[fin], ref[fin])
    assert d.max() <= 1 and (d > 0).mean() < 1e-4, (fn, int(d.max()), float((d > 0).mean()))
    g = call(oracle, 'cpuref_math_eval', fn, x, y, 1)
    dg = ulps(p[fin], g[fin])
    assert dg.max() <= 2, (fn, int(dg.max()))     # glibc's float functions are within 1 ulp of exact


def test_special_cases(hostlib):
    x = np.array([0.0, -0.0, 1.0, 0.0, 2.0], dtype=np.float32)
    y = np.array([0.0, 0.0, 0.0, 2.0, 0.0], dtype=np.float32)
    p = call(hostlib, 'pdb_math_eval', 'pow', x, y)
    assert list(p) == [1.0, 1.0, 1.0, 0.0, 1.0]
    a = call(hostlib, 'pdb_math_eval', 'atan2', np.array([0, 1, -1, 0], dtype=np.float32), np.array([0, 0, 0, -1], dtype=np.float32))
    assert a[0] == 0 and abs(a[1] - np.pi / 2) < 1e-6 and abs(a[2] + np.pi / 2) < 1e-6 and abs(a[3] - np.pi) < 1e-6
    assert hostlib.pdb_math_eval(9, None, None, None, 0) < 0

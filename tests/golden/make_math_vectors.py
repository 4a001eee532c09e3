"""tests/golden/math_vectors.npz: arguments and the platform libm's (glibc, FMA variant) exp / log / pow of them,
called through ctypes so that nothing but libm computes the values.  The GPU test compares the device's results with
these bit for bit (tests/test_gpu_parity.py::test_device_math_equals_host_libm).  Data only.

    python tests/golden/make_math_vectors.py
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def libm():
    m = ctypes.CDLL('libm.so.6')
    for f, n in (('exp', 1), ('log', 1), ('pow', 2)):
        getattr(m, f).restype = ctypes.c_double
        getattr(m, f).argtypes = [ctypes.c_double] * n
    return m


def arguments(n=4000, seed=99):
    rng = np.random.default_rng(seed)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 0.5, 2.0, 3.0, -3.0, 5e-324, 2.0 ** 1023, 709.78,
                        -745.13, -745.14, 710.0, 1e-300, 2.0 ** -1022, 1e16])
    ex = np.concatenate([special, rng.uniform(-10, 10, n), rng.uniform(-750, 720, n), rng.uniform(-1e-3, 1e-3, n)])
    lg = np.concatenate([special, rng.uniform(0, 100, n), rng.uniform(0.9, 1.1, n), np.exp(rng.uniform(-700, 700, n)),
                         rng.uniform(0, 1, n) * 2.0 ** -1030])
    pa = np.concatenate([np.repeat(special, len(special)), rng.uniform(0, 100, n), rng.uniform(0, 50, n),
                         np.exp(rng.uniform(-50, 50, n)), -rng.uniform(0, 10, n)])
    pb = np.concatenate([np.tile(special, len(special)), rng.uniform(-4, 4, n), 1 - rng.uniform(0, 3, n),
                         rng.uniform(-20, 20, n), rng.integers(-10, 11, n).astype(float)])
    return ex, lg, pa, pb


if __name__ == '__main__':
    m = libm()
    ex, lg, pa, pb = arguments()
    out = dict(exp_x=ex, exp_y=np.array([m.exp(v) for v in ex]), log_x=lg, log_y=np.array([m.log(v) for v in lg]),
               pow_a=pa, pow_b=pb, pow_y=np.array([m.pow(a, b) for a, b in zip(pa, pb)]))
    f = os.path.join(HERE, 'math_vectors.npz')
    np.savez_compressed(f, **out)
    print('wrote', f, os.path.getsize(f))

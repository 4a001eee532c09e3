"""Oracle vs committed golden vectors, and the portable math against the glibc build of the oracle."""
import os

import numpy as np
import pytest

from make_golden_cases import CASES
from oracle_harness import Oracle
from parity import compare

HERE = os.path.dirname(os.path.abspath(__file__))


def check_against_golden(sol, g, exact):
    assert np.array_equal(sol.len, g['len']) and np.array_equal(sol.thlen, g['thlen'])
    assert sol.nevals == int(g['nevals'])
    tol = 0.0 if exact else 5e-12
    for it in g['periods']:
        for ist in range(sol.len.shape[1]):
            n, m = sol.len[it, ist], sol.thlen[it, ist]
            cell, thr = g['cell_%d_%d' % (it, ist)], g['thr_%d_%d' % (it, ist)]
            for k, arr in enumerate((sol.M, sol.C, sol.V)):
                a, b = arr[it, ist, :n], cell[k]
                if exact:
                    assert np.array_equal(a, b, equal_nan=True)
                else:
                    fin = np.isfinite(b)
                    assert np.array_equal(np.isfinite(a), fin)
                    assert np.all(np.abs(a[fin] - b[fin]) <= tol * np.maximum(1, np.abs(b[fin])))
            assert np.array_equal(sol.D[it, ist, :m], thr[0])
            assert np.all(np.abs(sol.TH[it, ist, :m] - thr[1]) <= tol * np.maximum(1, np.abs(thr[1])))


@pytest.mark.parametrize('name', sorted(CASES))
@pytest.mark.parametrize('native', [True, False])
def test_oracle_reproduces_golden(name, native):
    g = np.load(os.path.join(HERE, 'golden', '%s_%s.npz' % (name, 'native' if native else 'portable')))
    orc = Oracle(CASES[name](), native_math=native)
    sol = orc.solve()
    assert sol.rc == 0
    check_against_golden(sol, g, exact=True)
    sims = orc.sim(sol, g['sim_init'], g['sim_rand'])
    assert np.array_equal(sims, g['sims'], equal_nan=True)


@pytest.mark.parametrize('name', ['deaton2', 'retirement2', 'occ3', 'model2'])
def test_portable_math_is_within_libm_noise_of_glibc(name):
    """Same algorithm, two libm's: identical structure, values equal to ~1e-12 (the noise floor any other
    libm -- e.g. the GPU's ocml -- would also show against the reference)."""
    m = CASES[name]()
    a, b = Oracle(m, native_math=True).solve(), Oracle(m, native_math=False).solve()
    ok, rep = compare(b, a, rtol=5e-12, th_tol=5e-12)
    assert ok, rep

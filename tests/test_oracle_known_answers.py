"""Pin the CPU oracle to the reference outputs recorded in SURVEY.md §8(c).

The reference ships no tests or golden vectors and cannot be built in this image
without stand-ins for MATLAB's mex.h and MATLAB-generated modelspec.c, so these recorded values of
the five shipped example models are the known answers the oracle is held to.
"""
import json
import os

import numpy as np
import pytest

from egdst_amd import examples
from oracle_harness import Oracle

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, 'golden', 'survey_known_answers.json')))
RTOL = 1e-13  # recorded values carry 16-17 significant digits


def close(a, b):
    if b == 0:
        return abs(a) <= 1e-300
    return abs(a - b) <= RTOL * max(1.0, abs(b))


@pytest.fixture(scope='module', params=['retirement2', 'deaton2', 'retirement1', 'deaton1', 'occ3'])
def solved(request):
    m = examples.REGISTRY[request.param]()
    sol = Oracle(m, native_math=True).solve()  # glibc arithmetic, as the reference
    assert sol.rc == 0, sol.err
    return request.param, sol


def test_known_answers(solved):
    name, sol = solved
    ka = KA[name]
    assert sol.total_rows() == ka['total_rows']
    for it_s, ref in ka['cells'].items():
        it = int(it_s)
        cm, cd = sol.cell_M(it, 0), sol.cell_D(it, 0)
        if 'rows' in ref:
            assert cm.shape[0] == ref['rows']
        if 'D' in ref:
            assert cd[:, 0].tolist() == ref['D']
            assert all(close(a, b) for a, b in zip(cd[:, 1], ref['TH']))
        if 'TH1' in ref:
            assert close(cd[1, 1], ref['TH1'])
        if 'evf_a0' in ref:
            assert close(cm[0, 3], ref['evf_a0'])
        for k in ('row0', 'row1'):
            if k in ref:
                r = cm[int(k[-1])]
                assert all(close(a, b) for a, b in zip(r, ref[k])), (k, r.tolist(), ref[k])
        for j, col in enumerate('MCAV'):
            if 'row1_' + col in ref:
                assert close(cm[1, j], ref['row1_' + col])
            if 'last_' + col in ref:
                assert close(cm[-1, j], ref['last_' + col])
        if ref.get('V0_neg_inf'):
            assert cm[0, 3] == -np.inf
    if 'colsums' in ka:
        tot = np.zeros(4)
        for it in range(sol.nt):
            cm = sol.cell_M(it, 0)
            tot += np.where(np.isfinite(cm), cm, 0.0).sum(axis=0)
        for j, col in enumerate('MCAV'):
            assert abs(tot[j] - ka['colsums'][col]) <= 2e-15 * abs(ka['colsums'][col]) * sol.total_rows() ** 0.5 + 1e-9
    if ka.get('M_nondecreasing'):
        for it in range(sol.nt):
            assert np.all(np.diff(sol.M[it, 0, :sol.len[it, 0]]) >= 0)


def test_accessor_value_function_reproduces_the_tables():
    """egdst_call.c vf(): at the endogenous grid points of a solved period the value function is the table's V column,
    and below M(a0) it is the analytic u(c)+beta*evf; the bad-index and column-count rules of the gateway hold."""
    import numpy as np
    from egdst_amd import examples
    from oracle_harness import Oracle
    m = examples.retirement2()
    o = Oracle(m, native_math=True)
    sol = o.solve()
    it = 5
    n = sol.len[it, 0]
    Mg, Vg = sol.M[it, 0, 1:n], sol.V[it, 0, 1:n]
    v = o.call(sol, 6, np.column_stack([np.full(n - 1, it + m.t0), np.ones(n - 1), Mg]))
    assert np.allclose(v, Vg, rtol=1e-13, atol=0)
    r = o.call(sol, 1, [[m.t0 + 1, 1, 1, 2.0], [m.t0 + 1, 9, 1, 2.0], [m.t0 + 1, 1, 1, 2.0]])
    assert np.isfinite(r[0]) and np.isnan(r[1]) and np.isnan(r[2])      # a bad ist poisons the rest of the call
    assert np.array_equal(o.call(sol, 3, [[m.t0 + 1, 1, 1]]), [0.0])    # wrong column count: zeros

"""Ad-hoc GPU check: solve example models on the GPU and compare with the oracle (prints a table)."""
import sys
import time

sys.path.insert(0, 'tests')
sys.path.insert(0, '.')
import numpy as np  # noqa: E402
from egdst_amd import examples, runtime  # noqa: E402
from oracle_harness import Oracle  # noqa: E402
from parity import compare  # noqa: E402

CASES = [
    ('deaton1', lambda: examples.deaton1()),
    ('deaton2', lambda: examples.deaton2()),
    ('retirement1', lambda: examples.retirement1()),
    ('retirement2', lambda: examples.retirement2()),
    ('occ3', lambda: examples.occ3()),
    ('C1', lambda: examples.deaton_sig(a0=0, mmax=50, t0=1, T=20, ngridm=100, ny=5)),
    ('model2', lambda: examples.model2(T=20, ngridm=200, nquad=5, sigma=0.2, r=0.02, df=0.95)),
    ('retire8', lambda: examples.retirement8(T=12, ngridm=150, ny=5)),
    ('C2', lambda: examples.retirement2(T=60, ngridm=1000, ngridmax=10000, nthrhmax=1000, ny=10, a0=0)),
    ('C2neg60', lambda: examples.retirement2(T=60, ngridm=1000, ngridmax=10000, nthrhmax=1000, ny=10)),
    ('C2neg40', lambda: examples.retirement2(T=40, ngridm=1000, ngridmax=10000, nthrhmax=1000, ny=10)),
    ('C3', lambda: examples.occ3(ngridm=4000, ngridmax=40000, nthrhmax=4000, ny=15)),
]
only = sys.argv[1:]
for name, mk in CASES:
    if only and name not in only:
        continue
    m = mk()
    ref = Oracle(m).solve()
    m.compile()
    t = time.time()
    sol = m.solve()
    dt = time.time() - t
    ok, rep = compare(sol, ref)
    print('%-12s ok=%s status=%d where=%s rows=%d/%d evals=%d/%d max_rel=%.2e max_dth=%.2e t=%.3fs %s %s' % (
        name, ok, sol.status, sol.where, sol.total_rows(), ref.total_rows(), sol.nevals, ref.nevals, rep['max_rel'],
        rep['max_dth'], dt, rep.get('worst', ''), rep['problems'][:3]), flush=True)
    if sol.status:
        print('   dbg', m._solver.debug(0).tolist(), flush=True)

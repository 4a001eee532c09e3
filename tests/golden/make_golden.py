"""Generate the golden fixtures of tests/golden/ from the CPU oracle.

    python tests/golden/make_golden.py

For every case: the descriptor (run-time scalars, quadrature array, parameters), row/threshold counts of every
cell, the full cells of four periods (last, last-1, middle, 0), column checksums over all periods, the oracle's
evaluation count, and a simulated panel (init, uniforms and the resulting sims).  Two variants per case:
  *_native.npz    oracle built with glibc exp/log/pow -- the arithmetic of the reference; these are the vectors
                  pinned to the reference outputs recorded in SURVEY.md §8(c) (test_oracle_known_answers.py);
  *_portable.npz  oracle built with include/egdst_math.h -- bit-identical to what the GPU path must produce.
Data only: no reference source text is stored here.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from egdst_amd import examples  # noqa: E402
from oracle_harness import Oracle  # noqa: E402

CASES = {
    'deaton1': lambda: examples.deaton1(),
    'deaton2': lambda: examples.deaton2(),
    'retirement1': lambda: examples.retirement1(),
    'retirement2': lambda: examples.retirement2(),
    'occ3': lambda: examples.occ3(),
    'model2': lambda: examples.model2(T=12, ngridm=60, nquad=5, sigma=0.2, r=0.02, df=0.95),
    'retire8': lambda: examples.retirement8(T=6, ngridm=40, ny=3),
    'cakenormal': lambda: examples.cake_normal(),          # additive normal shocks (DISTRIB=2)
    'retiremortal': lambda: examples.retirement_mortal(),  # survival < 1: deaths in the simulated panel
    'retirehc': lambda: examples.retirement_hc(),          # a continuous state
}


def pack(model, native):
    orc = Oracle(model, native_math=native)
    sol = orc.solve()
    assert sol.rc == 0, sol.err
    d = model.descriptor()
    nt, nst = sol.len.shape
    its = sorted({nt - 1, max(nt - 2, 0), nt // 2, 0})
    out = dict(t0=d['t0'], T=d['T'], ngridm=d['ngridm'], ngridmax=d['ngridmax'], nthrhmax=d['nthrhmax'], ny=d['ny'],
               mmax=d['mmax'], a0=d['a0'], quadrature=d['quadrature'], params=model.param_vector(),
               len=sol.len, thlen=sol.thlen, nevals=np.int64(sol.nevals), periods=np.array(its))
    sums = np.zeros((nst, 3))
    for it in range(nt):
        for ist in range(nst):
            n = sol.len[it, ist]
            for k, arr in enumerate((sol.M, sol.C, sol.V)):
                col = arr[it, ist, :n]
                sums[ist, k] += col[np.isfinite(col)].sum()
    out['colsums'] = sums
    for it in its:
        for ist in range(nst):
            n, m = sol.len[it, ist], sol.thlen[it, ist]
            out['cell_%d_%d' % (it, ist)] = np.stack([sol.M[it, ist, :n], sol.C[it, ist, :n], sol.V[it, ist, :n]])
            out['thr_%d_%d' % (it, ist)] = np.stack([sol.D[it, ist, :m], sol.TH[it, ist, :m]])
    # simulation: 6 agents spread over states and cash, own shocks
    rng = np.random.default_rng(12345)
    nsim = 6
    # (with continuous states the reference double-counts the grid index of the initial state, egdst_simulator.c:319:
    # initial states beyond the middle of the grid run off the cell array; the fixtures start below it)
    nst0 = 3 if any(v.type == 'continuous' for v in model.s) else nst
    init = np.stack([1 + (np.arange(nsim) % nst0), np.linspace(max(d['a0'], 0) + 0.25, 0.8 * d['mmax'], nsim)], axis=1)
    rs = rng.random(4 * nt * nsim)
    out['sim_init'], out['sim_rand'] = init, rs
    out['sims'] = orc.sim(sol, init, rs, rndtype=0)
    return out


if __name__ == '__main__':
    for name, mk in CASES.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        for native in (True, False):
            f = os.path.join(HERE, '%s_%s.npz' % (name, 'native' if native else 'portable'))
            np.savez_compressed(f, **pack(mk(), native))
            print('wrote', f, os.path.getsize(f))

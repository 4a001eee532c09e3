"""Validation + timing of the forward simulator: many agents on the full-size C2 solution, bit for bit against the oracle."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')  # run from the repo root
import numpy as np
from egdst_amd import build, runtime, workloads
from oracle_harness import Oracle
nsim = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
m, gen = workloads.c2(a0=0)
lib = build.build_model(m)
P = gen(4)
for rndtype in (0, 1):
    for draw in (0, 2):
        s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=True)
        s.set_params(P[draw:draw + 1]); s.solve()
        nt = s.nt
        rng = np.random.default_rng(11 + draw)
        init = np.column_stack([np.ones(nsim), rng.uniform(0.0, 9.5, nsim)])
        rs = rng.random(4 * nt * (nsim if rndtype == 0 else 1))
        s.simulate(init, rs, rndtype)
        t = time.perf_counter(); sims = s.simulate(init, rs, rndtype); dt = time.perf_counter() - t
        orc = Oracle(m)
        ref = orc.solve(P[draw])
        t = time.perf_counter(); rsim = orc.sim(ref, init, rs, rndtype, params=P[draw]); dc = time.perf_counter() - t
        same = np.array_equal(np.nan_to_num(sims, nan=-777.0), np.nan_to_num(rsim, nan=-777.0))
        print('draw %d rndtype %d: %d agents x %d periods, GPU %.1f ms (incl. copies), oracle %.2f s, bit-identical %s, alive at T: %d' % (
            draw, rndtype, nsim, nt, dt * 1e3, dc, same, int((~np.isnan(sims[:, -1, 0])).sum())), flush=True)
        s.close()

"""Timing of the model-function accessor (egdst_call, value function of the solved C2 tables) against the oracle."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')  # run from the repo root
import numpy as np
from egdst_amd import build, runtime, workloads
from oracle_harness import Oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m, gen = workloads.c2(a0=0)
lib = build.build_model(m)
P = gen(1)
s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=True)
s.set_params(P); s.solve()
rng = np.random.default_rng(5)
args = np.column_stack([rng.integers(m.t0, m.t0 + s.nt, n).astype(float), np.ones(n), rng.uniform(m.a0, m.mmax, n)])
s.call(6, args[:1000])
t = time.perf_counter(); v = s.call(6, args); dt = time.perf_counter() - t
orc = Oracle(m); ref = orc.solve(P[0])
t = time.perf_counter(); r = orc.call(ref, 6, args, params=P[0]); dc = time.perf_counter() - t
print('value function, %d lookups: GPU %.1f ms incl. copies (%.0f M/s), oracle %.2f s (%.1f M/s), bit-identical %s' % (
    n, dt * 1e3, n / dt / 1e6, dc, n / dc / 1e6, np.array_equal(v, r, equal_nan=True)))

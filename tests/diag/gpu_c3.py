"""Diagnostic: kernel time breakdown of a single C3 solve (occ3, T=40, n=4000, ny=15)."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')  # run from the repo root
import numpy as np
from egdst_amd import build, runtime, workloads
m, _ = workloads.c3()
lib = build.build_model(m)
s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=True)
s.set_params(m.param_vector()[None]); s.solve(raise_on_error=False); s.set_profile(True)
t = time.perf_counter(); s.solve(raise_on_error=False); dt = (time.perf_counter() - t) * 1e3
print('C3 single solve %.1f ms' % dt, 'probe/grid/env ms', np.round(s.profile()[0], 1).tolist(), 'status', s.status()[0][0], 'evals', s.evals()[0])
sol = s.solution(0)
print('rows per period (first 5 from T):', [int(x) for x in sol.len[::-1, 0][:5]], 'thresholds', [int(x) for x in sol.thlen[::-1, 0][:5]])

"""Diagnostic: kernel time breakdown of a single C5 solve at full size (8 states, T=100, n=32768, ny=15)."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')  # run from the repo root
import numpy as np
from egdst_amd import build, runtime, workloads
m, gen = workloads.c5()
lib = build.build_model(m)
P = gen(1)
s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=False)
s.set_params(P); s.solve(raise_on_error=False); s.set_profile(True)
t = time.perf_counter(); s.solve(raise_on_error=False); dt = (time.perf_counter() - t) * 1e3
print('C5 single solve %.1f ms' % dt, 'probe/grid(+fixup)/env ms', np.round(s.profile()[0], 1).tolist(), 'status', s.status()[0][0], 'work', int(s.work()[0]))

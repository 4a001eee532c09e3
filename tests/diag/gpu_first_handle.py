"""Diagnostic: batch time vs (first handle of the process or not) x (HIP events recorded or not) x draw groups."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
m, gen = workloads.c2()
lib = build.build_model(m)
nd = 4096
P = gen(nd)
pre = int(sys.argv[1])
if pre:
    s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False); s.close()
for groups in [int(a) for a in sys.argv[2:]]:
    for prof in (0, 1):
        s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
        s.set_groups(groups)
        s.set_params(P)
        s.set_profile(bool(prof))
        ts = []
        for _ in range(3):
            t = time.perf_counter(); s.solve(raise_on_error=False); ts.append((time.perf_counter() - t) * 1e3)
        print('precreate', pre, 'groups', groups, 'profile', prof, ['%.0f' % t for t in ts], flush=True)
        s.close()

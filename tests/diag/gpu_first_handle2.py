"""Diagnostic: what cures the slower first handle of a process?"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
m, gen = workloads.c2(a0=0)
lib = build.build_model(m)
nd = 4096
P = gen(nd)
mode = sys.argv[1]
if mode in ('A', 'B'):
    d = runtime.Solver(lib, m.descriptor(), ndraw=4, keep_history=False)
    d.set_groups(4)
    if mode == 'B':
        d.set_params(P[:4]); d.solve(raise_on_error=False)
    d.close()
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
s.set_params(P)
ts = []
for _ in range(3):
    t = time.perf_counter(); s.solve(raise_on_error=False); ts.append((time.perf_counter() - t) * 1e3)
print('mode', mode, ['%.0f' % t for t in ts], flush=True)

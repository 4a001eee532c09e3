"""Diagnostic: the straggler measure (re-basing calls) of every draw of a batch, and the single-solve time of the top ones."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
m, gen = workloads.c2(a0=0)
lib = build.build_model(m)
nd = int(sys.argv[1])
P = gen(nd)
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
s.set_params(P); s.solve(raise_on_error=False)
w = s.work(); st = s.status()[0]
top = np.argsort(-w.astype(np.int64))[:16]
s.close()
s1 = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=False)
s1.set_profile(True)
for i in top:
    if w[i] == 0: break
    s1.set_params(P[i:i+1]); s1.solve(raise_on_error=False)
    t = time.perf_counter(); s1.solve(raise_on_error=False); dt = (time.perf_counter() - t) * 1e3
    print('draw', int(i), 'work', int(w[i]), 'status', int(st[i]), 'where', s1.status()[1][0].tolist(), 'single solve %.1f ms' % dt, 'probe/grid/env', np.round(s1.profile()[0], 1).tolist(), flush=True)

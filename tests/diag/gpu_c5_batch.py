import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
m, gen = workloads.c5()
lib = build.build_model(m)
nd = int(sys.argv[1])
P = gen(nd)
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
s.set_params(P)
t = time.perf_counter(); s.solve(raise_on_error=False); dt = time.perf_counter() - t
st = s.status()[0]; ev = s.evals()[0]
print('C5 full size, batch of %d: %.2f s, %.2f G evals/s, failed %d, schedule %s' % (nd, dt, ev / dt / 1e9, int((st != 0).sum()), s.schedule()), flush=True)
t = time.perf_counter(); s.solve(raise_on_error=False); dt = time.perf_counter() - t
print('second solve: %.2f s, %.2f G evals/s' % (dt, s.evals()[0] / dt / 1e9))

"""Diagnostic: single-solve wall time of every draw of a workload (who are the stragglers of a batch?)."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
wl, nd = sys.argv[1], int(sys.argv[2])
m, gen = workloads.WORKLOADS[wl]()
lib = build.build_model(m)
P = gen(nd)
s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=False)
s.set_profile(True)
ts = []
for i in range(nd):
    s.set_params(P[i:i + 1])
    s.solve(raise_on_error=False)
    t = time.perf_counter(); s.solve(raise_on_error=False); dt = (time.perf_counter() - t) * 1e3
    kms, kl, ab = s.profile()
    st, wh = s.status()
    ts.append(dt)
    print(i, np.round(P[i], 4).tolist(), 'status', st[0], '%.1f ms' % dt, 'probe/grid/env ms', np.round(kms, 1).tolist(), flush=True)
ts = np.array(ts)
print('min %.1f median %.1f mean %.1f p90 %.1f max %.1f' % (ts.min(), np.median(ts), ts.mean(), np.percentile(ts, 90), ts.max()))

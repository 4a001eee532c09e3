"""Diagnostic: wall time of one solve (ndraw = 1, best of 3) of the BASELINE workloads; EGDST_HIPCC_EXTRA picks a build variant."""
import sys, time, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
for wl in sys.argv[1:] or ('C2', 'C3', 'C5'):
    m = workloads.WORKLOADS[wl]()[0]
    lib = build.build_model(m)
    s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=False)
    s.set_params(m.param_vector()[None]); s.solve()
    ts = []
    for _ in range(3):
        t = time.perf_counter(); s.solve(); ts.append(time.perf_counter() - t)
    print(os.environ.get('EGDST_HIPCC_EXTRA', '(default)'), wl, 'solve %.2f ms' % (min(ts) * 1e3), flush=True)
    del s

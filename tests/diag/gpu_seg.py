"""Diagnostic: segmented vs one-wave envelope walks: single-solve time and merge/fallback counts."""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
for wl in ('C2', 'C3'):
    m = workloads.WORKLOADS[wl]()[0]
    lib = build.build_model(m)
    for noseg in ('1', '0'):
        os.environ['EGDST_NOSEG'] = noseg
        s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=False)
        s.set_profile(True)
        s.set_params(m.param_vector()[None]); s.solve()
        t = time.perf_counter()
        for _ in range(3): s.solve()
        dt = (time.perf_counter() - t) / 3 * 1e3
        print(wl, 'noseg', noseg, '%.1f ms' % dt, 'kernels', np.round(s.profile()[0], 1).tolist(), 'walks merged/fallback', s.walk_stats()[0].tolist(), flush=True)
        s.close()

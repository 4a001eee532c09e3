"""Diagnostic (-DEGDST_STAMPS): inside segmented walks, sum of the segments' times against the longest one."""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
for wl, tag in (('C2', '_stamps'), ('C3', '_stamps_c3')):
    m = workloads.WORKLOADS[wl]()[0]
    lib = build.build_model(m, build_dir='egdst_amd/_models/' + tag, extra_flags=['-DEGDST_STAMPS'])
    for noseg in ('1', '0'):
        os.environ['EGDST_NOSEG'] = noseg
        s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=False)
        s.set_params(m.param_vector()[None]); s.solve()
        b0 = s.debug(0).view(np.uint64).astype(np.float64)
        t = time.perf_counter(); s.solve(); dt = (time.perf_counter() - t) * 1e3
        b = s.debug(0).view(np.uint64).astype(np.float64)
        d = b - b0
        print(wl, 'noseg', noseg, 'solve %.1f ms | stop+compact %.1f sort %.1f walk phase %.1f ms | segments: sum %.1f ms, longest single segment ever %.3f ms | merged/fallback %s' % (
            dt, d[2] * 1e-5, d[5] * 1e-5, d[6] * 1e-5, d[1] * 1e-5, b[0] * 1e-5, s.walk_stats()[0].tolist()), flush=True)
        s.close()

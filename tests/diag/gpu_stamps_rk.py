"""Diagnostic: sort-phase time of one C2 / C3 solve for ENV_RK = 1, 2, 4 (points a thread ranks in lockstep)."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
for wl in ('C2', 'C3'):
    m = workloads.WORKLOADS[wl]()[0]
    for rk in (1, 2, 4):
        lib = build.build_model(m, build_dir='egdst_amd/_models/_stamps_rk%d_%s' % (rk, wl), extra_flags=['-DEGDST_STAMPS', '-DENV_RK=%d' % rk])
        s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=False)
        s.set_params(m.param_vector()[None]); s.solve()
        b0 = s.debug(0).view(np.uint64).copy()
        s.solve()
        d = (s.debug(0).view(np.uint64) - b0).astype(np.float64) * 1e-5
        print(wl, 'RK', rk, 'sort %.2f walk %.2f stop+compact %.2f ms' % (d[5], d[6], d[2]), flush=True)

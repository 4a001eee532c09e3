"""Diagnostic (-DEGDST_STAMPS): inside the wave walk -- generic steps (events) against regular batches, C2, one solve."""
import os, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
m = workloads.WORKLOADS['C2']()[0]
lib = build.build_model(m, build_dir='egdst_amd/_models/_stamps', extra_flags=['-DEGDST_STAMPS'])
for noseg in ('1', '0'):
    os.environ['EGDST_NOSEG'] = noseg
    s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=False)
    s.set_params(m.param_vector()[None]); s.solve()
    b0 = s.debug(0).view(np.uint64).copy()
    s.solve()
    d = s.debug(0).view(np.uint64) - b0
    nb, ns = int(d[3]) >> 32, int(d[3]) & 0xffffffff
    print('noseg', noseg, 'generic steps %d: %.2f ms (%.2f us each) | batches %d: %.2f ms (%.2f us each) | walk phase %.2f ms' % (
        ns, d[4] * 1e-5, d[4] * 1e-2 / max(ns, 1), nb, d[7] * 1e-5, d[7] * 1e-2 / max(nb, 1), d[6] * 1e-5), flush=True)
    s.close()

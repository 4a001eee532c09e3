"""Diagnostic (-DEGDST_STAMPS -DEGDST_STAMPS3): inside the sort phase of k_envelope (LDS path)."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
m = workloads.WORKLOADS['C2']()[0]
extra = sys.argv[1:]
lib = build.build_model(m, build_dir='egdst_amd/_models/_stamps3' + ''.join(extra).replace('-D', '_'), extra_flags=['-DEGDST_STAMPS', '-DEGDST_STAMPS3'] + extra)
s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=False)
s.set_params(m.param_vector()[None]); s.solve()
b0 = s.debug(0).view(np.uint64).copy()
s.solve()
d = (s.debug(0).view(np.uint64) - b0).astype(np.float64) * 1e-5
print(extra, 'C2 sort phase %.2f ms = [folds, pieces: %.2f] + staging/order check %.2f + ranks %.2f + scatter %.2f | walk %.2f stop+compact %.2f' % (
    d[5], d[5] - d[3] - d[4] - d[7], d[3], d[4], d[7], d[6], d[2]), flush=True)

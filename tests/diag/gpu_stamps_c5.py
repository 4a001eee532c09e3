"""Diagnostic (-DEGDST_STAMPS): phases of k_envelope in one C5 solve at full size (global-memory streams)."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
m = workloads.WORKLOADS['C5']()[0]
lib = build.build_model(m, build_dir='egdst_amd/_models/_stamps_c5', extra_flags=['-DEGDST_STAMPS'])
s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=False)
s.set_params(m.param_vector()[None]); s.solve()
b0 = s.debug(0).view(np.uint64).copy()
import time
t = time.perf_counter(); s.solve(); dt = time.perf_counter() - t
d = (s.debug(0).view(np.uint64) - b0).astype(np.float64) * 1e-5
print('C5 one solve %.0f ms | summed over the 8 cells of a period and 100 periods: stop+compact %.0f ms, sort (LDS path only) %.0f ms, walk phase incl. global sort %.0f ms, of which segments (sum over waves) %.0f ms | walks %s' % (
    dt * 1e3, d[2], d[5], d[6], d[1], s.walk_stats()[0].tolist()), flush=True)

"""Diagnostic: section timing inside k_envelope for a single C3 solve (-DEGDST_STAMPS build)."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
m = workloads.c3()[0]
lib = build.build_model(m, build_dir='egdst_amd/_models/_stamps_c3', extra_flags=['-DEGDST_STAMPS'])
s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=False)
s.set_profile(True)
s.set_params(m.param_vector()[None]); s.solve()
b0 = s.debug(0).view(np.uint64).astype(np.float64)
t = time.perf_counter(); s.solve(); dt = time.perf_counter() - t
buf = s.debug(0).view(np.uint64).astype(np.float64) - b0
nb, ns = int(s.debug(0).view(np.uint64)[3] - np.uint64(b0[3])) >> 32, int(s.debug(0).view(np.uint64)[3] - np.uint64(b0[3])) & 0xffffffff
print('C3 single solve %.1f ms (stamps build); kernels probe/grid/env ms %s' % (dt * 1e3, np.round(s.profile()[0], 1).tolist()))
print('inside k_envelope: classification pass %.1f ms | stop+compact %.1f ms, sort %.1f ms, walk %.1f ms | walks: %d batches %.1f ms (%.2f us each), %d generic steps %.1f ms (%.2f us each)' % (
    buf[1] * 1e-5, buf[2] * 1e-5, buf[5] * 1e-5, buf[6] * 1e-5, nb, buf[7] * 1e-5, buf[7] * 1e-2 / max(nb, 1), ns, buf[4] * 1e-5, buf[4] * 1e-2 / max(ns, 1)))

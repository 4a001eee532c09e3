"""Diagnostic: a few single solves in a row (run under rocprofv3 --kernel-trace, then tools/trace_chain.py).
   python tests/diag/gpu_single_trace.py WL [nsolves]"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from egdst_amd import build, runtime, workloads
wl = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
m, _ = workloads.WORKLOADS[wl]()
lib = build.build_model(m)
s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=False)
s.set_params(m.param_vector()[None])
for i in range(n):
    t = time.perf_counter(); s.solve(); print('%s solve %d: %.2f ms' % (wl, i, (time.perf_counter() - t) * 1e3), flush=True)
s.close()

"""Diagnostic: a few batched solves in a row (run under rocprofv3 --kernel-trace, then tools/trace_queues.py).
   python tests/diag/gpu_batch_trace.py WL NDRAW [nsolves]"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from egdst_amd import build, runtime, workloads
wl, nd = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
m, gen = workloads.WORKLOADS[wl]()
flags = workloads.BATCH_BUILD_FLAGS.get(wl, []) if nd >= workloads.BATCH_BUILD_MIN_DRAWS.get(wl, 1 << 30) else []
lib = build.build_model(m, extra_flags=flags)
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
s.set_params(gen(nd) if gen else m.param_vector()[None].repeat(nd, 0))
for i in range(n):
    t = time.perf_counter(); s.solve(raise_on_error=False); print('%s x %d solve %d: %.2f ms' % (wl, nd, i, (time.perf_counter() - t) * 1e3), flush=True)
s.close()

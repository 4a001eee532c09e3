"""Diagnostic: time a batched solve with a library built into another directory (compile-time variants)."""
import sys, time, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
m, gen = workloads.c2(a0=0)
nd = int(sys.argv[2])
P = gen(nd)
lib = runtime.ModelLibrary(os.path.join(build.MODELS_DIR, sys.argv[1], 'libegdst.so'))
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
s.set_params(P); s.solve(raise_on_error=False)
ts = []
for _ in range(3):
    t = time.perf_counter(); s.solve(raise_on_error=False); ts.append((time.perf_counter() - t) * 1e3)
print(sys.argv[1], nd, ['%.0f' % t for t in ts], 'evals', s.evals()[0], 'failed', int((s.status()[0] != 0).sum()))

"""Diagnostic: how long the HOST needs to enqueue a batched solve (egdst_solve_async returns) against the solve's wall time."""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m, gen = workloads.c2(a0=0)
lib = build.build_model(m, extra_flags=sys.argv[2:] + ['-DEGDST_WITH_GRAPH'])   # (the hipGraph replay is compiled into diagnostic builds only)
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
s.set_params(gen(nd))
s.solve(raise_on_error=False)
for _ in range(3):
    t0 = time.perf_counter(); s.solve_async(); t1 = time.perf_counter(); s.sync(raise_on_error=False); t2 = time.perf_counter()
    print('TP=%s GRAPH=%s groups=%s enqueue %.1f ms, total %.1f ms' % (os.environ.get('EGDST_ENV_TP'), os.environ.get('EGDST_GRAPH'), s.schedule()[0], (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)

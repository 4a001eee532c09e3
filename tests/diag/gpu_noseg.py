import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
m, gen = workloads.c2()
lib = build.build_model(m, extra_flags=workloads.BATCH_BUILD_FLAGS['C2'])
P = gen(4096)
for noseg in ('0', '1'):
    os.environ['EGDST_NOSEG'] = noseg
    s = runtime.Solver(lib, m.descriptor(), ndraw=4096, keep_history=False)
    s.set_params(P); s.solve(raise_on_error=False)
    ts = []
    for k in range(4):
        t = time.perf_counter(); s.solve(raise_on_error=False); ts.append((time.perf_counter() - t) * 1e3)
    ws = s.walk_stats().sum(axis=0)
    print('NOSEG', noseg, ['%.1f' % t for t in ts], 'walks merged / fell back:', ws.tolist(), flush=True)
    s.close()

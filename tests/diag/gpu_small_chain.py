"""Diagnostic: the fixed cost of a solve's kernel chain -- C2 forms with few grid points (little work per kernel), many draws."""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for ngridm in [int(a) for a in sys.argv[2:]] or [50, 200, 1000]:
    m, gen = workloads.c2(a0=0, ngridm=ngridm, T=60)
    lib = build.build_model(m)
    s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
    s.set_params(gen(nd)); s.solve(raise_on_error=False)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); s.solve(raise_on_error=False); ts.append((time.perf_counter() - t) * 1e3)
    print('TP=%s groups=%s ngridm=%d ndraw=%d: %s ms, evals %d, failed %d' % (os.environ.get('EGDST_ENV_TP'), s.schedule()[0], ngridm, nd, ['%.1f' % t for t in ts], s.evals()[0], int((s.status()[0] != 0).sum())), flush=True)
    s.close()

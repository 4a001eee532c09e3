"""Diagnostic: batched-step time vs number of draws and vs the physical row capacity.
    python tests/diag/gpu_ndraw_scaling.py <rows_cap|0> <ndraw>..."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads, examples
cap = int(sys.argv[1])
m, gen = workloads.c2(a0=0)
lib = build.build_model(m)
for nd in [int(a) for a in sys.argv[2:]]:
    P = gen(nd)
    s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False, rows_cap=cap)
    s.set_params(P)
    s.solve(raise_on_error=False)
    s.set_profile(True)
    t = time.perf_counter(); s.solve(raise_on_error=False); dt = time.perf_counter() - t
    st, _ = s.status(); ev = s.evals()[0]
    kms, kl, ab = s.profile()
    print('rows_cap', cap, 'geometry', s.geometry(), 'retries', s.capacity_retries, 'ndraw', nd, '%.1f ms' % (dt * 1e3), '%.2f G evals/s' % (ev / dt / 1e9), 'failed', int((st != 0).sum()),
          'probe/grid/env', np.round(kms, 1).tolist(), flush=True)
    print('   deferrals per draw (first 8):', [int(s.debug(i)[15]) for i in range(min(nd, 8))], 'mean', float(np.mean([s.debug(i)[15] for i in range(min(nd, 64))])))
    s.close()

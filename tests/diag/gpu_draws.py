"""Diagnostic: per-draw status of a batched workload on the GPU vs the oracle."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
from oracle_harness import Oracle
wl, nd = sys.argv[1], int(sys.argv[2])
m, gen = workloads.WORKLOADS[wl]()
lib = build.build_model(m)
P = gen(nd)
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
s.set_params(P); s.solve(raise_on_error=False)
st, wh = s.status(); ev = s.evals()[1]
orc = Oracle(m)
for i in range(nd):
    r = orc.solve(P[i])
    print(i, np.round(P[i], 4).tolist(), 'gpu status', st[i], tuple(wh[i]), 'evals', ev[i], '| oracle rc', r.rc, r.err.replace('\n', ' ')[:60], r.nevals, flush=True)

"""Diagnostic: which draws of a batch fail on the GPU, and what does the oracle say about them."""
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
bad = np.where(st != 0)[0]
print('failed', bad.tolist())
for i in bad[:8]:
    r = orc.solve(P[i])
    print(i, np.round(P[i], 4).tolist(), 'gpu status', st[i], tuple(wh[i]), 'dbg', s.debug(int(i)).tolist()[:8], '| oracle rc', r.rc, r.err.replace('\n', ' ')[:70], flush=True)

"""Diagnostic: (ndraw, keep_history) matrix on the GPU; evals per draw must equal the single-draw reference run."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
m, gen = workloads.WORKLOADS[sys.argv[1]]()
lib = build.build_model(m)
P = gen(16)
ref = []
for i in range(16):
    s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=True)
    s.set_params(P[i:i + 1]); s.solve(raise_on_error=False)
    ref.append((int(s.status()[0][0]), int(s.evals()[0])))
    s.close()
print('single keep=1 :', ref, flush=True)
for nd, kh in ((1, False), (16, True), (16, False), (4, False)):
    if nd == 1:
        out = []
        for i in range(16):
            s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=kh)
            s.set_params(P[i:i + 1]); s.solve(raise_on_error=False)
            out.append((int(s.status()[0][0]), int(s.evals()[0]))); s.close()
    else:
        out = []
        for j in range(0, 16, nd):
            s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=kh)
            s.set_params(P[j:j + nd]); s.solve(raise_on_error=False)
            st, ev = s.status()[0], s.evals()[1]
            out += [(int(a), int(b)) for a, b in zip(st, ev)]; s.close()
    print('ndraw=%d keep=%d:' % (nd, kh), 'MATCH' if out == ref else out, flush=True)

"""Diagnostic: a few single-draw C2 solves (for counter collection)."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from egdst_amd import build, runtime, workloads
m, gen = workloads.c2(a0=0)
lib = build.build_model(m)
P = gen(4)
s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=False)
s.set_params(P[0:1])
for _ in range(3):
    s.solve()
print('evals', s.evals()[0])

"""Validation sweep of batched draws of the big workloads against the oracle (status, message, evaluation counts).
    python tests/diag/gpu_sweep_big.py C5r 48      python tests/diag/gpu_sweep_big.py C4 8"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')  # run from the repo root
import numpy as np
from egdst_amd import build, runtime, workloads
from oracle_harness import Oracle
name, nd = sys.argv[1], int(sys.argv[2])
m, gen = {'C4': workloads.c4, 'C5r': lambda: workloads.c5(ngridm=2000, T=60)}[name]()
lib = build.build_model(m)
P = gen(nd)
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
s.set_params(P); s.solve(raise_on_error=False)
t = time.perf_counter(); s.solve(raise_on_error=False); dt = time.perf_counter() - t
st, wh = s.status(); ev = s.evals()[1]
orc = Oracle(m)
bad = 0
t0 = time.time()
for i in range(nd):
    r = orc.solve(P[i])
    ok = ((st[i] == 0) == (r.rc == 0)) and (st[i] != 0 or ev[i] == r.nevals)
    if st[i] != 0 and r.rc != 0:
        ok = ok and lib.lib.egdst_strerror(int(st[i])).decode().strip() == r.err.strip()
    if not ok:
        bad += 1
        print('MISMATCH draw', i, P[i].round(4).tolist(), 'gpu', st[i], wh[i].tolist(), ev[i], '| oracle', r.rc, r.err.strip()[:50], r.nevals, flush=True)
print('%s: %d draws in %.1f ms on the GPU (%.2f G evals/s), %d failed on both sides, %d mismatches (%.0f s of oracle)' % (
    name, nd, dt * 1e3, ev.sum() / dt / 1e9, int((st != 0).sum()), bad, time.time() - t0), flush=True)

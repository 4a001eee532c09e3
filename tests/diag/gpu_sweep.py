"""Validation sweep: many C2 draws on the GPU against the oracle -- status and evaluation counts for all of them, complete
tables (bit for bit) for a subset.    python tests/diag/gpu_sweep.py <ndraw> <ntables> [batch]   (batch: the build variant bench.py uses for large batches)"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')  # run from the repo root
import numpy as np
from egdst_amd import build, runtime, workloads
from oracle_harness import Oracle
from parity import compare
nd, nt = int(sys.argv[1]), int(sys.argv[2])
m, gen = workloads.c2(a0=0)
lib = build.build_model(m, extra_flags=workloads.BATCH_BUILD_FLAGS['C2'] if len(sys.argv) > 3 and sys.argv[3] == 'batch' else ())
print('library:', lib.path)
P = gen(nd)
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
s.set_params(P); s.solve(raise_on_error=False); s.solve(raise_on_error=False)   # second solve: adaptive schedule
st, wh = s.status(); ev = s.evals()[1]
orc = Oracle(m)
bad = 0
t = time.time()
refs = {}
for i in range(nd):
    r = orc.solve(P[i])
    if i < nt: refs[i] = r
    ok = ((st[i] == 0) == (r.rc == 0)) and (st[i] != 0 or ev[i] == r.nevals)
    if st[i] != 0 and r.rc != 0:
        ok = ok and lib.lib.egdst_strerror(int(st[i])).decode().strip() == r.err.strip()
    if i % 256 == 255:
        print('... %d draws checked, %d mismatches' % (i + 1, bad), flush=True)   # (a long silent run looks hung to gpurun)
    if not ok:
        bad += 1
        print('MISMATCH draw', i, P[i].round(4).tolist(), 'gpu', st[i], wh[i].tolist(), ev[i], '| oracle', r.rc, r.err.strip()[:50], r.nevals, flush=True)
print('%d draws: %d failed on both sides, %d mismatches (%.0f s of oracle)' % (nd, int((st != 0).sum()), bad, time.time() - t), flush=True)
s.close()
s = runtime.Solver(lib, m.descriptor(), ndraw=nt, keep_history=True)
s.set_params(P[:nt]); s.solve(raise_on_error=False)
tb = 0
for i in range(nt):
    ok, rep = compare(s.solution(i), refs[i], 0.0, 0.0)
    if not ok:
        tb += 1
        print('TABLE MISMATCH draw', i, rep['problems'][:3], flush=True)
print('%d draws compared table by table: %d mismatches' % (nt, tb))

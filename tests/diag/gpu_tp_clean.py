"""Diagnostic: what the draws on which the reference algorithm breaks down cost a batch -- the same batch with those draws replaced
by copies of a draw that solves; and with the draws that needed guess streams regenerated replaced as well."""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
flags = sys.argv[2:]
m, gen = workloads.c2(a0=0)
lib = build.build_model(m, extra_flags=flags)
P = gen(nd)
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
def run(P, tag):
    s.set_params(P); s.solve(raise_on_error=False)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); s.solve(raise_on_error=False); ts.append((time.perf_counter() - t) * 1e3)
    st = s.status()[0]; rg = s.regenerations()
    print('%s: %s ms; failed %d, draws with regenerated streams %d (streams %d), tp %s' % (tag, ['%.1f' % t for t in ts], int((st != 0).sum()), int((rg > 0).sum()), int(rg.sum()), s.tp_stats().sum(axis=0).tolist()), flush=True)
    return st, rg
st, rg = run(P, 'all draws')
good = np.nonzero((st == 0) & (rg == 0))[0]
P2 = P.copy(); P2[st != 0] = P[good[:int((st != 0).sum())]]
st2, rg2 = run(P2, 'failing draws replaced')
P3 = P2.copy(); bad = np.nonzero((st2 != 0) | (rg2 > 0))[0]; P3[bad] = P[good[np.arange(len(bad)) % len(good)]]
run(P3, 'failing and regenerating draws replaced')

"""Diagnostic: which draws of a C2 batch need k_fixup (sequential regeneration of a guess stream), and how often."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
import os
m, gen = workloads.c2(a0=float(os.environ.get('EGDST_DIAG_A0', '0')))
lib = build.build_model(m, extra_flags=['-DEGDST_FIXSTAT'] + sys.argv[2:])
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = gen(nd)
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
s.set_params(P); s.solve(raise_on_error=False)
c = s.regenerations().astype(np.int64)
st = s.status()[0]
print('draws %d, failed %d; regenerated streams per solve: total %d; draws with none %d, 1-5: %d, 6-20: %d, 21-59: %d, >=60: %d' % (
    nd, int((st != 0).sum()), int(c.sum()), int((c == 0).sum()), int(((c >= 1) & (c <= 5)).sum()), int(((c >= 6) & (c <= 20)).sum()),
    int(((c >= 21) & (c < 60)).sum()), int((c >= 60).sum())))
order = np.argsort(-c)
print('top draws:', [(int(d), int(c[d]), int(st[d])) for d in order[:12]])
for g in range(0, nd, 256):
    print('group of draws %d..: regenerations %d' % (g, int(c[g:g + 256].sum())), end='; ')
print()
print('by parameter (duw, wage, sigma) of the draws with >= 20 regenerations:', np.round(P[c >= 20][:8, [0, 2, 3]], 3).tolist())

D = np.stack([s.debug(d) for d in range(nd)]).astype(np.int64)
nb, ns, nr, nst = D[:, 8].sum(), D[:, 9].sum(), D[:, 10].sum(), D[:, 11].sum()
ticks = D[:, 12:14].copy().view(np.uint64).sum() if False else (D[:, 12] & 0xffffffff).sum() + (D[:, 13] << 32).sum()
print('both solves: %d regenerated streams: per stream %.1f batches of 256 guesses, %.1f guesses one at a time, %.1f resends, %.0f us' % (
    nst, nb / nst, ns / nst, nr / nst, ticks * 1e-2 / nst))

"""Diagnostic (-DEGDST_TPSTAMPS): where k_tp_sort spends its time, summed over the draws of a batch (ticks of 10 ns)."""
import os, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wl = os.environ.get('EGDST_DIAG_WL', 'C2')   # (C5, C3: the same stamps inside k_envelope's global-memory sorts)
if wl == 'C2':
    os.environ['EGDST_ENV_TP'] = '1'
    m, gen = workloads.c2(a0=float(os.environ.get('EGDST_DIAG_A0', '0')))
else:
    m, gen = workloads.WORKLOADS[wl]()
lib = build.build_model(m, extra_flags=['-DEGDST_TPSTAMPS'] + sys.argv[2:])
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
s.set_params(gen(nd))
s.solve(raise_on_error=False)
b0 = np.stack([s.debug(i).view(np.uint64) for i in range(nd)]).astype(np.float64)
s.solve(raise_on_error=False)
d = np.stack([s.debug(i).view(np.uint64) for i in range(nd)]).astype(np.float64) - b0
t = d.sum(axis=0)
print('draws', nd, 'sorts', int(t[4]), 'with a list out of order', int(t[5]))
print('summed workgroup time, ms: staging+order check %.1f  presort %.1f  ranks %.1f  tail %.1f | prologue %.1f  whole kernel %.1f' % tuple(
    x * 1e-5 for x in (t[0], t[1], t[2], t[3], t[6], t[7])))
w = d[:, 7]
print('per draw whole-kernel ms: median %.3f max %.3f; draws above 10x median: %d' % (np.median(w) * 1e-5, w.max() * 1e-5, int((w > 10 * np.median(w)).sum())))
print('per sort us: %.1f' % (t[7] * 1e-2 / max(t[4], 1)))
bad = d[:, 5]
idx = np.argsort(-bad)[:12]
print('draws with most sorts of lists out of order:', [(int(i), int(bad[i]), '%.1f ms' % (d[i, 1] * 1e-5)) for i in idx])
print('draws with any:', int((bad > 0).sum()), 'of', nd, '; total presort time %.1f ms, max per draw %.2f ms' % (d[:, 1].sum() * 1e-5, d[:, 1].max() * 1e-5))

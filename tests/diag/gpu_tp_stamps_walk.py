"""Diagnostic (-DEGDST_TPSTAMPS_WALK): where k_tp_walk spends its time, summed over the draws of a batch (ticks of 10 ns)."""
import os, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
os.environ['EGDST_ENV_TP'] = '1'
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m, gen = workloads.c2(a0=float(os.environ.get('EGDST_DIAG_A0', '0')))
lib = build.build_model(m, extra_flags=['-DEGDST_TPSTAMPS_WALK'] + sys.argv[2:])
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
s.set_params(gen(nd))
s.solve(raise_on_error=False)
b0 = np.stack([s.debug(i).view(np.uint64) for i in range(nd)]).astype(np.float64)
s.solve(raise_on_error=False)
d = np.stack([s.debug(i).view(np.uint64) for i in range(nd)]).astype(np.float64) - b0
t = d.sum(axis=0)
for st in (0, 1):
    n = max(t[6 + st], 1)
    print('stage %d: %d walks; per walk us: set-up+load %.1f  walk %.1f  rest %.1f' % (st, int(t[6 + st]), t[3 * st] * 1e-2 / n, t[3 * st + 1] * 1e-2 / n, t[3 * st + 2] * 1e-2 / n))
w = d[:, 1] / np.maximum(d[:, 6], 1) * 1e-2
print('stage-0 walk us per draw: median %.1f  p90 %.1f  max %.1f' % (np.median(w), np.percentile(w, 90), w.max()))
w = d[:, 4] / np.maximum(d[:, 7], 1) * 1e-2
print('stage-1 walk us per draw: median %.1f  p90 %.1f  max %.1f' % (np.median(w), np.percentile(w, 90), w.max()))

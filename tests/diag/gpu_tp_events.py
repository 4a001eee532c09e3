"""Diagnostic (-DEGDST_STAMPS): generic steps (events) and regular batches of the envelope walks of a batch, and their ticks."""
import os, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
os.environ['EGDST_ENV_TP'] = '1'
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m, gen = workloads.c2(a0=float(os.environ.get('EGDST_DIAG_A0', '0')))
lib = build.build_model(m, extra_flags=['-DEGDST_STAMPS'] + sys.argv[2:])
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
s.set_params(gen(nd))
s.solve(raise_on_error=False)
b0 = np.stack([s.debug(i).view(np.uint64) for i in range(nd)])
s.solve(raise_on_error=False)
d = np.stack([s.debug(i).view(np.uint64) for i in range(nd)]) - b0
cnt = d[:, 3]
nstep = int((cnt & np.uint64(0xffffffff)).sum()); nbatch = int((cnt >> np.uint64(32)).sum())
tstep = float(d[:, 4].sum()) * 1e-2; tbatch = float(d[:, 7].sum()) * 1e-2
cells = nd * 59
print('draws %d: generic steps %d (%.1f per cell), %.2f us each; regular batches %d (%.1f per cell), %.2f us each' % (
    nd, nstep, nstep / cells, tstep / max(nstep, 1), nbatch, nbatch / cells, tbatch / max(nbatch, 1)))
print('per cell: steps %.1f us, batches %.1f us' % (tstep / cells, tbatch / cells))
per = (cnt & np.uint64(0xffffffff)).astype(np.float64) / 59
print('generic steps per cell by draw: median %.1f  p90 %.1f  max %.1f' % (np.median(per), np.percentile(per, 90), per.max()))

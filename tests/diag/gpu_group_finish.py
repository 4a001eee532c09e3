"""Diagnostic: how evenly the draw groups of a C2 batch finish, and each group's summed kernel time per class (HIP events on the
group streams, egdst_get_group_profile).   python tests/diag/gpu_group_finish.py [a0=-5] [ndraw=4096]
(EGDST_DIAG_WL=C5 / C4 / C3: that stress workload on its batch build instead; a0 is then ignored)"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
a0 = float(sys.argv[1]) if len(sys.argv) > 1 else -5.0
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
import os
wl = os.environ.get('EGDST_DIAG_WL', 'C2')
m, gen = workloads.c2(a0=a0) if wl == 'C2' else workloads.WORKLOADS[wl]()
if gen is None:
    gen = lambda n: np.tile(m.param_vector(), (n, 1))
lib = build.build_model(m, extra_flags=workloads.BATCH_BUILD_FLAGS.get(wl, []))
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
s.set_params(gen(nd)); s.solve(raise_on_error=False); s.solve(raise_on_error=False)
s.set_profile(True)
s.solve(raise_on_error=False)          # (creates the events)
t = time.perf_counter(); s.solve(raise_on_error=False); ms = (time.perf_counter() - t) * 1e3
fin, cls = s.group_profile()
names = runtime.Solver.PROFILE_CLASSES
print('a0=%g ndraw=%d: solve %.1f ms (events on); groups finish after (ms): %s' % (a0, nd, ms, ' '.join('%.0f' % x for x in fin)))
print('   spread: min %.1f  median %.1f  max %.1f' % (fin.min(), np.median(fin), fin.max()))
print('   per group, summed HIP-event ms by class (the wait in the hardware queue included):')
print('   %-6s' % 'group' + ''.join('%13s' % n for n in names) + '%10s' % 'sum')
for g in range(len(fin)):
    print('   %-6d' % g + ''.join('%13.1f' % x for x in cls[g]) + '%10.1f' % cls[g].sum())
print('   %-6s' % 'mean' + ''.join('%13.1f' % x for x in cls.mean(axis=0)) + '%10.1f' % cls.sum(axis=1).mean())
st = s.status()[0]
ng = len(fin)
per = [int((st[g * nd // ng:(g + 1) * nd // ng] != 0).sum()) for g in range(ng)]
print('   failed draws per group:', per, ' regenerated streams per group:', [int(s.regenerations()[g * nd // ng:(g + 1) * nd // ng].sum()) for g in range(ng)])
s.close()

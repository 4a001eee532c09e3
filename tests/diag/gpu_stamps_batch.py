"""Diagnostic: time inside k_envelope workgroups summed over a whole batch (build with -DEGDST_STAMPS), next to the wall time
of the solve: sum / 256 CUs is the floor one-workgroup-per-CU puts under a batch.  Compare with the same draws solved
one at a time (an idle GPU) to see what co-resident kernels cost the walking wave."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
m, gen = workloads.c2(a0=0)
import os
variant = os.environ.get('EGDST_HIPCC_EXTRA', '')
lib = build.build_model(m, build_dir='egdst_amd/_models/_stamps' + ''.join(c for c in variant if c.isalnum()), extra_flags=['-DEGDST_STAMPS', '-DEGDST_STAMPS5'])
print('build variant:', variant or '(default)')
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = gen(nd)
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
s.set_params(P); s.solve(raise_on_error=False)
t = time.perf_counter(); s.solve(raise_on_error=False); dt = time.perf_counter() - t
tot = np.zeros(8)
for d in range(nd):
    tot += s.debug(d).view(np.uint64).astype(np.float64)
tot *= 1e-2 / 2   # ticks of 10 ns -> us, two solves accumulated
print('batch of %d: solve %.1f ms | per solve, summed over workgroups: stop+compact %.0f ms, sort %.0f ms, walk %.0f ms -> total/256 CUs = %.0f ms; whole workgroups (entry to exit) / 256 CUs = %.0f ms' % (
    nd, dt * 1e3, tot[2] * 1e-3, tot[5] * 1e-3, tot[6] * 1e-3, (tot[2] + tot[5] + tot[6]) * 1e-3 / 256, tot[0] * 1e-3 / 256))
s.close()
s1 = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=False)
acc = np.zeros(8); n1 = min(nd, 64)
for d in range(n1):
    s1.set_params(P[d:d + 1]); s1.solve(raise_on_error=False)
    b0 = s1.debug(0).view(np.uint64).astype(np.float64)
    s1.solve(raise_on_error=False)
    acc += s1.debug(0).view(np.uint64).astype(np.float64) - b0
acc *= 1e-2
print('the first %d draws one at a time: per draw stop+compact %.2f ms, sort %.2f ms, walk %.2f ms' % (n1, acc[2] / n1 * 1e-3, acc[5] / n1 * 1e-3, acc[6] / n1 * 1e-3))

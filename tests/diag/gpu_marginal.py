"""Diagnostic: marginal cost of the kernel classes in the 16-stream batch -- the step time with a class launched twice
(EGDST_DIAG_DOUBLE, egdst_host.inc) minus the plain step time.   python tests/diag/gpu_marginal.py [a0=-5] [ndraw=4096]"""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
a0 = float(sys.argv[1]) if len(sys.argv) > 1 else -5.0
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
m, gen = workloads.c2(a0=a0)
lib = build.build_model(m, extra_flags=workloads.BATCH_BUILD_FLAGS['C2'])
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
P = gen(nd)
s.set_params(P); s.solve(raise_on_error=False); s.solve(raise_on_error=False)
ref = None
base = None
for name, bits in (('plain', 0), ('k_probe x2', 1), ('grid x2', 2), ('k_tp_prep x2', 4), ('k_tp_sort0 x2', 8), ('k_tp_sort1 x2', 16), ('all five x2', 31), ('plain', 0)):
    os.environ['EGDST_DIAG_DOUBLE'] = str(bits)
    ts = []
    for k in range(4):
        t = time.perf_counter(); s.solve(raise_on_error=False); ts.append((time.perf_counter() - t) * 1e3)
    t = float(np.median(ts))
    sig = (s.status()[0].copy(), s.evals()[1].copy(), s.objective().copy())
    if ref is None:
        ref, base = sig, t
    same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(sig, ref))
    print('%-28s %7.1f ms  (+%5.1f)  same results %s' % (name, t, t - base, same), flush=True)

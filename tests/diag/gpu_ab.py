"""Diagnostic: step time of a C2 batch for several build variants (flags on top of the batch build; ';'-separated sets) and/or
environment settings, results compared with the first.   python tests/diag/gpu_ab.py <a0> <ndraw> "<flags>;<flags>;..." [ENV=val,...;...]"""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
a0, nd = float(sys.argv[1]), int(sys.argv[2])
fsets = [f.split() for f in sys.argv[3].split(';')]
esets = [dict(kv.split('=') for kv in e.split(',') if kv) for e in (sys.argv[4].split(';') if len(sys.argv) > 4 else [''])]
m, gen = workloads.c2(a0=a0)
P = gen(nd)
ref = None
for fl in fsets:
    lib = build.build_model(m, extra_flags=workloads.BATCH_BUILD_FLAGS['C2'] + fl)
    for env in esets:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
        s.set_params(P); s.solve(raise_on_error=False); s.solve(raise_on_error=False)
        ts = []
        for k in range(5):
            s.set_params(P * (1 + 0.005 * (2 * np.random.default_rng(k).random(P.shape) - 1)))
            t = time.perf_counter(); s.solve(raise_on_error=False); ts.append((time.perf_counter() - t) * 1e3)
        s.set_params(P); s.solve(raise_on_error=False)
        sig = (s.status()[0].copy(), s.evals()[1].copy(), s.objective().copy())
        if ref is None:
            ref = sig
        same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(sig, ref))
        print('%-40s %-30s median %.1f ms  %s  same results %s' % (' '.join(fl) or '(batch build)', env or '', float(np.median(ts)), ['%.0f' % t for t in ts], same), flush=True)
        s.close()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

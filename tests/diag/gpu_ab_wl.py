"""Diagnostic: step time of a batch of a stress workload (C4, C5, C3) for several environment settings on its batch build.
   python tests/diag/gpu_ab_wl.py <workload> <ndraw> "ENV=val,...;ENV=val;..." ["extra flags;extra flags"]"""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
wl, nd = sys.argv[1], int(sys.argv[2])
esets = [dict(kv.split('=') for kv in e.split(',') if kv) for e in (sys.argv[3].split(';') if len(sys.argv) > 3 else [''])]
fsets = [f.split() for f in (sys.argv[4].split(';') if len(sys.argv) > 4 else [''])]
m, gen = workloads.WORKLOADS[wl]()
P = gen(nd) if gen else np.tile(m.param_vector(), (nd, 1))
ref = None
for fl in fsets:
    lib = build.build_model(m, extra_flags=workloads.BATCH_BUILD_FLAGS.get(wl, []) + fl)
    for env in esets:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
        s.set_params(P); s.solve(raise_on_error=False)
        ts = []
        for k in range(3):
            t = time.perf_counter(); s.solve(raise_on_error=False); ts.append((time.perf_counter() - t) * 1e3)
        sig = (s.status()[0].copy(), s.evals()[1].copy(), s.objective().copy())
        if ref is None:
            ref = sig
        same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(sig, ref))
        print('%s x %d %-24s %-40s median %.1f ms  %s  same results %s' % (wl, nd, ' '.join(fl), env or '', float(np.median(ts)), ['%.0f' % t for t in ts], same), flush=True)
        s.close()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

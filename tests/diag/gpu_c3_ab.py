"""Diagnostic: C3 single solve, default build against a variant (flags in argv)."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
wl = sys.argv[1]
m, _ = workloads.WORKLOADS[wl]()
ref = None
for name, flags, bdir in (('default', [], None), ('variant', sys.argv[3:], sys.argv[2])):
    lib = build.build_model(m, build_dir=bdir, extra_flags=flags)
    s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=True)
    s.set_params(m.param_vector()[None]); s.solve()
    ts = []
    for _ in range(4):
        t = time.perf_counter(); s.solve(); ts.append((time.perf_counter() - t) * 1e3)
    ck = s.checksums(0)
    s.set_profile(True); s.solve()
    print(wl, name, flags, ['%.2f' % t for t in ts], 'kernel ms', np.round(s.profile()[0], 2).tolist(), 'same' if ref is None else bool(np.array_equal(ck, ref)), flush=True)
    ref = ck if ref is None else ref
    s.close()

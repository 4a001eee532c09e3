"""Diagnostic: a batched solve with the grid kernel's C and V columns in global memory (EGDST_GRID_CV=0) and staged in LDS with the
   M column (1: k_grid_lds_cv), per build variant; same bits.   python tests/diag/gpu_grid_cv.py WL NDRAW [flag+flag ...]"""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
wl, nd = sys.argv[1], int(sys.argv[2])
m, gen = workloads.WORKLOADS[wl]()
P = gen(nd) if gen else np.tile(m.param_vector(), (nd, 1))
base = workloads.BATCH_BUILD_FLAGS.get(wl, []) if nd >= workloads.BATCH_BUILD_MIN_DRAWS.get(wl, 1 << 30) else []
ref = None
for var in (sys.argv[3:] or ['']):
    flags = base + [f for f in var.split('+') if f]
    lib = build.build_model(m, extra_flags=flags)
    for on in ('0', '1'):
        os.environ['EGDST_GRID_CV'] = on
        s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
        s.set_params(P); s.solve(raise_on_error=False)
        ts = []
        for _ in range(3):
            t = time.perf_counter(); s.solve(raise_on_error=False); ts.append((time.perf_counter() - t) * 1e3)
        r = (s.status()[0].copy(), s.evals()[1].copy(), s.objective().copy())
        same = None if ref is None else bool(np.array_equal(r[0], ref[0]) and np.array_equal(r[1], ref[1]) and np.array_equal(r[2], ref[2], equal_nan=True))
        ref = ref or r
        s.set_groups(1); s.set_profile(True); s.solve(raise_on_error=False)
        print('%s x %d %s CV=%s ms=%s failed=%d one-group kernel ms %s same=%s' % (wl, nd, flags, on, ['%.1f' % t for t in ts], int((r[0] != 0).sum()), np.round(s.profile()[0], 1).tolist(), same), flush=True)
        s.close()

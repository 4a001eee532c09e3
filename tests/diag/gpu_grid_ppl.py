"""Diagnostic: a batched solve with a lane per asset point (EGDST_GRID_PPL=0) and with several points per lane (1), per build variant.
   python tests/diag/gpu_grid_ppl.py WL NDRAW [flag+flag ...]      e.g.  C4 32 -DGRID_MINW=8+-DGRID_BS=1024"""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
wl, nd = sys.argv[1], int(sys.argv[2])
variants = sys.argv[3:] or ['']
m, gen = workloads.WORKLOADS[wl]()
P = gen(nd) if gen else np.tile(m.param_vector(), (nd, 1))
ref = None
for var in variants:
    flags = [f for f in var.split('+') if f]
    lib = build.build_model(m, extra_flags=list(flags) + ['-DEGDST_WITH_GRID_PPL'])   # (k_grid_lds_n is compiled into diagnostic builds only)
    for ppl in ('0', '1'):
        os.environ['EGDST_GRID_PPL'] = ppl
        s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
        s.set_params(P); s.solve(raise_on_error=False)
        ts = []
        for _ in range(2):
            t = time.perf_counter(); s.solve(raise_on_error=False); ts.append((time.perf_counter() - t) * 1e3)
        st, ev, ob = s.status()[0], s.evals()[1], s.objective()
        same = None
        if ref is None:
            ref = (st.copy(), ev.copy(), ob.copy())
        else:
            same = bool(np.array_equal(st, ref[0]) and np.array_equal(ev, ref[1]) and np.array_equal(ob, ref[2], equal_nan=True))
        s.set_profile(True); s.solve(raise_on_error=False)
        print('%s x %d flags=%s PPL=%s ms=%s failed=%d kernel ms (probe, grid, env, regen) %s same=%s' % (
            wl, nd, flags, ppl, ['%.1f' % t for t in ts], int((st != 0).sum()), np.round(s.profile()[0], 1).tolist(), same), flush=True)
        s.close()

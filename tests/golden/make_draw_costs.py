"""Recorded per-draw cost figures of the north_star's strong-scaling batches (C5 x 1024 draws, C4 x 256 draws), for the CPU test of the
cost-aware sharding (tests/test_multirank.py): evaluations, re-basing calls and status of every draw, and the wall time of every
chunk of the handle, measured on one MI355X.    python tests/golden/make_draw_costs.py   (on a GPU box; ~1 minute)
Data only: written to tests/golden/draw_costs_<workload>.npz."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
from egdst_amd import build, runtime, workloads

out_dir = sys.argv[1] if len(sys.argv) > 1 else HERE
for wl, total, chunk in (('C5', 1024, 128), ('C4', 256, 32)):
    m, gen = workloads.WORKLOADS[wl]()
    P = gen(total)
    lib = build.build_model(m, extra_flags=workloads.BATCH_BUILD_FLAGS[wl])
    s = runtime.Solver(lib, m.descriptor(), ndraw=chunk, keep_history=False)
    s.set_params(P[:chunk]); s.solve(raise_on_error=False)
    ev, wk, st, tms = [], [], [], []
    for c0 in range(0, total, chunk):
        s.set_params(P[c0:c0 + chunk])
        t = time.perf_counter(); s.solve(raise_on_error=False); tms.append((time.perf_counter() - t) * 1e3)
        ev.append(s.evals()[1].copy()); wk.append(s.work().copy()); st.append(s.status()[0].copy())
        print(wl, c0, '%.1f ms' % tms[-1], flush=True)
    s.close()
    np.savez_compressed(os.path.join(out_dir, 'draw_costs_%s.npz' % wl), evals=np.concatenate(ev), work=np.concatenate(wk),
                        status=np.concatenate(st), chunk_ms=np.array(tms), chunk=chunk)

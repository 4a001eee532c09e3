"""Diagnostic: a batched solve of a long-stream workload with the envelope step on k_envelope (EGDST_ENV_TP=0) and on the
   throughput path with walks over global memory (EGDST_ENV_TP=1, EGDST_TP_LONG=1); tables must be the same bits.
   python tests/diag/gpu_tp_long.py WL NDRAW [groups]      e.g.  C5 128"""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
wl, nd = sys.argv[1], int(sys.argv[2])
m, gen = workloads.WORKLOADS[wl]()
P = gen(nd) if gen else np.tile(m.param_vector(), (nd, 1))
kh = len(sys.argv) > 3 and sys.argv[3] == 'history'
flags = workloads.BATCH_BUILD_FLAGS.get(wl, []) if nd >= workloads.BATCH_BUILD_MIN_DRAWS.get(wl, 1 << 30) else []
lib = build.build_model(m, extra_flags=list(flags) + ['-DEGDST_WITH_TP_LONG'])   # (k_tp_walk_g is compiled into diagnostic builds only)
ref = None
for tp in ('0', '1'):
    os.environ['EGDST_ENV_TP'] = tp
    os.environ['EGDST_TP_LONG'] = tp
    s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=kh)
    s.set_params(P); s.solve(raise_on_error=False)
    ts = []
    for _ in range(2):
        t = time.perf_counter(); s.solve(raise_on_error=False); ts.append((time.perf_counter() - t) * 1e3)
    st, ev, ob, ck = s.status()[0], s.evals()[1], s.objective(), (np.stack([s.checksums(d) for d in range(nd)]) if kh else np.zeros(1))
    same = None
    if ref is None:
        ref = (st.copy(), ev.copy(), ob.copy(), np.array(ck).copy())
    else:
        same = bool(np.array_equal(st, ref[0]) and np.array_equal(ev, ref[1]) and np.array_equal(ob, ref[2], equal_nan=True) and np.array_equal(np.array(ck), ref[3]))
    tps = s.tp_stats().sum(axis=0).tolist()
    s.set_profile(True); s.solve(raise_on_error=False)
    print('%s x %d TP=%s ms=%s failed=%d tp done/left %s kernel ms %s same=%s' % (
        wl, nd, tp, ['%.1f' % t for t in ts], int((st != 0).sum()), tps, np.round(s.profile()[0], 1).tolist(), same), flush=True)
    s.close()

"""Diagnostic: a batched solve with the throughput path of the envelope step off and on (EGDST_ENV_TP), per build variant.
   python tests/diag/gpu_tp_compare.py [workload=C2] [ndraw=4096] [variants: default,batch]"""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
wl = sys.argv[1] if len(sys.argv) > 1 else 'C2'
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
variants = (sys.argv[3] if len(sys.argv) > 3 else 'default,batch').split(',')
m, gen = workloads.WORKLOADS[wl]()
P = gen(nd)
ref = None
for var in variants:
    flags = workloads.BATCH_BUILD_FLAGS.get(wl, []) if var == 'batch' else ([] if var == 'default' else var.split('+'))
    lib = build.build_model(m, extra_flags=flags)
    for tp in ('0', '1'):
        os.environ['EGDST_ENV_TP'] = tp
        s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
        s.set_params(P); s.solve(raise_on_error=False)
        ts = []
        for _ in range(3):
            t = time.perf_counter(); s.solve(raise_on_error=False); ts.append((time.perf_counter() - t) * 1e3)
        st, ev, ob = s.status()[0], s.evals()[1], s.objective()
        tps = s.tp_stats().sum(axis=0)
        same = None
        if ref is None:
            ref = (st.copy(), ev.copy(), ob.copy())
        else:
            same = bool(np.array_equal(st, ref[0]) and np.array_equal(ev, ref[1]) and np.array_equal(ob, ref[2], equal_nan=True))
        print('%s %s flags=%s TP=%s ms=%s evals=%d failed=%d tp_done/left=%s same_as_first=%s' % (
            wl, nd, flags, tp, ['%.1f' % t for t in ts], s.evals()[0], int((st != 0).sum()), tps.tolist(), same), flush=True)
        s.close()

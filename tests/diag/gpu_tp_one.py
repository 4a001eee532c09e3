"""Diagnostic: N solves of a C2 batch with the throughput path of the envelope step on or off (EGDST_ENV_TP set by the caller),
for rocprofv3 --kernel-trace --stats.   python tests/diag/gpu_tp_one.py [ndraw=4096] [solves=3] [variant flags...]"""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 3
flags = sys.argv[3:]
m, gen = workloads.c2(a0=0)
lib = build.build_model(m, extra_flags=flags)
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False, rows_cap=int(os.environ.get('ROWS_CAP', '0')))
s.set_params(gen(nd))
for _ in range(ns):
    t = time.perf_counter(); s.solve(raise_on_error=False); print('%.1f ms' % ((time.perf_counter() - t) * 1e3), flush=True)
print('tp', s.tp_stats().sum(axis=0).tolist(), 'failed', int((s.status()[0] != 0).sum()), 'schedule', s.schedule())

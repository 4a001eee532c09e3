"""Diagnostic: section timing inside k_envelope (build with -DEGDST_STAMPS)."""
import sys, ctypes as C
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
m, gen = workloads.c2(a0=0)
lib = build.build_model(m, build_dir='egdst_amd/_models/_stamps', extra_flags=['-DEGDST_STAMPS'])
P = gen(64)
for draw in [int(a) for a in sys.argv[1:]] or (0, 1):
    s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=False)
    s.set_params(P[draw:draw+1]); s.solve()
    s.solve()
    buf = s.debug(0).view(np.uint64)
    nb, ns = int(buf[3]) >> 32, int(buf[3]) & 0xffffffff
    print('draw', draw, '2 solves: separate classification pass %.0f us |' % (buf[1] * 1e-2), 'stop+compact %.0f us, sort %.0f us, walk %.0f us | inside walks: %d batches %.0f us (%.2f us each), %d generic steps %.0f us (%.2f us each)' % (
        buf[2] * 1e-2, buf[5] * 1e-2, buf[6] * 1e-2, nb, buf[7] * 1e-2, buf[7] * 1e-2 / max(nb, 1), ns, buf[4] * 1e-2, buf[4] * 1e-2 / max(ns, 1)))

"""Diagnostic: kernel time breakdown of C4 (Deaton, T=80, n=65536, ny=21): one draw and a batch of 32."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')  # run from the repo root
import numpy as np
from egdst_amd import build, runtime, workloads
m, gen = workloads.c4()
lib = build.build_model(m)
for nd in (1, 32):
    P = gen(nd)
    s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
    s.set_params(P); s.solve(raise_on_error=False); s.set_profile(True)
    t = time.perf_counter(); s.solve(raise_on_error=False); dt = (time.perf_counter() - t) * 1e3
    print('C4 %d draw(s): %.1f ms' % (nd, dt), 'probe/grid(+fixup)/env ms', np.round(s.profile()[0], 1).tolist(), '%.2f G evals/s' % (s.evals()[0] / dt / 1e6), 'schedule', s.schedule())
    s.close()

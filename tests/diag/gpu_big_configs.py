"""GPU check of the large configurations of SURVEY.md §8d against the oracle (first draw), plus batch timing.
    python tests/diag/gpu_big_configs.py C4 [ndraw]      Deaton stress, T=80, 65 536 points, 21 nodes
    python tests/diag/gpu_big_configs.py C5r [ndraw]     8-state retirement at the size the oracle is pinned on (T=60, n=2000)
    python tests/diag/gpu_big_configs.py C5 [ndraw]      8-state retirement, T=100, 32 768 points, 15 nodes
"""
import sys
import time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
from oracle_harness import Oracle
from parity import compare

name = sys.argv[1]
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 4
m, gen = {'C4': workloads.c4, 'C5': workloads.c5, 'C5r': lambda: workloads.c5(ngridm=2000, T=60)}[name]()
lib = build.build_model(m)
P = gen(nd)
s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=True)
s.set_params(P[:1])
t = time.perf_counter(); s.solve(raise_on_error=False); t1 = time.perf_counter() - t
t = time.perf_counter(); s.solve(raise_on_error=False); t1 = time.perf_counter() - t
sol = s.solution(0)
print(name, 'single solve %.1f ms' % (t1 * 1e3), 'status', sol.status, sol.where, 'evals', sol.nevals, 'rows', sol.total_rows(), flush=True)
t = time.perf_counter(); ref = Oracle(m).solve(P[0]); tc = time.perf_counter() - t
ok, rep = compare(sol, ref, 0.0, 0.0)
print(name, 'oracle %.1f s rc %d evals %d rows %d | bit-exact %s %s' % (tc, ref.rc, ref.nevals, ref.total_rows() if ref.rc == 0 else -1, ok, rep['problems'][:3]),
      '| status agree', (sol.status == 0) == (ref.rc == 0), flush=True)
s.close()
if nd > 1:
    s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
    s.set_params(P); s.solve(raise_on_error=False)
    t = time.perf_counter(); s.solve(raise_on_error=False); dt = time.perf_counter() - t
    st = s.status()[0]; ev = s.evals()[0]
    print(name, 'batch of %d: %.1f ms, %.2f G evals/s, failed %d' % (nd, dt * 1e3, ev / dt / 1e9, int((st != 0).sum())), flush=True)

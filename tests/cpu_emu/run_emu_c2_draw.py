"""Harness build (no sanitizer unless EMU_SANITIZE is set): ONE draw of the full-size C2 workload against the oracle, table by table.
   python tests/cpu_emu/run_emu_c2_draw.py <draw> [<draw> ...]      (environment: EGDST_ENV_TP, EGDST_TRACE_BAD ...)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'cpu_emu'))
import build_emu
from egdst_amd import build, codegen, runtime, workloads
from oracle_harness import Oracle
from parity import compare

if __name__ == '__main__':
    m, gen = workloads.c2(a0=0)
    P = gen(4096)
    text = codegen.generate_modelspec(m)
    d = os.path.join(build.MODELS_DIR, build.model_tag(m, text))
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, 'modelspec.h'), 'w').write(text)
    san = os.environ.get('EMU_SANITIZE', '0')
    lib = runtime.ModelLibrary(build_emu.build(d, {'0': False}.get(san, san), int(os.environ.get('EMU_ENV_BS', '1')), False, int(os.environ.get('EMU_WAVE', '1'))))
    orc = Oracle(m)
    for i in [int(a) for a in sys.argv[1:]]:
        s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=True)
        s.set_params(P[i][None])
        s.solve(raise_on_error=False)
        sol, ref = s.solution(0), orc.solve(P[i])
        ok, rep = compare(sol, ref, 0.0, 0.0)
        print('draw', i, 'ok=%s status=%d/%d rows=%d/%d evals=%d/%d tp done/left %s %s' % (ok, sol.status, ref.rc, sol.total_rows(), ref.total_rows(), sol.nevals, ref.nevals, s.tp_stats()[0].tolist(), rep['problems'][:2]))
        s.close()

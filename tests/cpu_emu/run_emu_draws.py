"""Harness build (no sanitizer, one-thread workgroups): given draws of the full-size C2 workload, history kept and
ping-pong tables, against the oracle's return code."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'cpu_emu'))
import build_emu
from egdst_amd import build, codegen, runtime, workloads
from oracle_harness import Oracle

if __name__ == '__main__':
    m, gen = workloads.c2(a0=0)
    P = gen(1024)
    idx = [int(a) for a in sys.argv[1:]]
    text = codegen.generate_modelspec(m)
    d = os.path.join(build.MODELS_DIR, build.model_tag(m, text))
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, 'modelspec.h'), 'w').write(text)
    lib = runtime.ModelLibrary(build_emu.build(d, False, 1, False, 1))
    orc = Oracle(m)
    for kh in (True, False):
        s = runtime.Solver(lib, m.descriptor(), ndraw=len(idx), keep_history=kh)
        s.set_params(P[idx])
        s.solve(raise_on_error=False)
        st, wh = s.status()
        ev = s.evals()[1]
        for j, i in enumerate(idx):
            print('keep_history', kh, i, (int(st[j]), tuple(int(x) for x in wh[j]), int(ev[j]), orc.solve(P[i]).rc))
        s.close()

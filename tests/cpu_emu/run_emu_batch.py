"""Sanitizer build, batched draws and ping-pong tables (keep_history=0): compare every draw with the oracle."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'cpu_emu'))
import numpy as np
import build_emu
from egdst_amd import build, codegen, runtime, workloads
from oracle_harness import Oracle
from parity import compare


def run(model, P, keep_history, sanitize='address', env_bs=1, wave=1):
    text = codegen.generate_modelspec(model)
    d = os.path.join(build.MODELS_DIR, build.model_tag(model, text))
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, 'modelspec.h'), 'w').write(text)
    lib = runtime.ModelLibrary(build_emu.build(d, sanitize, env_bs, False, wave))
    s = runtime.Solver(lib, model.descriptor(), ndraw=len(P), keep_history=keep_history, rows_cap=int(os.environ.get('EMU_ROWS_CAP', '0')))
    s.set_params(P)
    s.solve(raise_on_error=False)
    st, wh = s.status()
    ev = s.evals()[1]
    orc = Oracle(model)
    res = []
    for i in range(len(P)):
        r = orc.solve(P[i])
        okc = None
        if keep_history:
            okc, _ = compare(s.solution(i), r, 0.0, 0.0)
        res.append((int(st[i]), tuple(int(x) for x in wh[i]), int(ev[i]), r.nevals, okc))
    print('geometry', s.geometry(), 'capacity_retries', s.capacity_retries)
    return res


if __name__ == '__main__':
    m, gen = workloads.c2(a0=0, ngridm=int(sys.argv[1]), T=int(sys.argv[2]), ny=5)
    P = gen(int(sys.argv[3]))
    for kh in (True, False):
        for r in run(m, P, kh, os.environ.get('EMU_SANITIZE', 'address')):
            print('keep_history', kh, r)
